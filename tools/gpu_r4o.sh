#!/bin/bash
# round 4, session O: the essential family's bound step with a lane per model (k_prescore_e + k_score_needed<E>) against the fused
# wave-per-hypothesis kernel (DSM_SCORE_PREFILTER=3): tests first, schedules, then the timings
out=gpurun_out/r4o
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 > $out/check_schedules_ratio025.txt 2>&1; cat $out/check_schedules_ratio025.txt
timeout 600 python tools/check_score_bounds.py > $out/score_bounds.txt 2>&1; tail -4 $out/score_bounds.txt
for rep in 1 2; do for v in 3 1; do
  echo -n "DSM_SCORE_PREFILTER=$v: "; DSM_SCORE_PREFILTER=$v timeout 400 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step']['k_verify_pairs'], d['extra']['low_inlier_regime']['ms_per_step'])"
done; done | tee $out/ab_e_lane_per_model.txt
