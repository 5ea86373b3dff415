#!/bin/bash
# round 4, session M: a pair's correspondences staged into LDS with eight loads in flight (k_prescore, k_score_needed, k_models_score_e), A/B + suite
out=gpurun_out/r4m
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in head stage8; do
  echo -n "$v: "; DSM_LIB_PATH=$R/ab/lib_$v.so timeout 400 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step']['k_verify_pairs'], d['extra']['low_inlier_regime']['ms_per_step'])"
done; done | tee $out/ab_stage.txt
for v in head stage8; do
(cd /tmp && DSM_LIB_PATH=$R/ab/lib_$v.so DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$v -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof_$v.err)
find $out/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane_$v.csv
rm -rf $out/prof_$v
echo $v; grep "k_prescore\|k_score_needed\|k_models_score_e" $out/kernel_stats_1lane_$v.csv | cut -c1-110
done
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; grep -c "identical: True" $out/check_schedules.txt; grep "False" $out/check_schedules.txt; true
