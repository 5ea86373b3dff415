#!/bin/bash
# round 4, session P: the compacted bound kernel for the fundamental family as well (k_prescore_compact<F>): suite, schedules, one bench line
out=gpurun_out/r4p
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 600 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 400 python tools/check_schedules.py --images 150 --outlier-frac 0.5 > $out/check_schedules_ratio025.txt 2>&1; cat $out/check_schedules_ratio025.txt
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['extra']['low_inlier_regime']['ms_per_step'])" | tee $out/bench.txt
