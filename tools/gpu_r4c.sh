#!/bin/bash
# round 4, session C: F / H scoring as bound + exact (k_prescore + k_score_needed): suite, schedules byte-identical at full
# size (both inlier regimes), fuzz, kernel stats
out=gpurun_out/r4c
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout 900 python tools/check_schedules.py > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 > $out/check_schedules_ratio025.txt 2>&1; cat $out/check_schedules_ratio025.txt
timeout 600 python tools/check_schedules.py --images 200 --uncalibrated > $out/check_schedules_uncal.txt 2>&1; tail -3 $out/check_schedules_uncal.txt
timeout 600 python tools/fuzz_verify.py --batches 4 --pairs 1500 --seed 41 > $out/fuzz_verify.txt 2>&1; tail -4 $out/fuzz_verify.txt
timeout 600 python tools/fuzz_verify.py --batches 2 --pairs 300 --seed 43 --big > $out/fuzz_verify_big.txt 2>&1; tail -3 $out/fuzz_verify_big.txt
for v in 1 0; do
  DSM_SCORE_PREFILTER=$v timeout 400 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $out/bench_prefilter$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$out/bench_prefilter$v.json')); print('prefilter $v', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step']['k_verify_pairs'], d.get('extra'))"
done
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane.csv
rm -rf $out/prof
head -30 $out/kernel_stats_1lane.csv | cut -c1-110
