#!/bin/bash
# round 4, session D: E scoring with the fused bound step, prescore points by scalar loads (experiment), traces of short lists
out=gpurun_out/r4d
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 900 python tools/check_schedules.py > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 > $out/check_schedules_ratio025.txt 2>&1; cat $out/check_schedules_ratio025.txt
timeout 600 python tools/fuzz_verify.py --batches 3 --pairs 1500 --seed 51 > $out/fuzz_verify.txt 2>&1; tail -2 $out/fuzz_verify.txt
for rep in 1 2; do
for v in 1 3; do
  DSM_SCORE_PREFILTER=$v timeout 400 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime > $out/bench_prefilter$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$out/bench_prefilter$v.json')); print('prefilter $v', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step']['k_verify_pairs'])"
done
done
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane.csv
rm -rf $out/prof
head -16 $out/kernel_stats_1lane.csv | cut -c1-100
(cd /tmp && DSM_VERIFY_LANES=1 DSM_SCORE_PREFILTER=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof3 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof3.err)
find $out/prof3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane_scalar_points.csv
rm -rf $out/prof3
grep "k_prescore\|k_score_needed\|k_models_score" $out/kernel_stats_1lane.csv $out/kernel_stats_1lane_scalar_points.csv | cut -c1-140
for cfg in "shard8 --shard-of 8 --shard-index 3" "config1 --images 50 --feats 1024 --uncalibrated"; do
  set -- $cfg; tag=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace_$tag -o t -- python $R/bench.py $@ --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime > $R/$out/bench_trace_$tag.json 2> $R/$out/trace_$tag.err)
  f=$(find $out/trace_$tag -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $f > $out/trace_gaps_$tag.txt 2>&1; cat $out/trace_gaps_$tag.txt
  rm -rf $out/trace_$tag
done
