#!/bin/bash
# round 3 measurement session: every line DESIGN.md section 6 quotes
out=gpurun_out/r3m
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_retrieval.py tests/test_host_shim.py -m gpu -x -q > $out/pytest_retrieval.log 2>&1; tail -3 $out/pytest_retrieval.log
timeout 600 python bench.py --steps 5 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
rm -rf $out/prof
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $R/$out/bench_under_rocprof_1lane.json 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
timeout 600 python bench.py --uncalibrated --steps 3 --warmup 1 --cpu-seconds 0 > $out/bench_config2_uncalibrated.json 2>/dev/null
timeout 900 python bench.py --images 2000 --no-verify --steps 1 --warmup 1 --cpu-seconds 16 > $out/bench_2000img_match_only.json 2> $out/bench_2000.err; python -c "
import json; d=json.load(open('$out/bench_2000img_match_only.json')); print('2000 img match only', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline_native',{}).get('value'))"
timeout 900 python bench.py --images 10000 --pairs knn:200 --shard-of 8 --shard-index 3 --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench_config4_shard4of8_10000img_knn200.json 2> $out/bench_c4.err; python -c "
import json; d=json.load(open('$out/bench_config4_shard4of8_10000img_knn200.json')); print('config4 shard', d['value'], d['ms_per_step'], d['config']['workload'])"
timeout 1500 python bench.py --images 10000 --feats 8192 --pairs knn:200 --shard-of 8 --shard-index 3 --fixed-trials 4096 --steps 1 --warmup 1 --cpu-seconds 0 > $out/bench_config5_shard4of8_10000x8192_fixed4096.json 2> $out/bench_c5.err; python -c "
import json; d=json.load(open('$out/bench_config5_shard4of8_10000x8192_fixed4096.json')); print('config5 shard', d['value'], d['ms_per_step'], d['hypotheses_per_s'], d['config']['workload'])"
timeout 900 python tools/shard_sweep.py --images 2000 --pairs knn:200 --shards 8 --steps 1 > $out/shard_sweep_knn200_2000img.txt 2>&1; tail -2 $out/shard_sweep_knn200_2000img.txt | cut -c1-300
