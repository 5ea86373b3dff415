#!/bin/bash
# round 4, session F: sampler swap loop without branches (k_sample), Match() sliced under the asynchronous write-back
out=gpurun_out/r4f
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['extra'])"
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
grep "k_sample" $out/verify_kernel_stats_1lane.csv | cut -c1-120
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 > $out/bench_cli_500.txt 2>&1; cat $out/bench_cli_500.txt
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 125 > $out/bench_cli_125.txt 2>&1; grep "pairs in\|async" $out/bench_cli_125.txt
timeout 900 python tools/exp_verify_knobs.py --combos "DSM_VERIFY_LANES=2 DSM_VERIFY_GRID_DIV=1" > $out/knobs.txt 2>&1; cat $out/knobs.txt
