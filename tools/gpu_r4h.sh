#!/bin/bash
# round 4, session H: where a short list's verification time goes (kernel stats of a 1/8 shard, one lane), how many trials the
# families really run, the CLI with the device set-up overlapped, a few knobs on the shard
out=gpurun_out/r4h
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python tools/trial_histogram.py > $out/trial_histogram.txt 2>&1; cat $out/trial_histogram.txt
timeout 300 python tools/trial_histogram.py --images 150 --outlier-frac 0.5 > $out/trial_histogram_ratio025.txt 2>&1; cat $out/trial_histogram_ratio025.txt
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --shard-of 8 --shard-index 0 --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime > $R/$out/bench_shard_1lane.json 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/shard_kernel_stats_1lane.csv
rm -rf $out/prof1
head -45 $out/shard_kernel_stats_1lane.csv | cut -c1-110
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 > $out/bench_cli_500.txt 2>&1; cat $out/bench_cli_500.txt
timeout 900 python tools/exp_verify_knobs.py --combos "DSM_LO_JACOBI_GROUPS=1;DSM_LO_PREPARE_WAVE=1;DSM_VERIFY_ITEM_MODE=1;DSM_VERIFY_ITEM_MODE=0;DSM_VERIFY_LANES=3 DSM_VERIFY_GRID_DIV=1;DSM_VERIFY_LANES=1 DSM_VERIFY_GRID_DIV=1" > $out/knobs.txt 2>&1; cat $out/knobs.txt
