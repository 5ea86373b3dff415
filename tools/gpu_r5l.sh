#!/bin/bash
# round 5: kernel stats of ONE SHARD of an 8-way split (one lane: the kernels add up to the verification)
out=gpurun_out/r5l
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --shard-of 8 --shard-index 2 --steps 4 --warmup 1 --cpu-seconds 0 --no-second-regime > $R/$out/bench_shard_1lane.json 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/shard_kernel_stats_1lane.csv
rm -rf $out/prof1
head -45 $out/shard_kernel_stats_1lane.csv | cut -c1-130
