#!/bin/bash
# rocprofv3 kernel stats of the default bench (no CPU baseline) -> gpurun_out/<tag>/kernel_stats.csv
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 ${@:2} > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/prof
head -32 $out/kernel_stats.csv | cut -c1-120
DSM_VERIFY_DEBUG=1 python tools/exp_verify_prof.py 150 2>&1 | tail -6
