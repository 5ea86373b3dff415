#!/bin/bash
# kernel timeline (start/end per dispatch) of a small workload: where the fixed latency of a short pair list goes
# usage: tools/gpu_trace_small.sh <tag> [bench.py args]   -> gpurun_out/<tag>/kernel_trace.csv
tag=${1:-trace}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 ${@:2} > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/err.txt)
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv
rm -rf $out/prof
python3 tools/trace_summary.py $out/kernel_trace.csv
