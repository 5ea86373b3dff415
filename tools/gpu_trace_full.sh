#!/bin/bash
tag=${1:-trf}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 ${@:2} > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/err.txt)
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv
rm -rf $out/prof
python3 tools/trace_timeline.py $out/kernel_trace.csv 10
gzip -f $out/kernel_trace.csv
