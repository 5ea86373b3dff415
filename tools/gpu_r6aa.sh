#!/bin/bash
# round 6, session aa: timeline of one shard step on two lanes (what is on the critical path of a short list?)
out=gpurun_out/${1:-r6aa}
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --shard-of 8 --shard-index 3 --steps 2 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/err.txt)
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv
rm -rf $out/prof
python3 tools/trace_timeline.py $out/kernel_trace.csv 2 > $out/timeline.txt; cat $out/timeline.txt
python3 tools/trace_summary.py $out/kernel_trace.csv | head -4
gzip -f $out/kernel_trace.csv
