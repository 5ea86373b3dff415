#!/bin/bash
# round 6, session af: timeline of one config-2 step on two lanes (do the lanes end together?  where is a queue idle?)
out=gpurun_out/${1:-r6af}
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/err.txt)
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv
rm -rf $out/prof
python3 tools/trace_timeline.py $out/kernel_trace.csv 5 > $out/timeline.txt; tail -60 $out/timeline.txt
python3 tools/trace_summary.py $out/kernel_trace.csv | head -6
gzip -f $out/kernel_trace.csv
