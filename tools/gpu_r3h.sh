#!/bin/bash
out=gpurun_out/r3h
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python tools/exp_verify_knobs.py > $out/verify_knobs.txt 2>&1; cat $out/verify_knobs.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -o shard -- python $GRAFT_REPO_ROOT/bench.py --shard-of 8 --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2> $GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f > $out/trace_summary_shard8.txt 2>&1; head -60 $out/trace_summary_shard8.txt
rm -rf $out/trace
