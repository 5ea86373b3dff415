#!/bin/bash
# round 5 experiment: how much of the 24 ms of resolve / compaction per step could hide behind another chunk's MFMA pass?  Two (four) contexts
# matching halves (quarters) of the list concurrently, no verification.
out=gpurun_out/r5s
mkdir -p $out
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), {a: round(b,1) for a,b in k.items()})"; }
B="python bench.py --no-verify --steps 4 --warmup 1 --cpu-seconds 0"
for rep in 1 2; do
run "1 context" $B
run "2 contexts, turns" $B --contexts 2
run "2 contexts, overlapping" $B --contexts 2 --no-match-lock
run "4 contexts, overlapping" $B --contexts 4 --no-match-lock
done | tee $out/match_overlap.txt
