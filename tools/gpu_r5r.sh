#!/bin/bash
# round 5: contexts (and the second lane's stream) created BEFORE the RCCL process group vs after; larger hardware-queue pool
out=gpurun_out/r5r
mkdir -p $out
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"; }
B="python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3"
for rep in 1 2; do
run "no process group" $B
run "RCCL group, contexts first (new order)" $B --force-collectives
run "RCCL group, contexts after (old order)" $B --force-collectives --ctx-after-pg
run "RCCL group, contexts after, GPU_MAX_HW_QUEUES=32" env GPU_MAX_HW_QUEUES=32 $B --force-collectives --ctx-after-pg
done | tee $out/ctx_order.txt
run "shard 3/8, RCCL group, contexts first" $B --force-collectives --shard-of 8 --shard-index 2 | tee -a $out/ctx_order.txt
run "shard 3/8, no group" $B --shard-of 8 --shard-index 2 | tee -a $out/ctx_order.txt
