#!/bin/bash
# round 5: lane 0 on the context's stream (one stream fewer): lanes 2 / 3 / 4 under the default hardware-queue pool and under 8
out=gpurun_out/r5o
mkdir -p $out
for q in "" 8; do
for lanes in 2 3 4; do
  echo -n "shard 3/8 GPU_MAX_HW_QUEUES '${q}' lanes $lanes: "
  env ${q:+GPU_MAX_HW_QUEUES=$q} DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --shard-of 8 --shard-index 2 --steps 6 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done
done | tee $out/shard_lanes.txt
for lanes in 2 3 4; do
  echo -n "whole list default queues lanes $lanes: "; DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done | tee -a $out/shard_lanes.txt
