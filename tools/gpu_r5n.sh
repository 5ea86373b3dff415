#!/bin/bash
# round 5 experiment: are 3 / 4 lanes slower because their streams share hardware queues (ROCclr maps streams onto GPU_MAX_HW_QUEUES = 4 queues)?
out=gpurun_out/r5n
mkdir -p $out
for q in 4 8 16; do
for lanes in 2 3 4; do
  echo -n "shard 3/8 GPU_MAX_HW_QUEUES $q lanes $lanes: "
  GPU_MAX_HW_QUEUES=$q DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --shard-of 8 --shard-index 2 --steps 6 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done
done | tee $out/shard_hw_queues.txt
echo -n "whole list GPU_MAX_HW_QUEUES 8 lanes 2: "; GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])" | tee -a $out/shard_hw_queues.txt
echo -n "whole list GPU_MAX_HW_QUEUES 8 lanes 3: "; GPU_MAX_HW_QUEUES=8 DSM_VERIFY_LANES=3 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])" | tee -a $out/shard_hw_queues.txt
