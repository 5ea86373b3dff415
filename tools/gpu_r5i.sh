#!/bin/bash
# round 5: lanes on a short list (one shard of an 8-way split = 15 594 pairs; the default is 2 lanes from 4 096 pairs on)
out=gpurun_out/r5i
mkdir -p $out
for rep in 1 2; do
for lanes in 2 1 3 4; do
  echo -n "shard 3/8 lanes $lanes: "
  DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 4 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done
done | tee $out/shard_lanes.txt
for lanes in 1 2 3; do
  echo -n "config 1 (1 225 pairs) lanes $lanes: "
  DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --images 50 --feats 1024 --uncalibrated --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done | tee -a $out/shard_lanes.txt
