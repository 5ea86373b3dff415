#!/bin/bash
# round 5: lanes x grid divisor on a short list (the persistent grids of several lanes sized for a lane's share of the chip)
out=gpurun_out/r5j
mkdir -p $out
for cfg in "2 1" "2 2" "3 3" "4 4" "3 2" "4 2"; do
  set -- $cfg
  echo -n "shard 3/8 lanes $1 grid_div $2: "
  DSM_VERIFY_LANES=$1 DSM_VERIFY_GRID_DIV=$2 timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 4 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done | tee $out/shard_lanes_griddiv.txt
