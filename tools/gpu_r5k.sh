#!/bin/bash
# round 5: a shard's verification under the scheduling knobs of the product (item passes from the start of a round or the chain with a tail)
out=gpurun_out/r5k
mkdir -p $out
for rep in 1 2; do
for cfg in "1 8192" "0 8192" "0 2048" "0 16384" "0 0"; do
  set -- $cfg
  echo -n "shard 3/8 item_mode $1 lo_tail $2: "
  DSM_VERIFY_ITEM_MODE=$1 DSM_LO_TAIL=$2 timeout 300 python bench.py --shard-of 8 --shard-index 2 --steps 6 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done
done | tee $out/shard_item_mode.txt
