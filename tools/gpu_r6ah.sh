#!/bin/bash
# round 6, session ah: k_items_enum's workgroup size (waves per workgroup; the library in the snapshot is built with -DITEMS_GROUP=$2)
out=gpurun_out/${1:-r6ah}
mkdir -p $out
export TMPDIR=/tmp
for i in 1 2; do
for w in "--shard-of 8 --shard-index 3" "--images 50 --feats 1024" "--images 150 --feats 4096" ""; do
  timeout 600 python bench.py $w --steps 10 --warmup 3 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ITEMS_GROUP=$2 | $w |', round(d['value']), round(d['ms_per_step'],2), 'verify', round(d['kernel_ms_per_step'].get('k_verify_pairs'),2))"
done
done 2>&1 | tee $out/items_group_$2.txt
