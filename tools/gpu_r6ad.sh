#!/bin/bash
# round 6, session ad: how many trials beyond the dynamic stop should a later round speculate?  (check build, DSM_SPEC_MARGIN="e,f")
out=gpurun_out/${1:-r6ad}
mkdir -p $out
export TMPDIR=/tmp
for m in 8,4 16,8 32,16 4,2 8,4; do
  for w in "" "--shard-of 8 --shard-index 3"; do
    DSM_SPEC_MARGIN=$m timeout 600 python bench.py $w --steps 6 --warmup 2 --cpu-seconds 0 --no-config3 --no-extra-configs 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('margin $m | $w |', round(d['value']), round(d['ms_per_step'],1), 'verify', round(d['kernel_ms_per_step'].get('k_verify_pairs'),1), '| 0.25 regime', d.get('extra',{}).get('low_inlier_regime',{}).get('ms_per_step'))"
  done
done 2>&1 | tee $out/margins.txt
