#!/bin/bash
# sweep of DSM_VERIFY_LANES on the default workload, its 1/8 shard and config 1 (no CPU baseline)
mkdir -p gpurun_out/lanes

for L in 2 3; do
  for cfg in "full:" "s8:--shard-of 8" "s4:--shard-of 4" "c1:--images 50 --feats 1024 --uncalibrated --steps 5"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    DSM_VERIFY_LANES=$L python bench.py --steps 2 --warmup 1 --cpu-seconds 0 $args > gpurun_out/lanes/${tag}_L$L.json 2>/dev/null
    python3 -c "
import json
d=json.load(open('gpurun_out/lanes/${tag}_L$L.json'))
print('lanes $L $tag', round(d['ms_per_step'],1), 'ms/step verify', round(d['kernel_ms_per_step']['k_verify_pairs'],1))"
  done
done
