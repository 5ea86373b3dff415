#!/bin/bash
# sweep of DSM_VERIFY_LANES x DSM_VERIFY_LANE_SPLIT on the default workload and its 1/8 shard (no CPU baseline)
mkdir -p gpurun_out/lanes
for LS in "1:0.5" "2:0.5" "2:0.6" "2:0.7" "2:0.8" "3:0.5" "3:0.4"; do
  L=${LS%%:*}; S=${LS#*:}
  for cfg in "full:" "s8:--shard-of 8"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    DSM_VERIFY_LANES=$L DSM_VERIFY_LANE_SPLIT=$S python bench.py --steps 2 --warmup 1 --cpu-seconds 0 $args > gpurun_out/lanes/${tag}_L$L_$S.json 2>/dev/null
    python3 -c "
import json
d=json.load(open('gpurun_out/lanes/${tag}_L$L_$S.json'))
print('lanes $L split $S $tag', round(d['ms_per_step'],1), 'ms/step verify', round(d['kernel_ms_per_step']['k_verify_pairs'],1))"
  done
done
