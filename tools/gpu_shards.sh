#!/bin/bash
# per-rank work of the default workload split over S GPUs (first shard), both LO schedules; predicts strong scaling
mkdir -p gpurun_out/shards
for S in 8 4 2 1; do
  for IL in 0 1; do
    DSM_VERIFY_INLINE_LO=$IL python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --shard-of $S > gpurun_out/shards/s${S}_il$IL.json 2>/dev/null
    python3 -c "
import json
d=json.load(open('gpurun_out/shards/s${S}_il$IL.json'))
print('shard 1/$S inline_lo=$IL', round(d['ms_per_step'],1), 'ms/step verify', round(d['kernel_ms_per_step']['k_verify_pairs'],1), 'pairs', d['config']['pairs'])"
  done
done
