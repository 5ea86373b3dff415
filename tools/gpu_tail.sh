#!/bin/bash
# sweep of DSM_LO_TAIL (queue length at which a round of the batched schedule finishes inline)
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for T in 0 128 512 2048 8192; do
  for cfg in "full:" "s8:--shard-of 8"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    DSM_VERIFY_INLINE_LO=0 DSM_LO_TAIL=$T python bench.py --steps 3 --warmup 1 --cpu-seconds 0 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('tail $T $tag', round(d['ms_per_step'],1), 'ms/step verify', round(d['kernel_ms_per_step']['k_verify_pairs'],1))"
  done
done
