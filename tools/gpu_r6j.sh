#!/bin/bash
# round 6, session j: lanes 2 / 3 / 4 on a shard and on the whole list, after the same-address atomics left the replay
out=gpurun_out/${1:-r6j}
mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
for lanes in 2 3 4; do
  echo -n "shard lanes $lanes: "
  DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done
done | tee $out/lanes_shard.txt
for lanes in 2 3 4; do
  echo -n "whole lanes $lanes: "
  DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done | tee $out/lanes_whole.txt
