#!/bin/bash
# round 6, session p: item passes in every round / in the first round(s) only / never, on the short-list workloads
out=gpurun_out/${1:-r6p}
mkdir -p $out
export TMPDIR=/tmp
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"; }
for rep in 1 2; do
for im in default 0 1 2 3; do
  if [ $im = default ]; then unset DSM_VERIFY_ITEM_MODE; else export DSM_VERIFY_ITEM_MODE=$im; fi
  echo -n "item_mode $im | 0.25 ratio 150 img: "; timeout 300 python bench.py --images 150 --outlier-frac 0.5 --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | grep "^{" | tail -1 | run
  echo -n "item_mode $im | shard 3/8: "; timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1 | run
  echo -n "item_mode $im | config 1: "; timeout 300 python bench.py --images 50 --feats 1024 --uncalibrated --steps 20 --warmup 3 --cpu-seconds 0 --no-second-regime 2>/dev/null | grep "^{" | tail -1 | run
done
done | tee $out/item_mode.txt
