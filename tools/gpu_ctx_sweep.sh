#!/bin/bash
# default bench with 1 / 2 / 3 / 4 contexts per GPU
mkdir -p gpurun_out/ctxsweep
for c in 1 2 3 4; do
  python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --contexts $c > gpurun_out/ctxsweep/bench_c$c.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/ctxsweep/bench_c$c.json')); print('contexts', $c, round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
done
