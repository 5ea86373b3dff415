#!/bin/bash
# round 6, end-of-round session 3 (after the last kernel change): shard sweep of config 2 (both cuts, 5 steps), the shards of configs[3] / [4]
out=gpurun_out/r6final3
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python tools/shard_sweep.py --images 500 --feats 4096 --shards 8 --steps 5 > $out/shard_sweep_config2.txt 2> $out/shard_sweep.err; grep -v "^{" $out/shard_sweep_config2.txt
timeout 900 python bench.py --images 10000 --pairs knn:200 --shard-of 8 --shard-index 3 --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench_config4_shard4of8_10000img_knn200.json 2> $out/config4.err
timeout 1200 python bench.py --images 10000 --feats 8192 --pairs knn:200 --fixed-trials 4096 --shard-of 8 --shard-index 3 --steps 1 --warmup 1 --cpu-seconds 0 > $out/bench_config5_shard4of8_10000x8192_fixed4096.json 2> $out/config5.err
python - <<PY
import json
for f in ('bench_config4_shard4of8_10000img_knn200','bench_config5_shard4of8_10000x8192_fixed4096'):
    try:
        d=json.loads(open('$out/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d.get('hypotheses_per_s'))
    except Exception as e: print(f, 'ERR', e)
PY
