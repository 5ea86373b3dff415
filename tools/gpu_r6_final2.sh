#!/bin/bash
# round 6, end-of-round session 2: K1 variants with counters (configs[2]), shard sweep (both cuts), the other BASELINE configurations'
# shards, the drop-in CLI end to end, the retrieval row (500 x 4 096 and the 10 000-image line)
out=gpurun_out/r6final2
mkdir -p $out
export TMPDIR=/tmp
bash tools/k1_variants.sh r6final2/k1var 2000 > $out/k1_variants.txt 2>&1; tail -2 $out/k1_variants.txt | cut -c1-400
timeout 1500 python tools/shard_sweep.py --images 500 --feats 4096 --shards 8 --steps 3 > $out/shard_sweep_config2.txt 2> $out/shard_sweep.err; grep "cut:\|whole" $out/shard_sweep_config2.txt
timeout 900 python bench.py --images 10000 --pairs knn:200 --shard-of 8 --shard-index 3 --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench_config4_shard4of8_10000img_knn200.json 2> $out/config4.err
timeout 1200 python bench.py --images 10000 --feats 8192 --pairs knn:200 --fixed-trials 4096 --shard-of 8 --shard-index 3 --steps 1 --warmup 1 --cpu-seconds 0 > $out/bench_config5_shard4of8_10000x8192_fixed4096.json 2> $out/config5.err
python - <<PY
import json
for f in ('bench_config4_shard4of8_10000img_knn200','bench_config5_shard4of8_10000x8192_fixed4096'):
    try:
        d=json.loads(open('$out/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d.get('hypotheses_per_s'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 --modes "default,blocking,async+bulk_journal" > $out/bench_cli_500x4096.txt 2>&1
grep "pairs in" $out/bench_cli_500x4096.txt
for f in kdtree kmeans; do timeout 900 python tools/bench_retrieval.py --images 500 --flann $f > $out/bench_retrieval_500x4096_65536words_$f.json 2> /dev/null; cut -c1-260 $out/bench_retrieval_500x4096_65536words_$f.json; done
timeout 900 python tools/bench_retrieval.py --images 500 > $out/bench_retrieval_500x4096_65536words_exact.json 2> /dev/null; cut -c1-260 $out/bench_retrieval_500x4096_65536words_exact.json
timeout 1500 python tools/bench_retrieval.py --images 10000 --flann kdtree > $out/bench_retrieval_10000x4096_65536words_kdtree.json 2> $out/retr10000.err; cut -c1-400 $out/bench_retrieval_10000x4096_65536words_kdtree.json
timeout 900 python tools/bench_flann_search.py > $out/flann_search.json 2> /dev/null; cut -c1-200 $out/flann_search.json
