#!/bin/bash
# round 5: the retrieval row end to end (index + query of 500 images x 4 096 features, 65 536 words) with the exact word search and with the
# reference-identical FLANN search on the device
out=gpurun_out/r5t
mkdir -p $out
for mode in "" "--flann kdtree" "--flann kmeans"; do
  name=$(echo "exact $mode" | sed 's/exact --flann //; s/ //g')
  timeout 900 python tools/bench_retrieval.py --images 500 --feats 4096 --words 65536 $mode > $out/bench_retrieval_500x4096_$name.json 2> $out/err_$name.txt
  python -c "
import json; d=json.load(open('$out/bench_retrieval_500x4096_$name.json')); print('$name', round(d['value'],1), 'images/s index', round(d['index_device_ms'],1), 'ms query', round(d['query_device_ms'],1), 'ms pairs', d['candidate_pairs'], 'self-first', d['queries_retrieving_themselves_first'])" || tail -3 $out/err_$name.txt
done
