#!/bin/bash
# BASELINE.json configs other than the default line: config 1 latency, single-GPU shards of configs 4 and 5
out=gpurun_out/${1:-cfg}
mkdir -p $out
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --images 50 --feats 1024 --uncalibrated --steps 10 --warmup 2 --cpu-seconds 0 > $out/config1.json 2>/dev/null
DSM_VERIFY_INLINE_LO=1 python bench.py --images 50 --feats 1024 --uncalibrated --steps 10 --warmup 2 --cpu-seconds 0 > $out/config1_inline_lo.json 2>/dev/null
python bench.py --images 10000 --pairs knn:200 --shard-of 8 --steps 1 --warmup 1 --cpu-seconds 0 > $out/config4_shard.json 2> $out/config4_shard.err
python bench.py --images 2000 --feats 8192 --pairs knn:200 --fixed-trials 4096 --shard-of 8 --steps 1 --warmup 1 --cpu-seconds 0 > $out/config5_shaped.json 2> $out/config5_shaped.err
python3 - <<PY
import json
for f in ("config1","config1_inline_lo","config4_shard","config5_shaped"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, round(d["value"]), "pairs/s", round(d["ms_per_step"],2), "ms/step", "hyp/s %.3g"%d["hypotheses_per_s"], d["config"]["workload"][:90], {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $out/config4_shard.err $out/config5_shaped.err
