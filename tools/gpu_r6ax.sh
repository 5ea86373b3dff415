#!/bin/bash
# round 6, session ax: the FLANN lane kernel at five waves per SIMD (library built with -DFLANN_LANE_WAVES=$2)
out=gpurun_out/${1:-r6ax}
mkdir -p $out
timeout 900 python tools/bench_flann_search.py > $out/flann_search_$2.json 2> /dev/null; python - <<PY
import json
d=json.loads(open("$out/flann_search_$2.json").read().strip().splitlines()[-1])
print("FLANN_LANE_WAVES=$2", json.dumps(d)[:700])
PY
timeout 600 python -m pytest tests/test_retrieval_flann.py -m gpu -x -q 2>&1 | tail -2
