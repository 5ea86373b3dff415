#!/bin/bash
# quick GPU check: the GPU test suite, then the default bench line without the CPU baseline
tag=${1:-quick}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python bench.py --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
python3 - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print(round(d["value"]), "pairs/s", round(d["ms_per_step"],1), "ms/step", {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()}, "frac", round(d["roofline"]["frac"],3), "vfrac", round(d["roofline_verify"]["frac"],4))
PY
