#!/bin/bash
# round 6, session e: the driver's own command with the new line (parity_sample, extra.uncalibrated / config1)
out=gpurun_out/r6e
mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-line $out/bench_long.json > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/time.txt
echo "rc=$?"; tail -3 $out/time.txt; tail -5 $out/bench_default.err
python - <<'PY'
import json
l = open("gpurun_out/r6e/bench_default.json").read().strip().splitlines()[-1]
print(len(l), "bytes")
d = json.loads(l)
print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
print("parity_sample", d.get("parity_sample"))
for k, v in d.get("extra", {}).items():
    print(k, v)
print(d.get("from_profiles"))
PY
