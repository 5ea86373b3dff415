#!/bin/bash
# BASELINE configs[2]: 2 000 images x 4 096 features, exhaustive matching only; MFMA K1 vs the LDS-tiled v_dot4 variant,
# each with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE / SQ instruction + busy counters, separate passes).
tag=${1:-k1var}
n=${2:-2000}
out=gpurun_out/$tag
mkdir -p $out
python bench.py --images $n --no-verify --steps 1 --warmup 1 --cpu-seconds 0 > $out/bench_mfma.json 2>/dev/null
python tools/collect_pmc.py --util1 --out $out/pmc_mfma.json --images $n --no-verify --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $out/pmc_mfma.err
DSM_K1_DOT4=1 python bench.py --images $n --no-verify --steps 1 --warmup 0 --cpu-seconds 0 > $out/bench_dot4.json 2>/dev/null
DSM_K1_DOT4=1 python tools/collect_pmc.py --util1 --out $out/pmc_dot4.json --images $n --no-verify --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $out/pmc_dot4.err
rm -rf gpurun_out/pmc
python3 - <<PY
import json
for v in ("mfma","dot4"):
    b=json.load(open("$out/bench_%s.json"%v)); p=json.load(open("$out/pmc_%s.json"%v))
    print(v, round(b["value"]), "pairs/s", b["kernel_ms_per_step"], {k:p.get(k) for k in ("FETCH_SIZE","WRITE_SIZE","SQ_INSTS_VALU_MFMA_I8","SQ_INSTS_VALU","SQ_BUSY_CU_CYCLES","SQ_VALU_MFMA_BUSY_CYCLES")})
PY
