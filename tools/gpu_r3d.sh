#!/bin/bash
# K1 with the tile-maxima epilogue: parity, matching-only rate, PMC utilisation
out=gpurun_out/r3d
mkdir -p $out
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_golden.py tests/test_parity_fullsize_gpu.py -m gpu -x -q > $out/pytest_match.log 2>&1; tail -4 $out/pytest_match.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-verify --cpu-seconds 0 > $out/bench_match_only.json 2> $out/bench_match_only.err; python -c "
import json; d=json.load(open('$out/bench_match_only.json')); print(d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])"
timeout 600 python tools/collect_pmc.py --util1 --out $out/k1_pmc.json --steps 1 --warmup 0 --cpu-seconds 0 --no-verify > /dev/null 2> $out/pmc.err
python -c "
import json; d=json.load(open('$out/k1_pmc.json'))
for k in ('SQ_VALU_MFMA_BUSY_CYCLES','SQ_BUSY_CU_CYCLES','SQ_INSTS_VALU','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','k1_traffic_bytes_per_launch'):
    print(k, d.get(k))
b=d['SQ_VALU_MFMA_BUSY_CYCLES']; c=d['SQ_BUSY_CU_CYCLES']
for p in ('pass1','pass2'): print(p, 'mfma busy', b[p]['mean_per_dispatch']/(4*c[p]['mean_per_dispatch']))
"
