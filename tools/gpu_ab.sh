#!/bin/bash
# A/B of several builds of the library on the same box (matching only): tools/gpu_ab.sh <tag> <lib name>...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$v: "
    DSM_LIB_PATH=$PWD/ab/lib_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --no-verify --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), ' '.join('%s=%.1f' % (a.split('<')[-1][:12], b) for a, b in k.items()), 'frac', round(d['roofline']['frac'],4))"
  done
done | tee $out/ab.txt
