#!/bin/bash
# A/B of several builds of the library on the same box (whole step incl. verification): tools/gpu_ab_verify.sh <tag> <lib name>...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
for v in "$@"; do
  echo -n "$v: "
  DSM_LIB_PATH=$PWD/ab/lib_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done | tee $out/ab.txt
