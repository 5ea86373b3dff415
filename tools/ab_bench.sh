#!/bin/bash
# A/B of two builds of the library on the same box: alternate runs
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in old new; do
    cp ab/lib_$v.so dagsfm_amd/libdagsfm_mi355x.so
    echo -n "$v: "
    timeout 300 python bench.py --steps 3 --warmup 1 --no-verify --cpu-seconds 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])"
  done
done
