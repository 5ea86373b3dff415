#!/bin/bash
# which build flag broke the library load (r5b: segfault at the first launch)?  smoke() against each variant under ab/
out=gpurun_out/r5c
mkdir -p $out
for v in plain strip compress; do
  DSM_LIB_PATH=$PWD/ab/lib_$v.so timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke_$v.log 2>&1
  echo "$v rc=$? $(tail -1 $out/smoke_$v.log | cut -c1-200)"
done
