#!/bin/bash
# round 5: the FLANN-compatible word search on the device: parity (goldens, fresh indices of the reference's FLANN, the shim's modes),
# then its rate against the host restatement's
out=gpurun_out/r5f
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_retrieval_flann.py tests/test_retrieval.py tests/test_host_shim.py -m gpu -x -q > $out/pytest.log 2>&1; tail -12 $out/pytest.log | cut -c1-250
timeout 1200 python tools/bench_flann_search.py > $out/flann_search.json 2> $out/flann_search.err; tail -c 2500 $out/flann_search.json; tail -3 $out/flann_search.err
