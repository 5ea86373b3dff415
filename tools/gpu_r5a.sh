#!/bin/bash
# round 5, first session: the GPU suite against the rebuilt library and the default bench line in its new (short) form
out=gpurun_out/r5a
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 600 python bench.py --steps 5 --warmup 1 --dump-line $out/bench_default_long.json > $out/bench_default.json 2> $out/bench_default.err
wc -c $out/bench_default.json; tail -c 3000 $out/bench_default.json; tail -3 $out/bench_default.err
