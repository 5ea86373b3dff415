#!/bin/bash
# round 5: the lean product library + the check build: GPU suite, schedules compared across the two builds,
# a short bench against the compressed code objects
out=gpurun_out/r5b
mkdir -p $out
export TMPDIR=/tmp
ls -la dagsfm_amd/*.so > $out/libs.txt
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout 900 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt | cut -c1-220
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime > $out/bench_short.json 2> $out/bench_short.err; cut -c1-400 $out/bench_short.json; tail -2 $out/bench_short.err
