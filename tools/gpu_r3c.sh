#!/bin/bash
out=gpurun_out/r3c
mkdir -p $out
timeout 1700 python -m pytest tests -m gpu -x -q --durations=5 > $out/pytest.log 2>&1; tail -12 $out/pytest.log
