#!/bin/bash
out=gpurun_out/r3n
mkdir -p $out
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
tools/gpu_ab_verify.sh r3n pose2 pose3 pose4
timeout 600 python tools/check_schedules.py > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
