#!/bin/bash
# round 6, session ao: planar scenes (H local optimisations with hundreds of inliers): k_lo_prepare_reg<H, 3> against the general kernel
out=gpurun_out/${1:-r6ao}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 900 python tools/check_schedules.py --images 200 --planar > $out/check_schedules.txt 2>&1
timeout 900 python tools/check_schedules.py --images 200 --planar --uncalibrated >> $out/check_schedules.txt 2>&1; grep -c "identical: True" $out/check_schedules.txt; grep -c "identical: False" $out/check_schedules.txt; grep "lo_prepare_wave\|batched_check_build\|one_lane" $out/check_schedules.txt
for i in 1 2; do
timeout 300 python bench.py --images 200 --planar --steps 6 --warmup 2 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('planar 200 images:', round(d['value']), round(d['ms_per_step'],1), 'verify', round(d['kernel_ms_per_step'].get('k_verify_pairs'),1), d['config']['workload'])"
done
