timeout 600 python -m pytest tests/test_match_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
mkdir -p gpurun_out/r3f
timeout 300 python tools/exp_k1_power.py 2>&1 | tee gpurun_out/r3f/k1_power.txt
