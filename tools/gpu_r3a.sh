#!/bin/bash
# round 3, first GPU session: parity of the reference-order 5-point solver + its effect at full size
out=gpurun_out/r3a
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout 600 python tools/compare_fivept_orders.py > $out/fivept_orders_config2.txt 2>&1; cat $out/fivept_orders_config2.txt
timeout 900 python tools/compare_fivept_orders.py --images 150 --outlier-frac 0.5 > $out/fivept_orders_ratio025.txt 2>&1; cat $out/fivept_orders_ratio025.txt
timeout 600 python bench.py --steps 3 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; cat $out/bench_default.json
