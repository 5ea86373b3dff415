#!/bin/bash
# round 6, third (short) campaign on the last build: the sampler's in-place swap partners, k_score_needed / k_prescore_compact2 launch bounds
out=gpurun_out/r6fuzz3
mkdir -p $out
run() { name=$1; shift; echo "== $name: $(timeout 900 "$@" 2>&1 | grep 'FUZZ RESULT' | tail -1)" | tee -a $out/summary.txt; }
run fuzz_seed561_product python tools/fuzz_verify.py --batches 8 --pairs 2500 --seed 561
DSM_VERIFY_LANES=2 DSM_VERIFY_CHUNK_PAIRS=53 run fuzz_seed562_product_chunks python tools/fuzz_verify.py --batches 4 --pairs 2500 --seed 562
DSM_SAMPLER_SERIAL=1 run fuzz_seed563_sampler_serial_check_build python tools/fuzz_verify.py --batches 2 --pairs 2000 --seed 563
run fuzz_big_seed591 python tools/fuzz_verify.py --batches 3 --pairs 1200 --seed 591 --big
