#!/bin/bash
# round 5, last session: the fuzz of the final verification build (new seeds), then the end-of-round session
out=gpurun_out/r5w
mkdir -p $out
run() { name=$1; shift; echo "== $name: $(timeout 900 "$@" 2>&1 | grep 'FUZZ RESULT' | tail -1)" | tee -a $out/summary.txt; }
run fuzz_seed265_product_h2 python tools/fuzz_verify.py --batches 6 --pairs 2500 --seed 265
DSM_SCORE_PREFILTER=17 run fuzz_seed266_h_fp64_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 266
run fuzz_big_seed292 python tools/fuzz_verify.py --batches 3 --pairs 1200 --seed 292 --big
run fuzz_poison_verify_282 python tools/fuzz_verify.py --batches 4 --pairs 500 --seed 282 --grow --poison
run fuzz_stage_seed24 python tools/fuzz_stage.py --seed 24
bash tools/gpu_r5_final.sh
