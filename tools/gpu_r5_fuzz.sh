#!/bin/bash
# round 5: differential fuzz against the oracle, new seeds.  The product library (default schedule and its scheduling knobs) and the check build
# (a cross-check switch in the environment selects it: the binding creates the context from libdagsfm_mi355x_check.so) -- incl. the H bound
# step on the FP64 matrix pipe (DSM_SCORE_PREFILTER=9)
out=gpurun_out/r5fuzz
mkdir -p $out
run() { name=$1; shift; echo "== $name: $(timeout 900 "$@" 2>&1 | grep 'FUZZ RESULT' | tail -1)" | tee -a $out/summary.txt; }
run fuzz_seed261_product python tools/fuzz_verify.py --batches 8 --pairs 2500 --seed 261
run fuzz_seed265_product_h2 python tools/fuzz_verify.py --batches 6 --pairs 2500 --seed 265
DSM_SCORE_PREFILTER=17 run fuzz_seed266_h_fp64_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 266
DSM_SCORE_PREFILTER=9 run fuzz_seed262_h_bound_on_matrix_pipe_check_build python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 262
DSM_SCORE_PREFILTER=0 run fuzz_seed263_no_prefilter_check_build python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 263
DSM_VERIFY_LANES=2 DSM_VERIFY_CHUNK_PAIRS=37 run fuzz_sched_1_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 271
DSM_VERIFY_INLINE_LO=1 run fuzz_sched_2_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 272
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=0 DSM_LO_TAIL=4 run fuzz_sched_3_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 273
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=1 DSM_VERIFY_LANES=3 run fuzz_sched_4_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 274
run fuzz_poison_verify python tools/fuzz_verify.py --batches 5 --pairs 500 --seed 281 --grow --poison
run fuzz_big_seed291 python tools/fuzz_verify.py --batches 4 --pairs 1200 --seed 291 --big
run fuzz_match_seed22 python tools/fuzz_match.py --seed 22
run fuzz_stage_seed22 python tools/fuzz_stage.py --seed 22
run fuzz_retrieval_seed23 python tools/fuzz_retrieval.py --seed 23
run fuzz_host_seed22 python tools/fuzz_host.py --seed 22
