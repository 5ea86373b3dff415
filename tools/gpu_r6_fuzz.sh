#!/bin/bash
# round 6: differential fuzz against the oracle on the final build, new seeds -- the product library (k_replay_rp with split counters, the E / F
# bound step in packed f32, quad scoring, the register LU, the owed Givens step) and the check build's forms of round 5 beside it
out=gpurun_out/r6fuzz
mkdir -p $out
run() { name=$1; shift; echo "== $name: $(timeout 900 "$@" 2>&1 | grep 'FUZZ RESULT' | tail -1)" | tee -a $out/summary.txt; }
run fuzz_seed361_product python tools/fuzz_verify.py --batches 8 --pairs 2500 --seed 361
run fuzz_seed362_product python tools/fuzz_verify.py --batches 6 --pairs 2500 --seed 362
DSM_REPLAY_LEGACY=1 run fuzz_seed363_replay_legacy_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 363
DSM_SCORE_PREFILTER=33 run fuzz_seed364_ef_fp64_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 364
DSM_ELU_LDS=1 run fuzz_seed365_elu_lds_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 365
DSM_SCORE_PREFILTER=0 run fuzz_seed366_no_prefilter_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 366
DSM_VERIFY_LANES=2 DSM_VERIFY_CHUNK_PAIRS=37 run fuzz_sched_1_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 371
DSM_VERIFY_INLINE_LO=1 run fuzz_sched_2_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 372
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=0 DSM_LO_TAIL=4 run fuzz_sched_3_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 373
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=1 DSM_VERIFY_LANES=3 run fuzz_sched_4_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 374
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=2 run fuzz_sched_5_product python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 375
run fuzz_poison_verify python tools/fuzz_verify.py --batches 5 --pairs 500 --seed 381 --grow --poison
run fuzz_big_seed391 python tools/fuzz_verify.py --batches 4 --pairs 1200 --seed 391 --big
run fuzz_match_seed32 python tools/fuzz_match.py --seed 32
run fuzz_stage_seed32 python tools/fuzz_stage.py --seed 32
run fuzz_retrieval_seed33 python tools/fuzz_retrieval.py --seed 33
run fuzz_host_seed32 python tools/fuzz_host.py --seed 32
