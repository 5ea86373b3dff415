#!/bin/bash
# round 6, second campaign: differential fuzz against the oracle on the final build, new seeds -- the product library (the compact solver grid, the
# register prepare kernels of all three families in two sizes, the LDS chain sums, the one-wave counter kernels) and the check build's older forms beside it
out=gpurun_out/r6fuzz2
mkdir -p $out
run() { name=$1; shift; echo "== $name: $(timeout 900 "$@" 2>&1 | grep 'FUZZ RESULT' | tail -1)" | tee -a $out/summary.txt; }
run fuzz_seed461_product python tools/fuzz_verify.py --batches 8 --pairs 2500 --seed 461
run fuzz_seed462_product python tools/fuzz_verify.py --batches 6 --pairs 2500 --seed 462
DSM_HYP_GRID=pair run fuzz_seed463_hyp_pair_grid_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 463
DSM_LO_PREPARE_WAVE=1 run fuzz_seed464_lo_prepare_wave_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 464
DSM_REPLAY_LEGACY=1 run fuzz_seed465_replay_legacy_check_build python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 465
DSM_VERIFY_LANES=2 DSM_VERIFY_CHUNK_PAIRS=37 run fuzz_sched_1_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 471
DSM_VERIFY_INLINE_LO=1 run fuzz_sched_2_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 472
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=0 DSM_LO_TAIL=4 run fuzz_sched_3_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 473
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=1 DSM_VERIFY_LANES=3 run fuzz_sched_4_product python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 474
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=0 DSM_LO_TAIL=0 DSM_VERIFY_CHUNK_PAIRS=700 run fuzz_sched_5_product python tools/fuzz_verify.py --batches 3 --pairs 2000 --seed 475
run fuzz_poison_verify python tools/fuzz_verify.py --batches 5 --pairs 500 --seed 481 --grow --poison
run fuzz_big_seed491 python tools/fuzz_verify.py --batches 4 --pairs 1200 --seed 491 --big
run fuzz_stage_seed42 python tools/fuzz_stage.py --seed 42
