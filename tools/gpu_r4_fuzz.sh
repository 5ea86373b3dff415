#!/bin/bash
# round 4: differential fuzz of the final build against the oracle (new seeds; the scoring's bound step on and off)
out=gpurun_out/r4fuzz
mkdir -p $out
run() { name=$1; shift; echo "== $name: $(timeout 900 "$@" 2>&1 | grep 'FUZZ RESULT' | tail -1)" | tee -a $out/summary.txt; }
run fuzz_seed161 python tools/fuzz_verify.py --batches 8 --pairs 2500 --seed 161
run fuzz_seed162 python tools/fuzz_verify.py --batches 8 --pairs 2500 --seed 162
DSM_SCORE_PREFILTER=0 run fuzz_seed163_no_prefilter python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 163
DSM_VERIFY_LANES=2 DSM_VERIFY_CHUNK_PAIRS=37 run fuzz_sched_1 python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 171
DSM_VERIFY_INLINE_LO=1 run fuzz_sched_2 python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 172
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=0 DSM_LO_TAIL=4 run fuzz_sched_3 python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 173
DSM_VERIFY_INLINE_LO=0 DSM_VERIFY_ITEM_MODE=1 DSM_VERIFY_LANES=3 run fuzz_sched_4 python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 174
DSM_SAMPLER_SERIAL=1 DSM_VERIFY_FIXED_BATCH=1 run fuzz_sched_6 python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 176
run fuzz_poison_verify python tools/fuzz_verify.py --batches 5 --pairs 500 --seed 181 --grow --poison
run fuzz_big_seed191 python tools/fuzz_verify.py --batches 4 --pairs 1200 --seed 191 --big
run fuzz_match_seed12 python tools/fuzz_match.py --seed 12
run fuzz_stage_seed12 python tools/fuzz_stage.py --seed 12
run fuzz_retrieval_seed13 python tools/fuzz_retrieval.py --seed 13
run fuzz_host_seed12 python tools/fuzz_host.py --seed 12
