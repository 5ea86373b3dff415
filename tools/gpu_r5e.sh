#!/bin/bash
# round 5: (1) the RCCL gather library + the new host defaults on a GPU; (2) A/B of the replay scans compiled for 5 / 6 / 8 waves per
# SIMD (k_replay_lo modes 0 and 2: latency-bound at 4 waves, 103 - 108 VGPRs) against the baseline
out=gpurun_out/r5e
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gather_rccl_gpu.py tests/test_host_shim.py tests/test_rccl_single_rank_gpu.py "tests/test_verify_gpu.py::test_bound_and_exact_scoring_regimes" "tests/test_verify_gpu.py::test_debug_options_are_per_context_and_checked" -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
for rep in 1 2; do bash tools/gpu_ab_verify.sh r5e_ab$rep base rw5 rw6 rw8; done
# a shard of an 8-way split (short lists: item passes) with the same libraries
for v in base rw6 rw8; do
  echo -n "shard $v: "
  DSM_LIB_PATH=$PWD/ab/lib_$v.so timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done | tee $out/ab_shard.txt
