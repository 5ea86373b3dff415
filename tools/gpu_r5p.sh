#!/bin/bash
# round 5: does an RCCL process group in the process (one more stream) push the lanes onto shared hardware queues?  r04's --force-collectives
# line had 335 ms of verification against 287.  Now lane 0 runs on the context's stream; with and without a larger queue pool.
out=gpurun_out/r5p
mkdir -p $out
for q in "" 8; do
for lanes in 2 3; do
  echo -n "whole list --force-collectives GPU_MAX_HW_QUEUES '${q}' lanes $lanes: "
  env ${q:+GPU_MAX_HW_QUEUES=$q} DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --force-collectives --steps 4 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'], 'exchange %.2f' % k['exchange'])"
done
done | tee $out/force_collectives.txt
cp ab/lib_base.so /tmp/lib_old.so
echo -n "round-4 stream layout (lane 0 on its own stream), --force-collectives, default pool: "
DSM_LIB_PATH=/tmp/lib_old.so timeout 300 python bench.py --force-collectives --steps 4 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])" | tee -a $out/force_collectives.txt
