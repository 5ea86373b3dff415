#!/bin/bash
# round 5 experiment: do 3 / 4 verification lanes lose to 2 because of how their host threads wait (hipDeviceSchedule*)?
out=gpurun_out/r5m
mkdir -p $out
for sched in "" spin yield blocking; do
for lanes in 2 3 4; do
  echo -n "shard 3/8 schedule '${sched}' lanes $lanes: "
  DSM_VERIFY_LANES=$lanes timeout 300 python bench.py --shard-of 8 --shard-index 2 --steps 6 --warmup 1 --cpu-seconds 0 --no-second-regime ${sched:+--hip-schedule $sched} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done
done | tee $out/shard_hip_schedule.txt
