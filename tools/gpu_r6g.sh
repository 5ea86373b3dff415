#!/bin/bash
# round 6, session g: item-pass atomics batched (k_items_enum: one atomic per 8 pairs; H: the queue is the general list) -- parity + shard A/B
out=gpurun_out/${1:-r6g}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for rep in 1 2; do
for s in 3 6; do
  echo -n "shard $s: "
  timeout 300 python bench.py --shard-of 8 --shard-index $s --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done
done | tee $out/ab_shard.txt
echo -n "product: "; timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])" | tee -a $out/ab.txt
timeout 900 python tools/check_schedules.py > $out/check_schedules.txt 2>&1; tail -12 $out/check_schedules.txt | cut -c1-200
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof1 -o t -- python $GRAFT_REPO_ROOT/bench.py --shard-of 8 --shard-index 3 --steps 4 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$out/bench_trace_shard.json 2> $GRAFT_REPO_ROOT/$out/err1.txt)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/shard_kernel_stats_1lane.csv
rm -rf $out/prof1
head -40 $out/shard_kernel_stats_1lane.csv | cut -c1-130
