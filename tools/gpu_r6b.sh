#!/bin/bash
# round 6, session b: k_replay_rp (the replay scans with the pair resident in LDS) -- parity suites, A/B against the old scans on the
# check build, schedules across the two builds, per-dispatch trace
out=gpurun_out/${1:-r6b}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py tests/test_golden.py tests/test_camera_models.py -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
for rep in 1 2; do
  for v in new old; do
    echo -n "$v: "
    if [ $v = old ]; then export DSM_REPLAY_LEGACY=1; else unset DSM_REPLAY_LEGACY; fi
    DSM_LIBRARY=check DSM_LIB_PATH=$PWD/dagsfm_amd/libdagsfm_mi355x_check.so timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
  done
done | tee $out/ab.txt
unset DSM_REPLAY_LEGACY
echo -n "product: "; timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])" | tee -a $out/ab.txt
for v in new old; do
  if [ $v = old ]; then export DSM_REPLAY_LEGACY=1; else unset DSM_REPLAY_LEGACY; fi
  echo -n "shard $v: "
  DSM_LIBRARY=check DSM_LIB_PATH=$PWD/dagsfm_amd/libdagsfm_mi355x_check.so timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done | tee $out/ab_shard.txt
unset DSM_REPLAY_LEGACY
timeout 900 python tools/check_schedules.py > $out/check_schedules.txt 2>&1; tail -14 $out/check_schedules.txt
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof1 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs > $GRAFT_REPO_ROOT/$out/bench_trace1.json 2> $GRAFT_REPO_ROOT/$out/err1.txt)
find $out/prof1 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace_lanes1.csv
rm -rf $out/prof1
python3 tools/trace_dispatches.py $out/kernel_trace_lanes1.csv k_replay > $out/dispatches_lanes1.txt
grep TOTAL $out/dispatches_lanes1.txt; head -12 $out/dispatches_lanes1.txt
gzip -f $out/kernel_trace_lanes1.csv
