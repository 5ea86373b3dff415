#!/bin/bash
# round 6, session l: the E / F bound step with a packed-f32 first stage (k_prescore_compact2): parity, the bounds self-check on three
# workloads, A/B against the pure FP64 form (check build, DSM_SCORE_PREFILTER=33), schedules
out=gpurun_out/${1:-r6l}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py tests/test_golden.py tests/test_camera_models.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
( timeout 600 python tools/check_score_bounds.py; timeout 600 python tools/check_score_bounds.py --images 150 --outlier-frac 0.5; timeout 600 python tools/check_score_bounds.py --images 200 --uncalibrated ) 2>&1 | grep -v amdgpu.ids | tee $out/score_bounds_check.txt
for rep in 1 2; do
  for v in new f64; do
    echo -n "$v: "
    if [ $v = f64 ]; then export DSM_SCORE_PREFILTER=33; else unset DSM_SCORE_PREFILTER; fi
    DSM_LIBRARY=check DSM_LIB_PATH=$PWD/dagsfm_amd/libdagsfm_mi355x_check.so timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
  done
done | tee $out/ab.txt
unset DSM_SCORE_PREFILTER
for v in new f64; do
  if [ $v = f64 ]; then export DSM_SCORE_PREFILTER=33; else unset DSM_SCORE_PREFILTER; fi
  echo -n "low-inlier $v: "
  DSM_LIBRARY=check DSM_LIB_PATH=$PWD/dagsfm_amd/libdagsfm_mi355x_check.so timeout 300 python bench.py --images 150 --outlier-frac 0.5 --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done | tee -a $out/ab.txt
unset DSM_SCORE_PREFILTER
timeout 900 python tools/check_schedules.py > $out/check_schedules.txt 2>&1; tail -13 $out/check_schedules.txt | cut -c1-200
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof1 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs > $GRAFT_REPO_ROOT/$out/bench_trace.json 2> $GRAFT_REPO_ROOT/$out/err1.txt)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane.csv
rm -rf $out/prof1
head -30 $out/kernel_stats_1lane.csv | cut -c1-120
