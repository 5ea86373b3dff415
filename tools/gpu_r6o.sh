#!/bin/bash
# round 6, session o: the 2 x 2 split's Givens step owed and paid after the loop (pr_hessenberg_eigenvalues): parity incl. the legacy
# schedule's scratch solvers, k_roots_e's time on config 2 and at the 0.25 inlier ratio
out=gpurun_out/${1:-r6o}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py tests/test_golden.py tests/test_camera_models.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 900 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; tail -14 $out/check_schedules.txt | cut -c1-200
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 --legacy > $out/check_schedules_025.txt 2>&1; tail -14 $out/check_schedules_025.txt | cut -c1-200
for rep in 1 2; do
echo -n "config 2: "; timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
echo -n "0.25 ratio: "; timeout 300 python bench.py --images 150 --outlier-frac 0.5 --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"
done | tee $out/ab.txt
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof1 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs > $GRAFT_REPO_ROOT/$out/bench_trace.json 2> $GRAFT_REPO_ROOT/$out/err1.txt)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane.csv
rm -rf $out/prof1
grep "k_roots_e\|k_lo_e_roots\|k_solve<1>" $out/kernel_stats_1lane.csv | cut -c1-120
