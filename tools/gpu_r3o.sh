#!/bin/bash
out=gpurun_out/r3o
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
tools/gpu_ab_verify.sh r3o pose3 score score2 pose3 score2
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $R/$out/bench_under_rocprof_1lane.json 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3o/verify_kernel_stats_1lane.csv')))
for r in rows[:24]:
    print('%-60s calls %5d  %8.2f ms/step'%(r['Name'][:60], int(r['Calls']), float(r['TotalDurationNs'])/1e6/3))
PY
