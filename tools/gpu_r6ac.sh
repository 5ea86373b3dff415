#!/bin/bash
# round 6, session ac: the lane-per-hypothesis solvers of E / F on the compact grid (later rounds' hypotheses 64 per wave across the pairs):
# parity tests, schedules (hyp_pair_grid = the old grid, check build), bench before / after regimes, one-lane kernel stats
out=gpurun_out/${1:-r6ac}
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-config3 --no-extra-configs > $out/bench_$i.json 2> $out/bench_$i.err
python - <<PY
import json
d = json.loads(open("$out/bench_$i.json").read().strip().splitlines()[-1])
print('config 2:', round(d['value']), round(d['ms_per_step'],1), 'verify', d['kernel_ms_per_step'].get('k_verify_pairs'), '| 0.25 regime', d['extra']['low_inlier_regime'].get('ms_per_step'))
PY
done
timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 10 --warmup 3 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard 3/8:', round(d['value']), round(d['ms_per_step'],1), 'verify', d['kernel_ms_per_step'].get('k_verify_pairs'))"
timeout 1200 python tools/check_schedules.py > $out/check_schedules.txt 2>&1
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 >> $out/check_schedules.txt 2>&1
timeout 600 python tools/check_schedules.py --images 200 --uncalibrated >> $out/check_schedules.txt 2>&1; grep -c "identical: True" $out/check_schedules.txt; grep -c "identical: False" $out/check_schedules.txt; grep "hyp_pair_grid\|batched_check_build\|lo_prepare_wave" $out/check_schedules.txt
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > /dev/null 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
grep -E "k_prescore_compact2" $out/verify_kernel_stats_1lane.csv | cut -c1-120
