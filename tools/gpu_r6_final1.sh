#!/bin/bash
# round 6, end-of-round session 1: suite, smoke, the driver's own command (CPU baselines, parity samples, side measurements), kernel
# stats (two lanes / one lane), PMC of K1 and of the verification kernels (stamped with the commit and the source hashes), schedules
out=gpurun_out/r6final
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python tools/collect_pmc.py --util --out $out/k1_pmc.json --steps 1 --warmup 0 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > $out/k1_pmc_stdout.json 2> $out/k1_pmc.err; tail -c 300 $out/k1_pmc.err
DSM_VERIFY_LANES=1 timeout 900 python tools/collect_pmc.py --verify --out $out/verify_pmc.json --steps 1 --warmup 0 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > $out/verify_pmc_summary.json 2> $out/verify_pmc.err; tail -c 200 $out/verify_pmc.err
rm -rf gpurun_out/pmc
# the collections go where bench.py looks for them BEFORE the driver's command runs: its line then carries fresh from_profiles figures
cp $out/k1_pmc.json profiles/r06_k1_pmc.json; cp $out/verify_pmc.json profiles/r06_verify_pmc.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-line $out/bench_default_long.json > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_time.txt; echo "bench rc=$?"; tail -3 $out/bench_time.txt; wc -c $out/bench_default.json
python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_verify']['frac'])
print('parity_sample', d['parity_sample'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline_native']['value'])
for k, v in d['extra'].items(): print(k, {a: b for a, b in v.items() if a != 'workload'})
print(d.get('from_profiles'))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
rm -rf $out/prof
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > /dev/null 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
head -8 $out/bench_kernel_stats.csv | cut -c1-110
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 --legacy >> $out/check_schedules.txt 2>&1
timeout 600 python tools/check_schedules.py --images 200 --uncalibrated --legacy >> $out/check_schedules.txt 2>&1
echo "# planar scene, 200 images (H is the model: k_lo_prepare_reg<H, 3>), calibrated / uncalibrated" >> $out/check_schedules.txt
timeout 600 python tools/check_schedules.py --images 200 --planar >> $out/check_schedules.txt 2>&1
timeout 600 python tools/check_schedules.py --images 200 --planar --uncalibrated >> $out/check_schedules.txt 2>&1; grep -c "identical: True" $out/check_schedules.txt; grep -c "identical: False" $out/check_schedules.txt
( timeout 600 python tools/check_score_bounds.py; timeout 600 python tools/check_score_bounds.py --images 150 --outlier-frac 0.5; timeout 600 python tools/check_score_bounds.py --images 200 --uncalibrated ) 2>&1 | grep -v amdgpu.ids | tee $out/score_bounds_check.txt
