#!/bin/bash
# round 4, session A: suite on the refactored tree, new bench fields, forced RCCL exchange, verification counters,
# end-to-end CLI line, A/B of the exec-masked inlier count (ab/lib_base.so vs ab/lib_mask.so)
out=gpurun_out/r4a
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.err
python - <<PY
import json
d=json.load(open('$out/bench_default.json'))
print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline'].get('executed_frac'), d.get('exchange'), d.get('extra'))
PY
for mode in auto broadcast; do
  timeout 400 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime --force-collectives $mode > $out/bench_force_$mode.json 2> $out/bench_force_$mode.err
  python -c "
import json; d=json.load(open('$out/bench_force_$mode.json')); print('force $mode', round(d['value']), round(d['ms_per_step'],1), d['exchange'])" || tail -5 $out/bench_force_$mode.err
done
DSM_VERIFY_LANES=1 timeout 900 python tools/collect_pmc.py --verify --out $out/verify_pmc.json --steps 1 --warmup 0 --cpu-seconds 0 --no-second-regime > $out/verify_pmc_summary.json 2> $out/verify_pmc.err; tail -c 400 $out/verify_pmc.err; head -c 1500 $out/verify_pmc_summary.json; echo
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 50 > $out/bench_cli_500x4096_block50.txt 2>&1; tail -3 $out/bench_cli_500x4096_block50.txt
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 > $out/bench_cli_500x4096_block500.txt 2>&1; tail -3 $out/bench_cli_500x4096_block500.txt
# A/B: exec-masked inlier count
DSM_LIB_PATH=$R/ab/lib_mask.so timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py tests/test_camera_models.py tests/test_golden.py -m gpu -x -q > $out/pytest_mask.log 2>&1; tail -3 $out/pytest_mask.log
for v in base mask; do
  DSM_LIB_PATH=$R/ab/lib_$v.so timeout 400 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --dump-graph /tmp/g_$v.npz > $out/bench_$v.json 2>/dev/null
  (cd /tmp && DSM_LIB_PATH=$R/ab/lib_$v.so DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$v -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof_$v.err)
  find $out/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane_$v.csv
  rm -rf $out/prof_$v
done
python - <<PY
import numpy as np, json, csv
a=np.load('/tmp/g_base.npz'); b=np.load('/tmp/g_mask.npz')
print('graphs identical:', all((a[k].shape==b[k].shape and (a[k]==b[k]).all()) for k in a.files))
for v in ('base','mask'):
    d=json.load(open('$out/bench_%s.json'%v)); print(v, round(d['value']), d['kernel_ms_per_step']['k_verify_pairs'])
    rows={r['Name']:float(r['AverageNs'])*int(r['Calls'])/2e6 for r in csv.DictReader(open('$out/kernel_stats_1lane_%s.csv'%v))}
    print(v, {k:round(x,1) for k,x in rows.items() if 'k_score' in k or 'k_models_score' in k})
PY
