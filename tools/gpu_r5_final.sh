#!/bin/bash
# round 5, end-of-round session: suite, smoke, default bench line (with CPU baselines), kernel stats (two lanes / one lane),
# PMC of K1 and of the verification kernels, shard sweep, the other BASELINE configurations
out=gpurun_out/r5final
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 5 --warmup 1 --dump-line $out/bench_default_long.json > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline_verify']['frac'], d['cpu_baseline']['value'], d['cpu_baseline_native']['value'], d['extra'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
rm -rf $out/prof
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 > /dev/null 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
timeout 900 python tools/collect_pmc.py --util --out $out/k1_pmc.json --steps 1 --warmup 0 --cpu-seconds 0 --no-second-regime --no-config3 > $out/k1_pmc_stdout.json 2> $out/k1_pmc.err; tail -c 300 $out/k1_pmc.err
DSM_VERIFY_LANES=1 timeout 900 python tools/collect_pmc.py --verify --out $out/verify_pmc.json --steps 1 --warmup 0 --cpu-seconds 0 --no-second-regime --no-config3 > $out/verify_pmc_summary.json 2> $out/verify_pmc.err; tail -c 200 $out/verify_pmc.err
rm -rf gpurun_out/pmc
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 --legacy >> $out/check_schedules.txt 2>&1
timeout 600 python tools/check_schedules.py --images 200 --uncalibrated --legacy >> $out/check_schedules.txt 2>&1; grep -c "identical: True" $out/check_schedules.txt; grep -c "identical: False" $out/check_schedules.txt
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 --modes "default,blocking,async+bulk_journal" > $out/bench_cli_500x4096.txt 2>&1
grep "pairs in" $out/bench_cli_500x4096.txt
timeout 900 python tools/shard_sweep.py --shards 8 --steps 2 > $out/shard_sweep_config2.txt 2>&1; tail -2 $out/shard_sweep_config2.txt | cut -c1-300
timeout 300 python bench.py --uncalibrated --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 > $out/bench_config2_uncalibrated.json 2>/dev/null
timeout 300 python bench.py --images 50 --feats 1024 --uncalibrated --steps 10 --warmup 2 --cpu-seconds 0 > $out/bench_config1.json 2>/dev/null
timeout 900 python bench.py --images 2000 --no-verify --steps 2 --warmup 1 --cpu-seconds 10 > $out/bench_config3_2000x4096_match_only.json 2> $out/config3.err
timeout 900 python bench.py --images 10000 --pairs knn:200 --shard-of 8 --shard-index 3 --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench_config4_shard4of8_10000img_knn200.json 2> $out/config4.err
timeout 1200 python bench.py --images 10000 --feats 8192 --pairs knn:200 --fixed-trials 4096 --shard-of 8 --shard-index 3 --steps 1 --warmup 1 --cpu-seconds 0 > $out/bench_config5_shard4of8_10000x8192_fixed4096.json 2> $out/config5.err
python - <<PY
import json
for f in ('bench_config2_uncalibrated','bench_config1','bench_config3_2000x4096_match_only','bench_config4_shard4of8_10000img_knn200','bench_config5_shard4of8_10000x8192_fixed4096'):
    try:
        d=json.load(open('$out/'+f+'.json')); print(f, round(d['value']), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d.get('hypotheses_per_s'))
    except Exception as e: print(f, 'ERR', e)
PY
