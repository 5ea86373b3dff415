#!/bin/bash
# round 4, session G: the sampler's run loop with the plain mask assembled without sign extension; whole-chip grids for short lists
out=gpurun_out/r4g
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 --legacy > $out/check_schedules_ratio025.txt 2>&1; cat $out/check_schedules_ratio025.txt
timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['extra'])"
timeout 900 python tools/shard_sweep.py --shards 8 --steps 2 > $out/shard_sweep_config2.txt 2>&1; tail -2 $out/shard_sweep_config2.txt | cut -c1-400
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 > $out/bench_cli_500.txt 2>&1; grep "pairs in\|async" $out/bench_cli_500.txt
timeout 300 python bench.py --images 50 --feats 1024 --uncalibrated --steps 10 --warmup 2 --cpu-seconds 0 > $out/bench_config1.json 2>/dev/null; python -c "
import json; d=json.load(open('$out/bench_config1.json')); print('config1', round(d['value']), round(d['ms_per_step'],2), d['kernel_ms_per_step'])"
