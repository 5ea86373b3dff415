#!/bin/bash
out=gpurun_out/r3k
mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 900 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 900 python tools/shard_sweep.py --shards 8 --steps 2 > $out/shard_sweep_config2.txt 2>&1; tail -2 $out/shard_sweep_config2.txt | cut -c1-300
timeout 600 python bench.py --images 50 --feats 1024 --uncalibrated --steps 5 --warmup 1 --cpu-seconds 0 > $out/bench_config1.json 2>/dev/null; python -c "
import json; d=json.load(open('$out/bench_config1.json')); print('config1', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
