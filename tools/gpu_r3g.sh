#!/bin/bash
# round 3: parity of everything since r3c + the measurement lines the verdict asked for
out=gpurun_out/r3g
mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q --durations=5 > $out/pytest.log 2>&1; tail -12 $out/pytest.log
# headline + 0.25-inlier-ratio line
timeout 600 python bench.py --steps 3 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout 900 python bench.py --images 150 --outlier-frac 0.5 --steps 2 --warmup 1 --cpu-seconds 10 > $out/bench_ratio025_150img.json 2> $out/bench_ratio025.err; python -c "
import json; d=json.load(open('$out/bench_ratio025_150img.json')); print('ratio 0.25', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
# shard sweeps: config 2, and the configs[3] list shape on 2 000 images
timeout 900 python tools/shard_sweep.py --shards 8 --steps 2 > $out/shard_sweep_config2.txt 2>&1; tail -3 $out/shard_sweep_config2.txt | cut -c1-400
timeout 900 python tools/shard_sweep.py --images 2000 --pairs knn:200 --shards 8 --steps 1 > $out/shard_sweep_knn200_2000img.txt 2>&1; tail -3 $out/shard_sweep_knn200_2000img.txt | cut -c1-400
