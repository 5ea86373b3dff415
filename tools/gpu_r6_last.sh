#!/bin/bash
# round 6, last check of the tree as committed: the suite, smoke, the driver's command
out=gpurun_out/r6last
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-line $out/bench_default_long.json > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_time.txt; echo "bench rc=$?"; tail -3 $out/bench_time.txt; wc -c $out/bench_default.json
python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_verify']['frac'])
print('parity_sample', d['parity_sample'])
for k, v in d['extra'].items(): print(k, {a: b for a, b in v.items() if a != 'workload'})
print(d.get('from_profiles'))
PY
# the N > 1 line as the driver will see it (two ranks sharing this one GPU over gloo: the format and per_rank, not a scaling number)
timeout 900 python bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > $out/bench_two_ranks_one_gpu.json 2> $out/two_ranks.err; tail -c 1500 $out/bench_two_ranks_one_gpu.json
