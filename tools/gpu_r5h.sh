#!/bin/bash
# round 5: the default bench line with the config-3 side measurement; the config-2 CLI with no option at all (the new defaults)
out=gpurun_out/r5h
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py --dump-line $out/bench_default_long.json > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json; python - <<'P'
import json
d=json.loads(open('gpurun_out/r5h/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])
print(d['extra'])
print(d['cpu_baseline'])
P
tail -3 $out/bench_default.err
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 --modes "default,blocking,async+bulk_journal,default on_device" > $out/bench_cli_500x4096.txt 2>&1; grep "pairs in" $out/bench_cli_500x4096.txt
