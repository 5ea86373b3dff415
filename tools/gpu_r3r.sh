#!/bin/bash
out=gpurun_out/r3r
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python tools/collect_pmc.py --out $out/k1_pmc_2000img.json --images 2000 --no-verify --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $out/pmc1.err
timeout 600 python tools/collect_pmc.py --out $out/k1_pmc_150img.json --images 150 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $out/pmc2.err
timeout 600 python tools/collect_pmc.py --util1 --out $out/k1_pmc_config2.json --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2> $out/pmc3.err
python -c "
import json
for f in ('k1_pmc_2000img','k1_pmc_150img','k1_pmc_config2'):
    d=json.load(open('$out/'+f+'.json')); print(f, d.get('images'), d.get('pairs'), d.get('k1_traffic_bytes_per_launch'), d.get('algorithmic_bytes_per_launch'))
"
