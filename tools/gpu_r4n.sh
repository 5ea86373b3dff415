#!/bin/bash
# round 4, session N: where the 0.25-inlier-ratio workload (150 images, 11 175 pairs) spends its step, one lane
out=gpurun_out/r4n
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --images 150 --outlier-frac 0.5 --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > $R/$out/bench_ratio025_1lane.json 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_ratio025_1lane.csv
rm -rf $out/prof
head -32 $out/kernel_stats_ratio025_1lane.csv | cut -c1-120
python -c "
import json; d=json.load(open('$out/bench_ratio025_1lane.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config'].get('hypotheses_per_step'))"
