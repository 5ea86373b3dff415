#!/bin/bash
# round 4, session Q: the default bench line and the one-lane kernel stats of the last build
out=gpurun_out/r4q
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 5 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline_verify']['frac'], d['cpu_baseline']['value'], d['extra']['low_inlier_regime']['pairs_per_s'])"
(cd /tmp && DSM_VERIFY_LANES=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
grep "k_prescore\|k_score_needed" $out/verify_kernel_stats_1lane.csv | cut -c1-100
