#!/bin/bash
# One GPU-box session: default bench line, rocprofv3 kernel stats of the same command, PMC passes, side configs.
# usage: tools/gpu_round.sh <tag>      (outputs under gpurun_out/<tag>/)
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err
tail -c 600 $out/bench_default.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python tools/collect_pmc.py --util > $out/pmc.json 2> $out/pmc.err
cp gpurun_out/pmc/k1_pmc.json $out/k1_pmc.json 2>/dev/null
python bench.py --gpus 2 --oversubscribe --steps 1 --warmup 1 --cpu-seconds 0 --images 120 > $out/bench_2rank_oversub.json 2> $out/bench_2rank_oversub.err
python bench.py --images 120 --steps 1 --warmup 1 --cpu-seconds 0 > $out/bench_1rank_120.json 2>/dev/null
python bench.py --images 50 --feats 1024 --uncalibrated --steps 5 --warmup 1 --cpu-seconds 0 > $out/bench_config1.json 2>/dev/null
python bench.py --uncalibrated --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench_config2_uncal.json 2>/dev/null
rm -rf $out/prof gpurun_out/pmc
head -c 1500 $out/bench_default.json; echo; head -12 $out/kernel_stats.csv | cut -c1-150; cat $out/bench_2rank_oversub.json | head -c 600; tail -3 $out/bench_2rank_oversub.err
