#!/bin/bash
# end-of-round session: the whole GPU suite, smoke, the default bench line, rocprof kernel stats (default and one lane),
# the schedule check -- what profiles/r03_* and DESIGN.md section 6 quote
out=gpurun_out/final
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 5 --warmup 1 > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline_native']['value'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv
rm -rf $out/prof
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $R/$out/bench_under_rocprof_1lane.json 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
timeout 900 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python bench.py --uncalibrated --steps 3 --warmup 1 --cpu-seconds 0 > $out/bench_config2_uncalibrated.json 2>/dev/null
timeout 600 python bench.py --images 50 --feats 1024 --uncalibrated --steps 5 --warmup 1 --cpu-seconds 0 > $out/bench_config1.json 2>/dev/null
timeout 900 python tools/shard_sweep.py --shards 8 --steps 2 > $out/shard_sweep_config2.txt 2>&1; tail -2 $out/shard_sweep_config2.txt | cut -c1-200
python -c "
import json
for f in ('bench_config2_uncalibrated','bench_config1'):
    d=json.load(open('$out/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
