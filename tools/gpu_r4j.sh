#!/bin/bash
# round 4, session J: y / |y| as a sign and the two quotients by sqrt(1 + u^2) through one reciprocal in the 2 x 2 Jacobi step;
# the replay grids' size as a knob
out=gpurun_out/r4j
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 211 > $out/fuzz_verify.txt 2>&1; tail -2 $out/fuzz_verify.txt
timeout 600 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 > $out/bench_default.json 2> $out/bench_default.err; python -c "
import json; d=json.load(open('$out/bench_default.json')); print('default', round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step'], d['extra'])"
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
grep "k_final_pose\|k_lo_jacobi_reg\|k_sample\|k_replay_lo" $out/verify_kernel_stats_1lane.csv | cut -c1-120
timeout 900 python tools/exp_verify_knobs.py --combos "DSM_VERIFY_REPLAY_GRID=2;DSM_VERIFY_REPLAY_GRID=4" > $out/knobs.txt 2>&1; cat $out/knobs.txt
