#!/bin/bash
# round 4, session E: quotients with a shared denominator (shared_divisor / div_shared) in the register solvers
out=gpurun_out/r4e
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 1200 python tools/check_schedules.py --legacy > $out/check_schedules.txt 2>&1; cat $out/check_schedules.txt
timeout 600 python tools/check_schedules.py --images 150 --outlier-frac 0.5 --legacy > $out/check_schedules_ratio025.txt 2>&1; cat $out/check_schedules_ratio025.txt
timeout 600 python tools/fuzz_verify.py --batches 4 --pairs 2000 --seed 101 > $out/fuzz_verify.txt 2>&1; tail -1 $out/fuzz_verify.txt
for rep in 1 2; do for v in base sdiv; do
  echo -n "$v: "; DSM_LIB_PATH=$R/ab/lib_$v.so timeout 400 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), d['kernel_ms_per_step']['k_verify_pairs'])"
done; done | tee $out/ab_shared_divisor.txt
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime > /dev/null 2> $R/$out/rocprof.err)
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_1lane.csv
rm -rf $out/prof
grep "k_solve\|k_roots_e\|k_lo_prepare_reg\|k_solve_e_build" $out/kernel_stats_1lane.csv | cut -c1-110
