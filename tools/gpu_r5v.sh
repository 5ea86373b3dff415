#!/bin/bash
# round 5: the H bound step with a packed-f32 first stage (k_prescore_h2): parity, bounds on every slot, A/B against round 4's FP64 form, kernel stats
out=gpurun_out/r5v
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_parity_fullsize_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log | cut -c1-200
timeout 600 python tools/check_score_bounds.py > $out/score_bounds.txt 2>&1; tail -1 $out/score_bounds.txt | cut -c1-300
timeout 600 python tools/check_score_bounds.py --images 150 --outlier-frac 0.5 >> $out/score_bounds.txt 2>&1; tail -1 $out/score_bounds.txt | cut -c1-300
timeout 600 python tools/check_score_bounds.py --images 200 --uncalibrated >> $out/score_bounds.txt 2>&1; tail -1 $out/score_bounds.txt | cut -c1-300
for rep in 1 2; do bash tools/gpu_ab_verify.sh r5v_ab$rep base h2; done
(cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 > /dev/null 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/verify_kernel_stats_1lane.csv
rm -rf $out/prof1
grep "prescore\|score_needed" $out/verify_kernel_stats_1lane.csv | cut -c1-120
