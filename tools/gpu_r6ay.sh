#!/bin/bash
# round 6, session ay: kernel breakdown of the 0.25-inlier-ratio regime (150 images), one lane
out=gpurun_out/${1:-r6ay}
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && DSM_VERIFY_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o bench -- python $R/bench.py --images 150 --outlier-frac 0.5 --steps 2 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > $R/$out/bench.json 2> $R/$out/rocprof1.err)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/low_inlier_kernel_stats_1lane.csv
rm -rf $out/prof1
head -30 $out/low_inlier_kernel_stats_1lane.csv | cut -c1-120
