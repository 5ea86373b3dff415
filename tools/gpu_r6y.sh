#!/bin/bash
# round 6, session y: config 5's shard got slower (16.1 k vs 20.5 k pairs/s in round 5) -- which change?  A truncated shard, check build
out=gpurun_out/${1:-r6y}
mkdir -p $out
export TMPDIR=/tmp
run() { python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'])"; }
A="--images 10000 --feats 8192 --pairs knn:200 --fixed-trials 4096 --shard-of 8 --shard-index 3 --max-pairs 12000 --steps 1 --warmup 1 --cpu-seconds 0"
for v in product check ef_f64 replay_legacy lanes1; do
  unset DSM_SCORE_PREFILTER DSM_REPLAY_LEGACY DSM_LIBRARY DSM_LIB_PATH DSM_VERIFY_LANES
  if [ $v != product ]; then export DSM_LIBRARY=check DSM_LIB_PATH=$PWD/dagsfm_amd/libdagsfm_mi355x_check.so; fi
  if [ $v = ef_f64 ]; then export DSM_SCORE_PREFILTER=33; fi
  if [ $v = replay_legacy ]; then export DSM_REPLAY_LEGACY=1; fi
  if [ $v = lanes1 ]; then export DSM_VERIFY_LANES=1; fi
  echo -n "$v: "; timeout 600 python bench.py $A 2>/dev/null | grep "^{" | tail -1 | run
done | tee $out/config5_ab.txt
unset DSM_SCORE_PREFILTER DSM_REPLAY_LEGACY DSM_LIBRARY DSM_LIB_PATH DSM_VERIFY_LANES
(cd /tmp && DSM_VERIFY_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof1 -o t -- python $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $GRAFT_REPO_ROOT/$out/err1.txt)
find $out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/config5_kernel_stats_1lane.csv
rm -rf $out/prof1
head -22 $out/config5_kernel_stats_1lane.csv | cut -c1-110
