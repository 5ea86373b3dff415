#!/bin/bash
# round 6, session i: shard sweep of config 2 with both cuts (contiguous / interleaved) + the two-rank-on-one-GPU test
out=gpurun_out/${1:-r6i}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multi_rank_gpu.py tests/test_rccl_single_rank_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 1500 python tools/shard_sweep.py --images 500 --feats 4096 --shards 8 --steps 3 > $out/shard_sweep_config2.txt 2> $out/shard_sweep.err
grep -v "^{" $out/shard_sweep_config2.txt
