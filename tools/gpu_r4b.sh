#!/bin/bash
# round 4, session B: suite (FLANN word-search plumbing, bulk journal), K1 LDS-DMA A/B, shard / config-1 kernel traces
# (where does a short list's verification time go: host round trips or under-filled kernels?), end-to-end CLI with stage
# timers, SQLite's own insert ceiling on the box's file system
out=gpurun_out/r4b
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
DSM_LIB_PATH=$R/ab/lib_dma.so timeout 600 python -m pytest tests/test_match_gpu.py tests/test_golden.py tests/test_parity_fullsize_gpu.py -m gpu -x -q > $out/pytest_dma.log 2>&1; tail -3 $out/pytest_dma.log
for rep in 1 2; do
  for v in base dma; do
    echo -n "$v: "
    DSM_LIB_PATH=$R/ab/lib_$v.so timeout 300 python bench.py --steps 4 --warmup 1 --no-verify --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_ms_per_step'], d['roofline']['frac'])"
  done
done | tee $out/ab_k1_lds_dma.txt
for cfg in "shard8 --shard-of 8 --shard-index 3" "config1 --images 50 --feats 1024 --uncalibrated"; do
  set -- $cfg; tag=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace_$tag -o t -- python $R/bench.py $@ --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime > $R/$out/bench_trace_$tag.json 2> $R/$out/trace_$tag.err)
  f=$(find $out/trace_$tag -name "*kernel_trace.csv" | head -1)
  python tools/trace_summary.py $f > $out/trace_summary_$tag.txt 2>&1; head -4 $out/trace_summary_$tag.txt
  python tools/trace_gaps.py $f > $out/trace_gaps_$tag.txt 2>&1; head -12 $out/trace_gaps_$tag.txt
  rm -rf $out/trace_$tag
  timeout 300 python bench.py $@ --steps 5 --warmup 1 --cpu-seconds 0 --no-second-regime > $out/bench_$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('$out/bench_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],2), d['kernel_ms_per_step'])"
done
for bs in 500 125; do
  timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size $bs > $out/bench_cli_500x4096_block$bs.txt 2>&1; cat $out/bench_cli_500x4096_block$bs.txt | cut -c1-260
done
timeout 600 python tools/sqlite_ceiling.py > $out/sqlite_ceiling.txt 2>&1; cat $out/sqlite_ceiling.txt
