#!/bin/bash
# round 4, session I: the CLI's stage timers (match / verify / fetch inside the device stage), sliced vs whole-list device calls
out=gpurun_out/r4i
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python tools/bench_cli.py --images 500 --feats 4096 --block_size 500 --modes "blocking+bulk_journal,async+bulk_journal,async+bulk_journal unsliced,blocking+bulk_journal,async+bulk_journal,async+bulk_journal unsliced" > $out/bench_cli_ab2.txt 2>&1; cut -c1-360 $out/bench_cli_ab2.txt
