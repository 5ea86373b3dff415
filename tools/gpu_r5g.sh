#!/bin/bash
# round 5 checkpoint: the whole GPU suite, smoke, the FLANN search rates, the CLI end to end with the new defaults
out=gpurun_out/r5g
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python tools/bench_flann_search.py > $out/flann_search.json 2> $out/flann_search.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r5g/flann_search.json'))
for n,r in d['indices'].items():
    for k in ('k1','k5'):
        x=r[k]; print(n,k,'host 1t %.0f all %.0f  device %.0f /s (%.1f ms)  x%.1f identical %s'%(x['host_1_thread_searches_per_s'],x['host_all_threads_searches_per_s'],x['device_searches_per_s_kernel'],x['device_kernel_ms'],x['device_over_host_all_threads'],x['ids_and_distances_identical_on_the_host_sample']))
P
timeout 900 python tools/bench_cli.py --modes "blocking,async,async+bulk_journal" > $out/bench_cli.txt 2>&1; tail -8 $out/bench_cli.txt | cut -c1-300
