#!/bin/bash
# round 6, session x: counters of every verification kernel (one lane) with the current kernels
out=gpurun_out/${1:-r6x}
mkdir -p $out
export TMPDIR=/tmp
DSM_VERIFY_LANES=1 timeout 900 python tools/collect_pmc.py --verify --out $out/verify_pmc.json --steps 1 --warmup 0 --cpu-seconds 0 --no-second-regime --no-config3 --no-extra-configs > $out/verify_pmc_summary.json 2> $out/verify_pmc.err; tail -c 200 $out/verify_pmc.err
rm -rf gpurun_out/pmc
python - <<PY
import json
d=json.load(open("$out/verify_pmc.json"))
s=d['summary']
print({k:v for k,v in s.items() if k not in ('kernels','note','peaks')})
for k,v in sorted(s['kernels'].items(), key=lambda kv:-kv[1].get('ms_per_step',0))[:40]:
    print("%-34s ms %6.2f disp %4d issue %.3f f64frac %.2f lane %.2f wait_inst %.2f clk %.2f" % (k[:34], v['ms_per_step'], v['dispatches'], v['valu_issue_util'], v['executed_frac'], v['lane_util'] or 0, v.get('wait_inst_any_share') or 0, v['clock_ghz_while_busy'] or 0))
PY
