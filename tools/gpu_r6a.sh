#!/bin/bash
# round 6, session a: today's baseline on the box + a per-dispatch kernel trace of the verification (one lane and two) so that the
# replay chain's launches can be read one by one (where the 39 ms of k_replay_lo go)
out=gpurun_out/r6a
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime > $out/bench_base.json 2> $out/bench_base.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6a/bench_base.json").read().strip().splitlines()[-1])
print("baseline", round(d["value"]), round(d["ms_per_step"], 1), d["kernel_ms_per_step"])
PY
for lanes in 1 2; do
  (cd /tmp && DSM_VERIFY_LANES=$lanes timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof$lanes -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime > $GRAFT_REPO_ROOT/$out/bench_trace$lanes.json 2> $GRAFT_REPO_ROOT/$out/err$lanes.txt)
  find $out/prof$lanes -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace_lanes$lanes.csv
  rm -rf $out/prof$lanes
  python3 tools/trace_dispatches.py $out/kernel_trace_lanes$lanes.csv k_replay_lo k_lo_ k_items > $out/dispatches_lanes$lanes.txt
  tail -12 $out/dispatches_lanes$lanes.txt
  python3 tools/trace_summary.py $out/kernel_trace_lanes$lanes.csv > $out/summary_lanes$lanes.txt
  head -3 $out/summary_lanes$lanes.txt
  # the traces themselves are large: keep the text views only
  gzip -f $out/kernel_trace_lanes$lanes.csv
  ls -la $out/kernel_trace_lanes$lanes.csv.gz
done
