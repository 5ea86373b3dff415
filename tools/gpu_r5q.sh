#!/bin/bash
# round 5: why is the verification 12 % slower with an RCCL process group in the process?  Kernel durations (one lane) with and without it.
out=gpurun_out/r5q
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in plain rccl; do
  flag=""; [ $mode = rccl ] && flag="--force-collectives"
  (cd /tmp && DSM_VERIFY_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$mode -o bench -- python $R/bench.py $flag --steps 3 --warmup 1 --cpu-seconds 0 --no-second-regime --no-config3 > $R/$out/bench_$mode.json 2> $R/$out/rocprof_$mode.err)
  find $out/prof_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_$mode.csv
  rm -rf $out/prof_$mode
  grep "^{" $out/bench_$mode.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$mode', round(d['ms_per_step'],2), 'verify %.2f' % k['k_verify_pairs'])"
done
python - <<'P'
import csv
def load(f):
    return {r['Name']:(int(r['Calls']), float(r['TotalDurationNs'])/1e6) for r in csv.DictReader(open(f))}
a=load('gpurun_out/r5q/kernel_stats_plain.csv'); b=load('gpurun_out/r5q/kernel_stats_rccl.csv')
ta=sum(v[1] for k,v in a.items() if k.startswith(('k_','void k_'))); tb=sum(v[1] for k,v in b.items() if k.startswith(('k_','void k_')))
print('sum of dsm kernels ms (4 steps incl warm-up): plain %.1f rccl %.1f'%(ta,tb))
rows=sorted(((b.get(k,(0,0))[1]-v[1],k,v,b.get(k)) for k,v in a.items()), reverse=True)[:12]
for d,k,va,vb in rows: print('%+8.2f ms  %-60s plain %s rccl %s'%(d,k[:60],va,vb))
extra=[k for k in b if k not in a]
print('kernels only with rccl:', [(k[:70], b[k]) for k in extra][:10])
P
