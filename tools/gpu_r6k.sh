#!/bin/bash
# round 6, session k: a shard with its list cut over 2 / 3 contexts (matching of one part beside the verification of another)
out=gpurun_out/${1:-r6k}
mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do
for c in 1 2 3; do
  echo -n "shard contexts $c: "
  timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 5 --warmup 2 --cpu-seconds 0 --contexts $c 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'], 'k1 %.1f' % k['k1_best_rows<pass 1>'])"
done
echo -n "shard contexts 2 no lock: "
timeout 300 python bench.py --shard-of 8 --shard-index 3 --steps 5 --warmup 2 --cpu-seconds 0 --contexts 2 --no-match-lock 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'], 'k1 %.1f' % k['k1_best_rows<pass 1>'])"
done | tee $out/contexts_shard.txt
