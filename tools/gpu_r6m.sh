#!/bin/bash
# round 6, session m: dsm_ctx_set_memory_budget -- the GPU test, and the step time of config 2 under 4 / 8 / 16 / 38 GiB and the default
out=gpurun_out/${1:-r6m}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_memory_budget_gpu.py tests/test_capi_symbols.py -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for b in 0 38 16 8 4; do
  echo -n "budget $b GiB: "
  timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config3 --no-second-regime --no-extra-configs --memory-budget-gib $b 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],1), 'verify %.1f' % k['k_verify_pairs'], d.get('memory'))"
done | tee $out/memory_budget.txt
