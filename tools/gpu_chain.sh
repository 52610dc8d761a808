#!/bin/bash
# chained vs single-queue factorisation: parity tests first, then an A/B of the bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do
for cfg in "GPMPC_CHAIN=0" "GPMPC_CHAIN=1" "GPMPC_CHAIN=2"; do
  env $cfg timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/chain_err_$cfg.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-24s value %8.0f  ms/step %.3f  factor %.3f  vargemm %.3f' % ('$cfg', d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))"
  tail -2 gpurun_out/chain_err_$cfg.log
done; done
for c in 1 2; do GPMPC_CHAIN=$c timeout 300 python tools/bench_c3.py 2>&1 | grep "C3 fit"; done; timeout 300 python tools/bench_c3.py 2>&1 | tail -8
