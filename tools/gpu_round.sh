#!/bin/bash
# full check of a build: GPU parity tests, smoke, A/B of the factorisation modes, chain trace, full bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for rep in 1 2; do for c in 0 2 3; do
GPMPC_VERBOSE=1 GPMPC_CHAIN=$c timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/round_err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('chain=$c value %8.0f  ms/step %.3f  factor %.3f vargemm %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))"
grep "timed" gpurun_out/round_err.log | head -2
done; done
GPMPC_CHAIN_TRACE=gpurun_out/chain_trace_3.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/chain_trace_3.bin 64
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 3000 gpurun_out/bench_full.json
