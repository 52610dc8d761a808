#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for c in 2 3; do
GPMPC_VERBOSE=1 GPMPC_CHAIN=$c GPMPC_CHAIN_TRACE=gpurun_out/chain_trace_$c.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/trace_err_$c.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('chain=$c value %8.0f  ms/step %.3f  factor %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor']))"
grep gpmpc gpurun_out/trace_err_$c.log | sort | uniq -c | head -5
python tools/chain_trace.py gpurun_out/chain_trace_$c.bin 64
done
for rep in 1 2; do for c in 2 3; do
GPMPC_CHAIN=$c timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('chain=$c value %8.0f  ms/step %.3f  factor %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor']))"
done; done
