#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %8.0f  ms/step %.3f  factor %.3f vargemm %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))"
done
GPMPC_CHAIN_TRACE=gpurun_out/chain_trace_5.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/chain_trace_5.bin 64 | tail -4
