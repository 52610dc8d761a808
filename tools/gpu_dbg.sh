#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for sp in 1 2 3 4; do
GPMPC_VERBOSE=1 GPMPC_CHAIN=3 GPMPC_CHAIN_TRACE=gpurun_out/chain_trace_3.bin timeout 120 python bench.py --steps 30 --warmup 2 --no-cpu-baseline 2>gpurun_out/dbg_err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %8.0f  ms/step %.3f  factor %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor']))"
grep -v amdgpu.ids gpurun_out/dbg_err.log | grep -A9 "start times" | tail -2; grep "timed" gpurun_out/dbg_err.log; grep -c "tile-owner" gpurun_out/dbg_err.log
done
python tools/chain_trace.py gpurun_out/chain_trace_3.bin 64
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
