#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for nw in 64 96 128 160 192; do
GPMPC_NW2=$nw timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/q_err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('NW2=$nw value %8.0f  ms/step %.3f  factor %.3f vargemm %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))"
grep -c "timed" gpurun_out/q_err.log
done
