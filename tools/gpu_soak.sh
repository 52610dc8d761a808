#!/bin/bash
# soak: thousands of fits through the persistent-kernel path; any hand-off time-out prints a message
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python bench.py --steps 2000 --warmup 3 --no-cpu-baseline --no-secondary 2>gpurun_out/soak_err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('soak: value %8.0f  ms/step %.3f  factor %.3f vargemm %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))"
echo "time-outs: $(grep -c 'timed out' gpurun_out/soak_err.log)"
