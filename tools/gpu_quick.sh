#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for cfg in "GPMPC_VAR_ORDER=0" "GPMPC_VAR_ORDER=2"; do
env $cfg timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$cfg value %8.0f  ms/step %.3f  factor %.3f vargemm %.3f (%.1f TF) parity %s' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm'], d['roofline']['achieved'], d.get('parity_vs_cpu')))"
done; done
