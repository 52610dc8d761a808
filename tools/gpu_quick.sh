#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
timeout 120 python bench.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %8.0f  ms/step %.3f  phases %s parity %s' % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, d.get('parity_vs_cpu')))"
done
