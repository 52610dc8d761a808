#!/bin/bash
# C3 A/B of one switch: usage tools/gpu_c3.sh ENVVAR "v1 v2 ..." [pytest -k expr]; then C3-size parity tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
VAR=${1:-GPMPC_TWOLEVEL}; VALS=${2:-"8"}
for v in $VALS; do
env $VAR=$v timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>gpurun_out/c3_err.log | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$VAR=$v  ms/step %.1f  factor %.2f ms (%.1f TF)  invK %.2f  rollout %s finite %s' % (d['ms_per_step'], d['phases_ms_per_step']['factor'], d['roofline']['achieved'], d['phases_ms_per_step'].get('invK',0), {k: round(v,1) for k,v in d['rollout_ms_per_call'].items()}, d['finite']))
else: print('$VAR=$v failed')"
grep -i "timed out\|error" gpurun_out/c3_err.log | head -3
done
timeout 600 python -m pytest tests -m gpu -x -q -k "${3:-c3 or c5_pattern_c3}" 2>&1 | tail -3
