#!/bin/bash
# C3 fit A/B: two-level panel widths (0 = the one-level flagged execution) x look-ahead, then parity tests at C3 size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "8 0" "8 1" "4 1" "16 1"; do
set -- $cfg
GPMPC_TWOLEVEL=$1 GPMPC_LOOKAHEAD=$2 timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>gpurun_out/c3_err.log | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('W=$1 lookahead=$2  ms/step %.1f  factor %.2f ms (%.1f TF)  invK %.2f  rollout %s finite %s' % (d['ms_per_step'], d['phases_ms_per_step']['factor'], d['roofline']['achieved'], d['phases_ms_per_step'].get('invK',0), {k: round(v,1) for k,v in d['rollout_ms_per_call'].items()}, d['finite']))
else: print('W=$1 failed')"
grep -i "timed out\|error" gpurun_out/c3_err.log | head -3
done
timeout 600 python -m pytest tests -m gpu -x -q -k "c3 or c5_pattern_c3 or worker_path_odd" 2>&1 | tail -3
