#!/bin/bash
# C3 fit A/B: two-level panel widths (0 = the one-level flagged execution), then the C3 bench line and the C2 line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for w in 0 4 8 16; do
GPMPC_TWOLEVEL=$w timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>gpurun_out/c3_err_$w.log | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('W=$w  ms/step %.1f  factor %.2f ms (%.1f TF)  invK %.2f  rollout %s finite %s' % (d['ms_per_step'], d['phases_ms_per_step']['factor'], d['roofline']['achieved'], d['phases_ms_per_step'].get('invK',0), {k: round(v,1) for k,v in d['rollout_ms_per_call'].items()}, d['finite']))
else: print('W=$w failed')"
tail -2 gpurun_out/c3_err_$w.log
done
timeout 300 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-2500
