#!/bin/bash
# A/B of stream configurations for the look-ahead factorisation
cd "$GRAFT_REPO_ROOT"
for cfg in "GPMPC_SINGLE_STREAM=1" "GPMPC_SIDE_MASK_MOD=1 GPMPC_PRIORITY=0" "GPMPC_SIDE_MASK_MOD=1" "GPMPC_SIDE_MASK_MOD=8" "GPMPC_SIDE_MASK_MOD=4" "GPMPC_SIDE_MASK_MOD=16" "GPMPC_SIDE_MASK_MOD=2"; do
  env $cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-40s value %8.0f  ms/step %.3f  factor %.3f  vargemm %.3f' % ('$cfg', d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))"
done
