#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for cfg in "GPMPC_PAD_MIN=100000" "GPMPC_PAD_MIN=16" "GPMPC_PAD_MIN=1"; do
  env $cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-24s value %8.0f  ms/step %.3f  factor %.3f  vargemm %.3f' % ('$cfg', d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))"
done; done
