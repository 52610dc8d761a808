#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
cat > /tmp/c3fit.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
p = go.synthetic_problem(8192, 8, 6, 4, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
for i in range(3):
    t0 = time.perf_counter(); h.fit(p['hyper'], want_invK=False); h.synchronize(); print('fit', (time.perf_counter() - t0) * 1e3, 'ms')
PY
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_c3" -o t -- python /tmp/c3fit.py > "$R/gpurun_out/prof_c3.log" 2>&1; echo "rocprof rc=$?"
grep fit "$R/gpurun_out/prof_c3.log"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_c3/t_results.db" --steps 3 > "$R/gpurun_out/prof_c3.txt"; head -40 "$R/gpurun_out/prof_c3.txt"
