#!/bin/bash
# PMC passes over one C3 fit with K^-1 (6 x 8192^2): L2 hits / misses and fetched bytes per kernel
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp; cd /tmp
cat > /tmp/c3fit.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
p = go.synthetic_problem(8192, 8, 6, 4, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
h.fit(p['hyper'], want_invK=True); h.synchronize()
PY
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  GPMPC_CHAIN=0 timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$R/gpurun_out/pmc_c3_$N" -o p -- python /tmp/c3fit.py > "$R/gpurun_out/pmc_c3_$N.log" 2>&1
  echo "== $C rc=$?"
  python "$R/tools/pmc_summary.py" "$R/gpurun_out/pmc_c3_$N/p_results.db" > "$R/gpurun_out/pmc_c3_$N.txt" 2>&1
  grep -A4 "gemm_f64_dma_kernel<128, 128, 2, 4, 2, 4, true, true" "$R/gpurun_out/pmc_c3_$N.txt" | head -8
done
