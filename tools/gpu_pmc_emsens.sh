#!/bin/bash
# PMC passes over gpmpc_predict_em_sens at C3 size: clock, matrix-pipe and VALU occupancy of em_pair_sens_kernel
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp; cd /tmp
cat > /tmp/ems.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
p = go.synthetic_problem(8192, 8, 6, 4, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
h.fit(p['hyper'], want_invK=True); h.synchronize()
for _ in range(3): h.predict_em_sens(p['Z'][:1], p['Sigma'][:1])
pass
h.synchronize()
PY
for C in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$R/gpurun_out/pmc_ems_$N" -o p -- python /tmp/ems.py > "$R/gpurun_out/pmc_ems_$N.log" 2>&1
  echo "== $C rc=$?"
  python "$R/tools/pmc_summary.py" "$R/gpurun_out/pmc_ems_$N/p_results.db" > "$R/gpurun_out/pmc_ems_$N.txt" 2>&1
  grep -E -A5 "em_pair_sens_kernel|em_pair_kernel<false>" "$R/gpurun_out/pmc_ems_$N.txt" | head -16
done
