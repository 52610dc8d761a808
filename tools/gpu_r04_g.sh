#!/bin/bash
# r04: exp of the exact-moment pair sums through the 2^(j/2048) table -- parity, C3 A/B, PMC of em_pair_kernel (both exps)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "moment or wide or em or c3_rollout or gp_class or c2_full or cholesky" 2>&1 | tail -5
c3() {
  timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 C3 ms/step %.1f' % j['ms_per_step'], 'phases', {k: round(v,2) for k,v in j.get('phases_ms_per_step',{}).items()})"
}
GPMPC_EM_EXP_TAB=1 c3 "exp table "
GPMPC_EM_EXP_TAB=0 c3 "exp_lean  "
GPMPC_EM_EXP_TAB=1 c3 "exp table "
R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp
cat > /tmp/emv.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
p = go.synthetic_problem(8192, 8, 6, 4, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
h.fit(p['hyper'], want_invK=True); h.synchronize()
for _ in range(3): h.predict('EM', p['Z'][:1], p['Sigma'][:1])
h.synchronize()
PY
for TAB in 1 0; do
for C in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  GPMPC_EM_EXP_TAB=$TAB timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$R/gpurun_out/pmc_em${TAB}_$N" -o p -- python /tmp/emv.py > "$R/gpurun_out/pmc_em${TAB}_$N.log" 2>&1
  echo "== tab=$TAB $C rc=$?"
  python "$R/tools/pmc_summary.py" "$R/gpurun_out/pmc_em${TAB}_$N/p_results.db" > "$R/gpurun_out/r04_pmc_em_tab${TAB}_$N.txt" 2>&1
  grep -E -A5 "em_pair_kernel" "$R/gpurun_out/r04_pmc_em_tab${TAB}_$N.txt" | head -14
  rm -rf "$R/gpurun_out/pmc_em${TAB}_$N"
done
done
GPMPC_EM_EXP_TAB=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_em" -o t -- python /tmp/emv.py > "$R/gpurun_out/prof_em.log" 2>&1
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_em/t_results.db" | grep -i "em_" | head; rm -rf "$R/gpurun_out/prof_em"
GPMPC_EM_EXP_TAB=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_em" -o t -- python /tmp/emv.py > "$R/gpurun_out/prof_em.log" 2>&1
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_em/t_results.db" | grep -i "em_" | head; rm -rf "$R/gpurun_out/prof_em"
