#!/bin/bash
# kernel trace of the less-travelled entry points at C3 size: EM with derivative outputs, predict_sens, append
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp; cd /tmp
cat > /tmp/misc.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
p = go.synthetic_problem(8192, 8, 6, 30, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
h.fit(p['hyper'], want_invK=True)
def timeit(name, fn, n=5):
    fn(); h.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h.synchronize(); print('%-40s %.3f ms per call' % (name, (time.perf_counter() - t0) / n * 1e3))
timeit('predict_em_sens B=1', lambda: h.predict_em_sens(p['Z'][:1], p['Sigma'][:1]))
timeit('predict_em_sens B=1, no cov value', lambda: h.predict_em_sens(p['Z'][:1], p['Sigma'][:1], want_cov=False))
timeit('predict EM B=1', lambda: h.predict('EM', p['Z'][:1], p['Sigma'][:1]))
timeit('predict_sens B=30', lambda: h.predict_sens(p['Z'][:30]))
timeit('predict_jac TA B=30', lambda: h.predict_jac('TA', p['Z'][:30], p['Sigma'][:30]))
timeit('predict old_TA B=30', lambda: h.predict('old_TA', p['Z'][:30], p['Sigma'][:30]))
q = go.synthetic_problem(64, 8, 6, 1, seed=5, sn=1e-2)
t0 = time.perf_counter(); h.append(q['X'], q['Y']); print('append +64 at N=8192, Ny=6: %.1f ms' % ((time.perf_counter() - t0) * 1e3), 'handoff_timeouts', h.counter('handoff_timeouts'))
for k in range(6):
    q = go.synthetic_problem(64, 8, 6, 1, seed=6 + k, sn=1e-2)
    t0 = time.perf_counter(); h.append(q['X'], q['Y']); print('append +64 again: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
PY
python /tmp/misc.py
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/misctrace" -o t -- python /tmp/misc.py > "$R/gpurun_out/misctrace.log" 2>&1
python "$R/tools/prof_summary.py" "$R/gpurun_out/misctrace/t_results.db" --steps 1 | head -28 | cut -c1-110
