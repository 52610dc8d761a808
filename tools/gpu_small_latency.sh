#!/bin/bash
# per-call latency at the reference's own model sizes (car: N = 200, d = 5, Ny = 3; tank: N = 60, d = 6, Ny = 4), host pointers
cd "$GRAFT_REPO_ROOT"
cat > /tmp/lat.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
def timeit(name, fn, n=200):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print('  %-34s %8.1f us per call' % (name, (time.perf_counter() - t0) / n * 1e6))
for (N, d, Ny) in ((200, 5, 3), (60, 6, 4), (1000, 5, 3)):
    p = go.synthetic_problem(N, d, Ny, 30, seed=3, sn=1e-2)
    h = Handle(get_lib(), p['X'], p['Y'])
    h.fit(p['hyper'], want_invK=True)
    print('N=%d d=%d Ny=%d' % (N, d, Ny))
    timeit('fit', lambda: h.fit(p['hyper'], want_invK=True), 50)
    for B in (1, 30):
        Z, S = p['Z'][:B], p['Sigma'][:B]
        timeit('predict ME B=%d' % B, lambda: h.predict('ME', Z, S))
        timeit('predict TA B=%d' % B, lambda: h.predict('TA', Z, S))
        timeit('predict EM B=%d' % B, lambda: h.predict('EM', Z, S))
        timeit('predict_sens B=%d' % B, lambda: h.predict_sens(Z))
        timeit('predict_em_sens B=%d' % B, lambda: h.predict_em_sens(Z, S, want_cov=False), 50)
    U = np.tile(p['Z'][0, Ny:], (30, 1))
    S0 = np.eye(d) * 1e-6
    for m in ('ME', 'TA', 'EM'):
        timeit('rollout %s, 30 steps' % m, lambda: h.rollout(m, p['Z'][0], U, S0), 30)
    h.close()
PY
python /tmp/lat.py
