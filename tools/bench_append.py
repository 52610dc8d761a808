#!/usr/bin/env python3
"""gpmpc_append against a full refit: N0 = 4032 points + 64 new ones (N=4096, d=6, one output)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from gp_mpc_amd.synthetic import synthetic_problem
from gp_mpc_amd._lib import Handle, get_lib
lib = get_lib()
p = synthetic_problem(4096, 6, 1, 10, seed=1234, sn=1e-2)
X, Y, H = p['X'], p['Y'], p['hyper']
for n in (1, 16, 64, 256):
    N0 = 4096 - n
    ts = []
    for rep in range(3):
        h = Handle(lib, X[:N0], Y[:N0]); h.fit(H); h.synchronize()
        t0 = time.perf_counter(); h.append(X[N0:], Y[N0:]); h.synchronize(); ts.append(time.perf_counter() - t0)
        if rep == 2:
            f = h.get_factors(); m, v = h.predict_mean_var(p['Z'])
        h.close()
    h = Handle(lib, X, Y); h.fit(H); h.synchronize()
    t0 = time.perf_counter(); h.fit(H); h.synchronize(); t_fit = time.perf_counter() - t0
    f2 = h.get_factors(); m2, v2 = h.predict_mean_var(p['Z']); h.close()
    print(json.dumps({'bench': 'append', 'N0': N0, 'n': n, 'append_ms': min(ts) * 1e3, 'refit_ms': t_fit * 1e3,
                      'relF_L': float(np.linalg.norm(f['chol'] - f2['chol']) / np.linalg.norm(f2['chol'])),
                      'mean_maxabs': float(np.abs(m - m2).max()), 'var_maxabs': float(np.abs(v - v2).max())}))
