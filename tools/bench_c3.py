#!/usr/bin/env python3
"""Secondary benchmarks on one MI355X (BASELINE configs C3 and C5), printed as JSON lines.
C3: 6-output GP, N=8192, d=8: fit (+K^-1), 30-step ME / TA / EM uncertainty propagation (rollout).
C5: IPOPT-pattern driver: Nt=30 shooting nodes per call, value + mean Jacobian + TA covariance, 50 calls."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from gp_mpc_amd.synthetic import synthetic_problem
from gp_mpc_amd._lib import Handle, get_lib

N = int(os.environ.get('C3_N', 8192))
Ny, d, T = 6, 8, 30
lib = get_lib()
p = synthetic_problem(N, d, Ny, T, seed=1234, sn=1e-2)
h = Handle(lib, p['X'], p['Y'])
h.profile_enable(True)
t0 = time.perf_counter(); h.fit(p['hyper'], want_invK=True); h.synchronize(); t_fit = time.perf_counter() - t0
t0 = time.perf_counter(); h.fit(p['hyper'], want_invK=True); h.synchronize(); t_fit2 = time.perf_counter() - t0
prof = h.profile_read()
print(json.dumps({'bench': 'C3 fit', 'N': N, 'Ny': Ny, 'd': d, 'fit_s_first': t_fit, 'fit_s': t_fit2,
                  'phases_ms': {k: v[0] / 2 for k, v in prof.items() if v[1]}}))
Nu = d - Ny
x0 = p['Z'][0, :Ny]
U = p['Z'][:, Ny:]
S0 = np.eye(d) * 1e-6
S0[:Ny, :Ny] = np.diag(p['hyper'][:, d + 1] ** 2)
for method in ('ME', 'TA', 'EM'):
    h.profile_read()
    mean_t, S = x0.copy(), S0.copy()
    t0 = time.perf_counter()
    for t in range(T):
        z = np.concatenate([mean_t, U[t]])
        m, c = h.predict(method, z.reshape(1, d), S.reshape(1, d, d))
        mean_t = m[0]
        S[:Ny, :Ny] = c[0]
    dt = time.perf_counter() - t0
    prof = h.profile_read()
    print(json.dumps({'bench': f'C3 rollout {method}', 'steps': T, 'total_s': dt, 'ms_per_step': dt / T * 1e3,
                      'finite': bool(np.all(np.isfinite(mean_t)) and np.all(np.isfinite(S))),
                      'phases_ms_per_step': {k: v[0] / T for k, v in prof.items() if v[1]}}))
# the same propagation as ONE device call (gpmpc_rollout: no host round trip per step)
h.profile_enable(False)
for method in ('ME', 'TA', 'EM'):
    z0 = np.concatenate([x0, U[0]])
    h.rollout(method, z0, U[:T], S0)
    t0 = time.perf_counter()
    m, c = h.rollout(method, z0, U[:T], S0)
    dt = time.perf_counter() - t0
    print(json.dumps({'bench': f'C3 rollout {method}, one device call', 'steps': T, 'total_s': dt, 'ms_per_step': dt / T * 1e3,
                      'finite': bool(np.all(np.isfinite(m)) and np.all(np.isfinite(c)))}))
h.profile_enable(True)
# C5: 30 nodes per call, value + Jacobian + TA covariance
Z = p['Z'][:30]
Sg = p['Sigma'][:30]
h.predict('TA', Z, Sg)
h.profile_read()
t0 = time.perf_counter()
for it in range(50):
    m, c = h.predict('TA', Z, Sg)
    m2, J = h.mean_jac(Z)
dt = time.perf_counter() - t0
prof = h.profile_read()
print(json.dumps({'bench': 'C5 phases', 'phases_ms_per_call': {k: v[0] / 50 for k, v in prof.items() if v[1]}}))
print(json.dumps({'bench': 'C5 IPOPT-pattern (Nt=30, value+J+TA cov per call; predict + mean_jac)', 'calls': 50,
                  'ms_per_call': dt / 50 * 1e3, 'node_evals_per_s': 50 * 30 / dt}))
h.predict_jac('TA', Z, Sg)
t0 = time.perf_counter()
for it in range(50):
    m, c, J = h.predict_jac('TA', Z, Sg)
dt = time.perf_counter() - t0
print(json.dumps({'bench': 'C5 IPOPT-pattern (Nt=30, value+J+TA cov per call; one pass, gpmpc_predict_jac)', 'calls': 50,
                  'ms_per_call': dt / 50 * 1e3, 'node_evals_per_s': 50 * 30 / dt}))
h.close()
