#!/usr/bin/env python3
"""C4 on one GPU: cost of one NLL (+ analytic gradient) evaluation at N=4096, d=6 and a short sharded
multistart run (restarts x optimiser iterations); with torch.distributed initialised the same script
shards the restarts over the ranks (gp_mpc_amd/train.py)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from gp_mpc_amd.synthetic import synthetic_problem
from gp_mpc_amd._lib import Handle, get_lib
from gp_mpc_amd.train import train_gp
N, d = int(os.environ.get('C4_N', 4096)), 6
p = synthetic_problem(N, d, 1, 8, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
hp = p['hyper'][0].copy()
h.nll(0, hp, want_grad=True)
h.profile_enable(True); h.profile_read()
t0 = time.perf_counter()
for i in range(10): v = h.nll(0, hp * (1 + 0.01 * i))
t_val = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
for i in range(10): v, g = h.nll(0, hp * (1 + 0.01 * i), want_grad=True)
t_grad = (time.perf_counter() - t0) / 10
prof = h.profile_read()
print(json.dumps({'bench': 'C4 NLL evaluation', 'N': N, 'd': d, 'nll_ms': t_val * 1e3, 'nll_grad_ms': t_grad * 1e3,
                  'phases_ms_per_eval': {k: v[0] / 20 for k, v in prof.items() if v[1]}}))
t0 = time.perf_counter()
opt = train_gp(h, p['X'], p['Y'], multistart=int(os.environ.get('C4_RESTARTS', 8)), random_restarts=True, seed=1234,
               numpy_path_conventions=False, optimizer_opts={'maxiter': 5}, optimizer='scipy')
dt = time.perf_counter() - t0
print(json.dumps({'bench': 'C4 multistart (5 SLSQP iterations per restart)', 'restarts': int(os.environ.get('C4_RESTARTS', 8)),
                  'world': opt['world'], 'total_s': dt, 'n_eval_this_rank': opt['n_eval'], 'evals_per_s': opt['n_eval'] / dt,
                  'best_nll': float(np.min(opt['obj']))}))
