#!/usr/bin/env python3
"""Fit time of nb matrices of one size as ONE batched factorisation (a model with nb outputs): what a lock-step round of
nb hyper-parameter restarts would cost on the two-level path (r04 design study for the batched restart search)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
import numpy as np
from gp_mpc_amd.synthetic import synthetic_problem
from gp_mpc_amd._lib import Handle, get_lib

lib = get_lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for nb in (1, 2, 4, 8, 16, 32, 64):
    p = synthetic_problem(N, 6, 1, 1, seed=1, sn=1e-2)
    Y = np.repeat(p['Y'], nb, axis=1)
    hyp = np.repeat(p['hyper'], nb, axis=0) * (1.0 + 0.01 * np.arange(nb))[:, None]
    h = Handle(lib, p['X'], Y)
    for _ in range(2):
        h.fit(hyp)
    h.profile_enable(True)
    h.profile_read(reset=True)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        h.fit(hyp)
    h.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pr = h.profile_read()
    fl = nb * 2.0 * N ** 3 / 3.0
    print('N=%d nb=%2d  fit %.3f ms (%.3f per matrix)  factor %.3f ms = %.1f TFLOP/s (chol+inverse)  gram %.3f' % (
        N, nb, dt * 1e3, dt * 1e3 / nb, pr['factor'][0] / reps, fl / (pr['factor'][0] / reps * 1e-3) * 1e-12, pr['gram'][0] / reps), flush=True)
    h.close()
