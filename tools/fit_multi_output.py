#!/usr/bin/env python3
"""Fit time of multi-output models at the reference's typical sizes (Ny outputs share X; one factorisation per output)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
import numpy as np
from gp_mpc_amd.synthetic import synthetic_problem
from gp_mpc_amd._lib import Handle, get_lib

lib = get_lib()
for N, Ny in ((200, 4), (500, 4), (1000, 6), (2000, 6), (4096, 2), (4096, 6)):
    p = synthetic_problem(N, 6, Ny, 1, seed=1, sn=1e-2)
    h = Handle(lib, p['X'], p['Y'])
    hyp = p['hyper']
    for _ in range(3):
        h.fit(hyp)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        h.fit(hyp)
    dt = (time.perf_counter() - t0) / reps
    print('N=%5d Ny=%d  fit %.3f ms  (%.3f ms per output)' % (N, Ny, dt * 1e3, dt * 1e3 / Ny), flush=True)
    h.close()
