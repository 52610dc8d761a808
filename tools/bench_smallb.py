#!/usr/bin/env python3
"""Variance product for small batches of test points on the C3-size model (N=8192, Ny=6, d=8): time per
mean+variance call against the batch size (the L^-1 stream, 1.6 GB, bounds it below ~64 columns)."""
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
Ny, d = 6, 8
lib = get_lib()
p = synthetic_problem(N, d, Ny, 256, seed=1234, sn=1e-2)
h = Handle(lib, p['X'], p['Y'])
h.fit(p['hyper'])
h.profile_enable(True)
for B in [int(b) for b in os.environ.get('SMALLB_LIST', '1,2,3,4,8,16,32,48,64,96,128,256').split(',')]:
    Z = p['Z'][:B]
    h.predict_mean_var(Z)
    h.profile_read()
    t0 = time.perf_counter()
    for it in range(20):
        h.predict_mean_var(Z)
    dt = (time.perf_counter() - t0) / 20
    prof = h.profile_read()
    vg = prof.get('vargemm', (0, 0))[0] / 20
    print(json.dumps({'bench': 'small batch mean+var', 'B': B, 'ms_per_call': dt * 1e3, 'vargemm_ms': vg,
                      'Linv_stream_TBps': 4.0 * N * (N + 1) * Ny / (vg * 1e-3) * 1e-12 if vg else None}))
