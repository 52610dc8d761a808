import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
from gp_mpc_amd.synthetic import synthetic_problem
from gp_mpc_amd._lib import Handle, get_lib
lib = get_lib()
N, d, B = 4096, 6, 10000
p = synthetic_problem(N, d, 1, B, seed=1234, sn=1e-2)
h = Handle(lib, p['X'], p['Y'])
dev = torch.device('cuda:0')
z = torch.tensor(p['Z'], dtype=torch.float64, device=dev)
mean = torch.empty((B, 1), dtype=torch.float64, device=dev); var = torch.empty((B, 1), dtype=torch.float64, device=dev)
hyper = np.ascontiguousarray(p['hyper'])
h.set_pointer_mode(True)
def step():
    h.fit(hyper); h.predict_mean_var_dev(B, z.data_ptr(), mean.data_ptr(), var.data_ptr())
for prof in (False, True, False, True):
    for _ in range(5): step()
    h.synchronize()
    h.profile_enable(prof); h.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(100): step()
    h.synchronize()
    dt = (time.perf_counter() - t0) / 100
    h.profile_enable(False); h.profile_read(reset=True)
    print('profiling %s: %.3f ms per step' % ('on ' if prof else 'off', dt * 1e3), flush=True)
