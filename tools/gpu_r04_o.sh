#!/bin/bash
# r04: what the alpha / mean kernels that run next to the variance product cost it: the same step with the mean not asked for
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cat > /tmp/novar.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, torch, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
lib = get_lib()
N, d, B = 4096, 6, 10000
p = go.synthetic_problem(N, d, 1, B, seed=1234, sn=1e-2)
h = Handle(lib, p['X'], p['Y'])
z = torch.from_numpy(p['Z']).to('cuda:0')
mean = torch.empty((B, 1), dtype=torch.float64, device='cuda:0'); var = torch.empty((B, 1), dtype=torch.float64, device='cuda:0')
hyper = np.ascontiguousarray(p['hyper']); h.set_pointer_mode(True)
for with_mean in (1, 0, 1, 0):
    for it in range(25):
        if it == 5: h.synchronize(); t0 = time.perf_counter()
        h.fit(hyper)
        h.predict_mean_var_dev(B, z.data_ptr(), mean.data_ptr() if with_mean else 0, var.data_ptr())
    h.synchronize()
    print('mean %d: %.3f ms per step' % (with_mean, (time.perf_counter() - t0) / 20 * 1e3))
PY
timeout 300 python /tmp/novar.py 2>&1 | tail -5
