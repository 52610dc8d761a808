#!/bin/bash
# r04 design study: kernel trace of a batched fit (nb x 4096^2 through the two-level path)
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp; cd /tmp
cat > /tmp/bf.py <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/oracle")
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
lib = get_lib(); nb = int(sys.argv[1]); N = 4096
p = go.synthetic_problem(N, 6, 1, 1, seed=1, sn=1e-2)
Y = np.repeat(p['Y'], nb, axis=1); hyp = np.repeat(p['hyper'], nb, axis=0) * (1.0 + 0.01 * np.arange(nb))[:, None]
h = Handle(lib, p['X'], Y)
for _ in range(3): h.fit(hyp)
h.synchronize(); h.close()
PY
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_bf" -o t -- python /tmp/bf.py 16 > "$R/gpurun_out/prof_bf.log" 2>&1; echo "rocprof rc=$?"
python - <<PY
import sqlite3, re
db = sqlite3.connect("$R/gpurun_out/prof_bf/t_results.db")
rows = list(db.execute('select name,start,end,queue_id,grid_x,grid_z from kernels order by start'))
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '').replace('gpmpc::', '')[:70]
g = [i for i, r in enumerate(rows) if 'gram_kernel' in r[0]]
i0 = g[-1]; t0 = rows[i0][1]
agg = {}
for n, s, e, q, gx, gz in rows[i0:]:
    k = short(n); a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
end = max(r[2] for r in rows[i0:])
print('last fit: %.1f us wall' % ((end - t0) / 1e3))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print('%8.1f us  x%4d  %s' % (t, c, k))
print('--- first 60 launches')
for n, s, e, q, gx, gz in rows[i0:i0 + 60]:
    print('%9.1f -> %9.1f (%7.1f) q%-2s grid %6d z%3d %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, gx, gz, short(n)))
PY
