#!/bin/bash
# where does the host time of one NLL + gradient evaluation go?  HIP API trace + kernel trace of 100 evaluations at C4 size
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp; cd /tmp
cat > /tmp/nll_loop.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/oracle')
import numpy as np, gp_oracle as go
from gp_mpc_amd._lib import Handle, get_lib
p = go.synthetic_problem(4096, 6, 1, 1, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
hp = p['hyper'][0].copy()
for i in range(5): h.nll(0, hp, want_grad=True)
t0 = time.perf_counter()
for i in range(100): h.nll(0, hp * (1 + 1e-3 * (i % 7)), want_grad=True)
print('ms per NLL+grad evaluation: %.3f' % ((time.perf_counter() - t0) * 10))
t0 = time.perf_counter()
for i in range(100): h.nll(0, hp * (1 + 1e-3 * (i % 7)))
print('ms per NLL evaluation: %.3f' % ((time.perf_counter() - t0) * 10))
PY
python /tmp/nll_loop.py
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats -d "$R/gpurun_out/nlltrace" -o t -- python /tmp/nll_loop.py > "$R/gpurun_out/nlltrace.log" 2>&1; echo "rc=$?"
ls "$R/gpurun_out/nlltrace" | head
python - <<PY
import sqlite3, collections
con = sqlite3.connect("$R/gpurun_out/nlltrace/t_results.db")
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'region' in t or 'api' in t.lower() or 'hip' in t.lower()][:20])
for t in ('regions', 'rocpd_region', 'hip_api'):
    try:
        cols = [d[1] for d in con.execute(f'pragma table_info({t})')]
        print(t, cols)
    except Exception as e:
        pass
try:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in con.execute("select name, start, end from regions"):
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print('%-40s calls %6d  total %10.1f us  avg %8.2f us' % (k[:40], v[0], v[1], v[1] / v[0]))
except Exception as e:
    print('regions query failed', e)
PY
