#!/usr/bin/env python3
"""Per-tile time stamps of the first 8 workers (GPMPC_CHAIN_TRACE=<file>, chol_worker.hpp: entry 135168 + (worker * 64 + step) * 10 + slot
= when that slot's tile update of that step was finished): time between consecutive tiles of a worker's step."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.int64)
t = raw[135168:135168 + 8 * 64 * 10].reshape(8, 64, 10).astype(float) / 100.0
wk = raw[4096:4096 + 2 * 256 * 64 * 4].reshape(2, 256, 64, 4).astype(float) / 100.0     # [launch][worker][step][stamp]
for k in (1, 2, 4, 8, 12, 16, 24):
    rows = []
    for w in range(8):
        v = np.sort(t[w, k][t[w, k] > 0])
        if len(v) < 2: continue
        start = wk[0, w, k, 2]                    # colready seen (part 3 starts)
        d = np.diff(np.concatenate([[start], v]))
        rows.append((len(v), d))
    if not rows: continue
    print('step %2d:' % k, ' | '.join('%d tiles: first %.1f, then %s' % (n, d[0], ' '.join('%.1f' % x for x in d[1:])) for n, d in rows[:4]))
