#!/usr/bin/env python3
"""Decode the worker time stamps written next to the chain's (GPMPC_CHAIN_TRACE=<file>, chol_worker.hpp):
per panel step and over all workers: when they arrive at the step (relative to the chain's publication of leafdone/pan1),
when their panel tiles and hand-off tiles are done, when the panel column is complete, when their updates end."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
chain = raw[:nb * 8].reshape(nb, 8).astype(float) / 100.0
wk = raw[4096:4096 + 2 * 256 * 64 * 4].reshape(2, 256, 64, 4).astype(float) / 100.0
pub = chain[:, 6]            # leafdone[k] + pan1[k] are out (merged publication)
print('step  pub->next_pub | arrive(min/med/max)   part1+2 done(med/max)   colready seen(min/med/max)   part 3 end(med/max)   part3 len(med/max)  [us after publication k]')
for k in range(0, nb - 2):
    for l in range(2):
        t = wk[l, :, k, :]
        m = t[:, 0] > 0
        if not m.any():
            continue
        t = t[m] - pub[k]
        full = t[:, 3] > -1e6
        has3 = (wk[l, :, k, 3][m] > 0)
        a, b, c, d = t[:, 0], t[:, 1], t[has3, 2], t[has3, 3]
        nxt = pub[k + 1] - pub[k] if k + 1 < nb else float('nan')
        if k % 2 == 0 or k < 8:
            print('%3d%s  %6.1f | %6.1f %6.1f %6.1f   %6.1f %6.1f   %6.1f %6.1f %6.1f   %6.1f %6.1f   %6.1f %6.1f  (%d workers)' % (
                k, 'ab'[l], nxt, a.min(), np.median(a), a.max(), np.median(b), b.max(),
                c.min() if len(c) else 0, np.median(c) if len(c) else 0, c.max() if len(c) else 0,
                np.median(d) if len(d) else 0, d.max() if len(d) else 0,
                np.median(d - c) if len(d) else 0, (d - c).max() if len(d) else 0, m.sum()))
