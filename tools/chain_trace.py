#!/usr/bin/env python3
"""Decode the time stamps of chol_chain_kernel (GPMPC_CHAIN_TRACE=<file>, 100 MHz clock):
per panel step: leaf | publish | wait for the trailing tiles | load | panel row + store | publish | diagonal update."""
import sys
import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8).astype(float) / 100.0   # microseconds
nb = int(sys.argv[2]) if len(sys.argv) > 2 else t.shape[0]
t = t[:nb]
names = ['leaf', 'publish leaf', 'wait tiles', 'load A(k+1,k)', 'panel row+store', 'publish row', 'diag update']
seg = np.diff(t[:-1], axis=1)
print('steps %d, chain total %.1f us' % (nb, t[-1, 2] - t[0, 0]))
for i, n in enumerate(names):
    print('%-18s mean %6.2f us   sum %8.1f us   (first 8: %s)' % (n, seg[:, i].mean(), seg[:, i].sum(),
          ' '.join('%.1f' % v for v in seg[:8, i])))
gap = t[1:, 0] - t[:-1, 7]
print('%-18s mean %6.2f us   sum %8.1f us' % ('loop gap', gap[:-1].mean(), gap[:-1].sum()))
q = nb // 4
for j in range(4):
    sl = slice(j * q, min((j + 1) * q, nb - 1))
    print('quarter %d: step %.2f us (wait %.2f)' % (j, (t[sl, 7] - t[sl, 0]).mean(), seg[sl, 2].mean()))

# the leaf's own stamps (thread 0 = the wave that factors the panels), entries 200000 + 16 k + i
raw = np.fromfile(sys.argv[1], dtype=np.int64)
if raw.size >= 200000 + 16 * nb and raw[200000:200000 + 16 * nb].any():
    lf = raw[200000:200000 + 16 * nb].reshape(nb, 16).astype(float) / 100.0
    lf = lf[1:nb - 1]
    lab = ['panel 0', 'barrier', 'rank-16 update 0 + barrier', 'panel 1', 'barrier', 'update 1 + barrier', 'panel 2', 'barrier (+flag look)',
           'update 2 + barrier', 'panel 3', 'barrier (+landing)', 'last 16x16 inverse + barrier', 'inverse of the 32-blocks', 'inverse of the 64-block']
    idx = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 11), (11, 12), (12, 13), (13, 14)]
    print('inside the leaf (wave 0), mean over steps 1..%d:' % (nb - 2))
    for (a, b), n in zip(idx, lab):
        print('  %-32s %6.2f us' % (n, (lf[:, b] - lf[:, a]).mean()))
    print('  %-32s %6.2f us' % ('total', (lf[:, 14] - lf[:, 0]).mean()))
