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
