#!/usr/bin/env python3
"""The courier workgroup's time stamps (GPMPC_CHAIN_TRACE=<file>, chol_worker.hpp), per panel step relative to the chain's
publications: leafdone[k] (chain stamp 2) and pan1[k] (stamp 6); the chain looks for the tiles at the end of leaf k+1."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.int64)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
chain = raw[:nb * 8].reshape(nb, 8).astype(float) / 100.0
c = raw[300000:300000 + nb * 8].reshape(nb, 8).astype(float) / 100.0
print('step | tiles handed (before leafdone) | leafdone seen (+) | L(k+2,k) published (after leafdone) | pan1 seen (+) | tdone published (after pan1) | chain: pan1 -> end of next leaf')
rows = []
for k in range(1, nb - 3):
    if c[k, 4] == 0: continue
    ld, p1 = chain[k, 2], chain[k, 6]
    nxt = chain[k + 1, 1] - p1
    rows.append((ld - c[k, 0], c[k, 1] - ld, c[k, 2] - ld, c[k, 3] - p1, c[k, 4] - p1, nxt))
    if k % 4 == 1: print('%3d  | %6.1f | %5.1f | %5.1f | %5.1f | %5.1f | %5.1f' % ((k,) + rows[-1]))
r = np.array(rows)
print('mean | %6.1f | %5.1f | %5.1f | %5.1f | %5.1f | %5.1f' % tuple(r.mean(axis=0)))
