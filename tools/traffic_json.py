#!/usr/bin/env python3
"""profiles/traffic.json from three separate rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1` (tools/gpu_refresh.sh pmc):
    python tools/traffic_json.py <FETCH_SIZE db> <WRITE_SIZE db> <MFMA-busy db> <tag>
HBM bytes per launch of the dominant kernel (vargemm_persist_kernel): FETCH_SIZE (KB) x 2 -- gfx950 tallies the 16 B/lane
coalesced reads this kernel issues at half (/opt/skills/guides/MI355X_MICROARCH.md, HBM) -- + WRITE_SIZE (KB, uncorrected);
Infinity-Cache hits are counted as fetches.  Matrix-pipe duty = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)."""
import collections
import json
import sqlite3
import sys

KERNEL = 'vargemm_persist_kernel'


def per_dispatch(path, counters):
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute('pragma table_info(pmc_events)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for n, c, v, disp in db.execute(f'select {name_col}, counter_name, counter_value, dispatch_id from pmc_events'):
        if KERNEL in n and c in counters:
            acc[disp][c] += float(v)
    return acc


fetch = per_dispatch(sys.argv[1], {'FETCH_SIZE'})
write = per_dispatch(sys.argv[2], {'WRITE_SIZE'})
busy = per_dispatch(sys.argv[3], {'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES'})
f_kb = sum(d['FETCH_SIZE'] for d in fetch.values()) / max(len(fetch), 1)
w_kb = sum(d['WRITE_SIZE'] for d in write.values()) / max(len(write), 1)
mf = sum(d['SQ_VALU_MFMA_BUSY_CYCLES'] for d in busy.values())
cu = sum(d['SQ_BUSY_CU_CYCLES'] for d in busy.values())
N, B = 4096, 10000
out = {
    'kernel': KERNEL + ': variance GEMM as one persistent launch over a static schedule (128 x 128 tiles, 512 resident workgroups)',
    'N': N, 'B': B, 'fetch_size_kb_raw': f_kb, 'write_size_kb_raw': w_kb, 'dispatches_used': len(fetch),
    'hbm_bytes_per_launch': (2.0 * f_kb + w_kb) * 1024.0,
    'correction': 'FETCH_SIZE x2 (gfx950: 16 B/lane coalesced reads are tallied at half, MI355X_MICROARCH.md section HBM); '
                  'WRITE_SIZE uncorrected; Infinity-Cache hits are counted as fetches',
    'algorithmic_bytes': 4 * N * (N + 1) + 8 * N * B,      # lower triangle of L^-1 + the cross-covariances, read once
    'matrix_pipe_busy': mf / (4.0 * cu) if cu else None,
    'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (separate passes) '
              '-- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary; tools/gpu_refresh.sh %s pmc' % (sys.argv[4] if len(sys.argv) > 4 else ''),
}
print(json.dumps(out, indent=1))
