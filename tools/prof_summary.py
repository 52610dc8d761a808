#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (rocpd sqlite .db) per kernel and grid shape.
Usage: python tools/prof_summary.py <results.db> [--steps K]  (K divides totals into per-step figures)"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 1
rows = list(db.execute('select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels order by start'))
agg = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
for n, gx, gy, gz, wx, s, e in rows:
    short = re.sub(r'\(.*', '', n).replace('void ', '').replace('gpmpc::', '')
    agg[(short, gx // max(wx, 1), gy, gz)][0] += 1
    agg[(short, gx // max(wx, 1), gy, gz)][1] += (e - s) / 1e3
    tot[short][0] += 1
    tot[short][1] += (e - s) / 1e3
total = sum(v[1] for v in tot.values())
print('%-46s %8s %12s %10s %7s' % ('KERNEL (all grids)', 'calls', 'total_us', 'avg_us', '%'))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print('%-46s %8d %12.1f %10.2f %6.1f%%' % (k[:46], v[0], v[1], v[1] / v[0], 100 * v[1] / total))
print()
print('%-40s %16s %8s %12s %10s %12s' % ('KERNEL x GRID', 'grid', 'calls', 'total_us', 'avg_us', 'us/step'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print('%-40s %5dx%5dx%3d %8d %12.1f %10.2f %12.1f' % (k[0][:40], k[1], k[2], k[3], v[0], v[1], v[1] / v[0], v[1] / steps))
