#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters from a rocpd sqlite DB.
Usage: python tools/pmc_summary.py <results.db>"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in db.execute('pragma table_info(pmc_events)')]
if '--schema' in sys.argv:
    print(cols)
    for r in db.execute('select * from pmc_events limit 3'):
        print(r)
name_col = 'name' if 'name' in cols else 'kernel_name'
rows = db.execute(f'select {name_col}, counter_name, counter_value, dispatch_id from pmc_events')
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
per_disp = collections.defaultdict(float)
for n, c, v, disp in rows:
    short = re.sub(r'\(.*', '', n).replace('void ', '').replace('gpmpc::', '')
    per_disp[(short, c, disp)] += float(v)
for (short, c, disp), v in per_disp.items():
    agg[short][c][0] += 1
    agg[short][c][1] += v
for k in sorted(agg):
    print(k)
    for c, (n, tot) in sorted(agg[k].items()):
        print('    %-32s dispatches %6d   avg/dispatch %18.1f   total %20.1f' % (c, n, tot / n, tot))
