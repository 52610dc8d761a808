#!/usr/bin/env python3
"""One bench step (last gram_kernel .. end of the variance GEMM) of a rocprofv3 kernel trace as a timeline.
Usage: python tools/step_timeline.py <t_results.db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute('select name,start,end,queue_id from kernels order by start'))
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '').replace('gpmpc::', '')[:60]
grams = [i for i, r in enumerate(rows) if 'gram_kernel' in r[0]]
i0 = grams[-1]
t0 = rows[i0][1]
for n, s, e, q in rows[i0:]:
    print('%9.1f -> %9.1f  (%7.1f)  q%-3s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
    if 'var_finish' in n:
        break
