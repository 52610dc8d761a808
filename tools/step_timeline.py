#!/usr/bin/env python3
"""One bench step (a gram_kernel .. end of the variance GEMM) of a rocprofv3 kernel trace as a timeline.
Usage: python tools/step_timeline.py <t_results.db> [index of the gram_kernel launch, default -1 = the last; e.g. 3 = the fourth
step of the run, in the middle of the timed region: kernels of the PREVIOUS step that are still running show up with negative times]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute('select name,start,end,queue_id from kernels order by start'))
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '').replace('gpmpc::', '')[:60]
grams = [i for i, r in enumerate(rows) if 'gram_kernel' in r[0]]
i0 = grams[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
t0 = rows[i0][1]
for n, s, e, q in [r for r in rows[:i0] if r[2] > t0] + rows[i0:]:
    print('%9.1f -> %9.1f  (%7.1f)  q%-3s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
    if 'var_finish' in n:
        break
