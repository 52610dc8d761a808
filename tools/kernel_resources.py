#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/AGPR/spills/LDS/occupancy) for the gfx950 build.
Usage: python tools/kernel_resources.py [source.hip]"""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gp_mpc_amd', 'csrc', 'gpmpc_api.hip')
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-value', '-c',
       '-Rpass-analysis=kernel-resource-usage', src, '-o', '/dev/null']
out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r'remark: .*?(Function Name|Name): (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', name)}
        rows.append(cur)
        continue
    m = re.search(r'remark: .*?\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
keys = ['VGPRs', 'AGPRs', 'VGPRs Spill', 'SGPRs Spill', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]',
        'LDS Size [bytes/block]']
print('%-58s %5s %5s %6s %6s %7s %4s %7s' % ('kernel', 'VGPR', 'AGPR', 'vspill', 'sspill', 'scratch', 'occ', 'LDS'))
for r in rows:
    print('%-58s %5s %5s %6s %6s %7s %4s %7s' % tuple([r['name'][:58]] + [r.get(k, '-') for k in keys]))
