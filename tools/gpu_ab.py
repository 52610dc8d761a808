#!/usr/bin/env python3
"""One parametrised same-box A/B runner (replaces the single-use tools/gpu_r0x_*.sh of rounds 1-4).

    python tools/gpu_ab.py [--rounds 3] [--cmd "python bench.py ..."] [--fields a,b.c,...] [--out FILE] VARIANT...

VARIANT = label[:KEY=VAL[,KEY=VAL...]] -- environment switches of the library ("Run-time switches") for that
variant; the pseudo key `lib=<name>` copies tools/ab_libs/<name>.so over the product library for the variant's runs (built here
from another commit: `tools/build_ab_lib.sh <commit> <name>`; the .so files travel with gpurun, git ignores them).
The variants run interleaved, `--rounds` times each, on whatever box the call landed on; every run is the JSON line of
`--cmd` (default: the C2 bench without CPU baseline / secondary configs); `--fields` are dotted paths into that line.
Prints one row per run and the per-variant mean / min; writes the same text to --out (under gpurun_out/)."""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'gp_mpc_amd', 'csrc', 'libgpmpc_hip.so')


def dig(j, path):
    for k in path.split('.'):
        if j is None:
            return None
        j = j.get(k) if isinstance(j, dict) else None
    return j


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--cmd', default='python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary')
    ap.add_argument('--fields', default='ms_per_step,phases_ms_per_step.chain,phases_ms_per_step.factor,'
                                        'phases_ms_per_step.crosscov,phases_ms_per_step.vargemm')
    ap.add_argument('--out', default=None)
    ap.add_argument('--timeout', type=int, default=600)
    ap.add_argument('variants', nargs='+')
    a = ap.parse_args()
    fields = a.fields.split(',')
    variants = []
    for v in a.variants:
        label, _, kv = v.partition(':')
        env = {k: v.replace(';', ',') for k, v in (x.split('=', 1) for x in kv.split(',') if x)}     # (';' in a value stands for ',')
        variants.append((label, env))
    backup = LIB + '.ab_backup'
    shutil.copy(LIB, backup)
    rows = {label: [] for label, _ in variants}
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    say('cmd: ' + a.cmd)
    say('%-28s %s' % ('variant', ' '.join('%14s' % f.split('.')[-1][:14] for f in fields)))
    try:
        for rnd in range(a.rounds):
            for label, env in variants:
                e = dict(os.environ)
                lib = None
                for k, val in env.items():
                    if k == 'lib':
                        lib = val
                    else:
                        e[k] = val
                shutil.copy(os.path.join(ROOT, 'tools', 'ab_libs', lib + '.so') if lib else backup, LIB)
                try:
                    r = subprocess.run(a.cmd, shell=True, cwd=ROOT, env=e, capture_output=True, text=True, timeout=a.timeout)
                    js = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
                    j = json.loads(js[-1]) if js else None
                except subprocess.TimeoutExpired:
                    j, r = None, None
                if j is None:
                    say('%-28s FAILED rc=%s %s' % (label, getattr(r, 'returncode', 'timeout'), (r.stderr[-300:] if r else '').replace('\n', ' | ')))
                    continue
                vals = [dig(j, f) for f in fields]
                rows[label].append(vals)
                say('%-28s %s' % (label, ' '.join('%14.4f' % v if isinstance(v, (int, float)) else '%14s' % str(v)[:14] for v in vals)))
    finally:
        shutil.copy(backup, LIB)
        os.remove(backup)
    say('-- mean / min over %d rounds' % a.rounds)
    for label, _ in variants:
        rs = rows[label]
        if not rs:
            continue
        cols = list(zip(*rs))
        num = [[x for x in c if isinstance(x, (int, float))] for c in cols]
        say('%-28s %s' % (label + ' mean', ' '.join('%14.4f' % (sum(c) / len(c)) if c else '%14s' % '-' for c in num)))
        say('%-28s %s' % (label + ' min', ' '.join('%14.4f' % min(c) if c else '%14s' % '-' for c in num)))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
