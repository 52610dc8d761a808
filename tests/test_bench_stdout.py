"""bench.py's contract is ONE JSON line on stdout; everything else a process writes to file descriptor 1 (RCCL prints a
version banner from C stdio at exit) must not end up there.  CPU only: the plumbing, not the benchmark."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def test_only_the_json_line_reaches_stdout():
    code = (
        "import os, sys, ctypes\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "bench._own_stdout()\n"
        "print('python-level noise')\n"
        "os.write(1, b'fd-level noise\\n')\n"
        "ctypes.CDLL(None).puts(b'C stdio noise, flushed at exit')\n"
        "bench._emit({'metric': 'm', 'value': 1.5})\n"
        "print('more noise after the line')\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {'metric': 'm', 'value': 1.5}, r.stdout
    for noise in ('python-level noise', 'fd-level noise', 'C stdio noise', 'more noise'):
        assert noise in r.stderr


def test_without_a_gpu_bench_says_so_and_prints_no_line():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('GPU present')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == '' and 'GPU' in r.stderr
