"""CPU tier: the C-ABI shared library builds for gfx950, loads, and exports exactly the symbols
that include/gpmpc.h declares (no compute calls here -- there is no GPU)."""
import os
import re
import subprocess

import pytest

from gp_mpc_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'gpmpc.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(gpmpc_[a-z_0-9]+)\s*\(', text))


@pytest.fixture(scope='module')
def built():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    return _lib.LIB_PATH


def test_header_binding_and_exports_agree(built):
    hdr = header_symbols()
    assert hdr == set(_lib.SIGNATURES), hdr ^ set(_lib.SIGNATURES)
    out = subprocess.run(['nm', '-D', '--defined-only', built], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (gpmpc_[a-z_0-9]+)', out))
    assert hdr <= exported, hdr - exported


def test_library_loads_and_reports_without_gpu(built):
    lib = _lib.GpmpcLib(built)          # resolves every declared symbol or raises AttributeError
    assert lib.dll.gpmpc_abi_version() == 2
    n = lib.device_count()
    if n == 0:                           # build container: creating a model must fail loudly, not fall back
        import numpy as np
        with pytest.raises(_lib.GpmpcError) as e:
            _lib.Handle(lib, np.zeros((4, 2)), np.zeros((4, 1)))
        assert e.value.code == _lib.EHIP and 'HIP device' in str(e.value)


def test_persistent_schedules_cover_every_tile_once_balanced_and_local(built):
    """The static schedules of the persistent variance / K^-1 products (vargemm_persist.hpp; host code, no GPU): every tile
    exactly once, the heaviest slot within 1 % of the mean at the benchmark shapes, and -- the XCDs are levelled first -- all
    but a few tiles on the XCD that holds the rest of their operand panel."""
    lib = _lib.GpmpcLib(built)
    c2 = lib.schedule_stats(0, 32, 79, 1, 4096, 512)              # C2: N = 4096, B = 10 000
    assert c2['wrong'] == 0 and c2['tiles'] == 32 * 79
    assert c2['max_load'] <= 1.01 * c2['mean_load'] and c2['home'] >= c2['tiles'] - 40, c2
    c3 = lib.schedule_stats(0, 64, 79, 6, 8192, 512)
    assert c3['wrong'] == 0 and c3['max_load'] <= 1.005 * c3['mean_load'], c3
    for n in (64, 53, 9, 6):                                       # K^-1 of the lock-step search's batches (Np = 4096)
        k = lib.schedule_stats(1, 32, 32, n, 4096, 512)
        assert k['wrong'] == 0 and k['tiles'] == 528 * n and k['max_load'] <= 1.02 * k['mean_load'], (n, k)
        assert k['home'] >= k['tiles'] - 16 * 8, (n, k)
    odd = lib.schedule_stats(0, 3, 2, 2, 320, 16)                  # ragged: Np = 320 (last block row partial), emulator size
    assert odd['wrong'] == 0 and odd['tiles'] == 12
    one = lib.schedule_stats(1, 5, 5, 1, 640, 512)                 # fewer tiles than slots
    assert one['wrong'] == 0 and one['tiles'] == 15 and one['longest'] == 1


def test_gfx950_code_object_present(built):
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', built], capture_output=True, text=True)
    txt = out.stdout + out.stderr
    assert 'gfx950' in txt


FORBIDDEN_MODULES = ('oracle', 'gp_oracle', 'make_golden', 'emu', 'emu_runtime')
FORBIDDEN_TARGETS = ('oracle', 'gp_oracle', 'libgpmpc_emu', 'tests/emu', 'emu_runtime')


def _python_violations(path):
    """Parse one product source: imports (static and through importlib / __import__), and every string that reaches a
    loader or a process launcher (ctypes.CDLL / cdll.LoadLibrary / dlopen, subprocess.*, os.system / os.exec* / os.popen,
    runpy, open / exec of a file).  Comments and docstrings cannot trip it; a real use cannot hide in a string constant
    passed to one of those calls, and string constants naming an oracle / emulator path anywhere else are reported too."""
    import ast
    tree = ast.parse(open(path).read(), path)
    bad = []

    def mod_forbidden(name):
        parts = (name or '').split('.')
        return any(p in FORBIDDEN_MODULES for p in parts)

    def str_forbidden(sv):
        low = sv.replace('\\', '/').lower()
        return any(t in low for t in FORBIDDEN_TARGETS)

    docstrings = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Module, ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) and node.body:
            first = node.body[0]
            if isinstance(first, ast.Expr) and isinstance(first.value, ast.Constant) and isinstance(first.value.value, str):
                docstrings.add(id(first.value))
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            bad += [f'import {a.name}' for a in node.names if mod_forbidden(a.name)]
        elif isinstance(node, ast.ImportFrom):
            if mod_forbidden(node.module) or any(mod_forbidden(a.name) for a in node.names):
                bad.append(f'from {node.module} import ...')
        elif isinstance(node, ast.Constant) and isinstance(node.value, str) and id(node) not in docstrings:
            # every non-docstring string constant: arguments of CDLL / dlopen / subprocess / importlib / open included
            if str_forbidden(node.value):
                bad.append(f'string constant {node.value!r} (line {node.lineno})')
    return bad


def _native_violations(path):
    """C++ / HIP sources: #include targets and string literals (dlopen, system, popen arguments) after comments are removed."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    bad = []
    for inc in re.findall(r'#\s*include\s*[<"]([^>"]+)[>"]', src):
        if any(t in inc.lower() for t in FORBIDDEN_TARGETS):
            bad.append(f'#include {inc}')
    for lit in re.findall(r'"((?:[^"\\\n]|\\.)*)"', src):
        if any(t in lit.lower() for t in FORBIDDEN_TARGETS):
            bad.append(f'string literal "{lit}"')
    return bad


def test_no_oracle_or_cpu_fallback_in_product():
    """The shipped package must not import, dlopen, execute or read the oracle or the emulator (comments may mention them)."""
    seen = 0
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'gp_mpc_amd')):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith('.py'):
                assert not _python_violations(path), (f, _python_violations(path))
                seen += 1
            elif f.endswith(('.hip', '.hpp', '.inl', '.h')):
                assert not _native_violations(path), (f, _native_violations(path))
                seen += 1
    assert seen >= 10


def test_guard_catches_real_uses_and_ignores_comments(tmp_path):
    ok = tmp_path / 'ok.py'
    ok.write_text('"""mentions oracle/gp_oracle.py in a docstring"""\n# and gp_oracle in a comment\nimport numpy\n')
    assert not _python_violations(str(ok))
    for body in ('import gp_oracle\n', 'from oracle import gp_oracle\n', 'import importlib\nimportlib.import_module("gp_oracle")\n',
                 'import ctypes\nctypes.CDLL("tests/emu/_build/libgpmpc_emu.so")\n',
                 'import subprocess\nsubprocess.run(["python", "oracle/gp_oracle.py"])\n', '__import__("gp_oracle")\n'):
        f = tmp_path / 'bad.py'
        f.write_text(body)
        assert _python_violations(str(f)), body
    c = tmp_path / 'x.hpp'
    c.write_text('// oracle/gp_oracle.py is only named here\nint f();\n')
    assert not _native_violations(str(c))
    c.write_text('void* p = dlopen("libgpmpc_emu.so", 2);\n')
    assert _native_violations(str(c))


def test_inline_dpp_instructions_respect_the_operand_hazard():
    """The leaf's v_fmac_f64_dpp statements are inline assembly, invisible to hipcc's hazard recognizer: the generated
    device code must keep 2 wait states between a VALU write of a register and a DPP read of it (tools/check_dpp_hazards.py
    compiles the library's device code with the Makefile's flags and walks every kernel)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_dpp_hazards as chk
    # the checker itself: a violation in a hand-written snippet is found, its s_nop-protected twin is not
    bad = "k1:\n\tv_mul_f64 v[2:3], v[4:5], v[6:7]\n\tv_fmac_f64_dpp v[8:9], -v[2:3], v[10:11] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
    good = bad.replace('\tv_fmac', '\ts_nop 1\n\tv_fmac')
    assert len(chk.check(bad)) == 1 and chk.check(good) == []
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_dpp_hazards.py')], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    n = int(re.search(r'(\d+) DPP instructions checked', r.stdout).group(1))
    # the leaf's panel (one copy per kernel since r03) and block 0's 16 x 16 inverse are in there (r05: 752 -- the other three
    # blocks' inverses come out of the panels' own instructions, leaf64.hpp panel_potrf_dpp_at)
    assert n >= 700, r.stdout


def test_every_environment_switch_is_documented():
    """INTEGRATION.md lists every GPMPC_* variable the library reads (a maintainer of the reference side finds each A/B switch there)."""
    csrc = os.path.join(ROOT, 'gp_mpc_amd', 'csrc')
    names = set()
    for fn in os.listdir(csrc):
        if fn.endswith(('.inl', '.hpp', '.hip')):
            names |= set(re.findall(r'getenv\("(GPMPC_[A-Z0-9_]+)"\)', open(os.path.join(csrc, fn)).read()))
    assert len(names) > 40, len(names)
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_hot_kernels_keep_their_register_budget():
    """Resource regressions the compiler makes silently (tools/kernel_resources.py, a cross-compile of the gfx950 code):
    the worker kernels without scratch (r04: two coordinate arrays with a run-time index had been put there, two scratch
    loads in front of every tile's DMA requests), no vector spills in the chain and worker kernels, the persistent variance
    product inside the 120 registers (one allocation granule of 8: r06) that leave room for a fifth wave per SIMD."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'kernel_resources.py')], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {}
    for line in r.stdout.splitlines()[1:]:
        m = re.match(r'(.{58}) +(\S+) +(\S+) +(\S+) +(\S+) +(\S+) +(\S+) +(\S+)$', line)
        if m:
            rows[m.group(1).strip()] = [int(x) if x.isdigit() else -1 for x in m.groups()[1:]]
    workers = [k for k in rows if 'chol_worker_kernel' in k]
    assert len(workers) == 2, sorted(rows)[:5]
    for k in workers:
        vgpr, agpr, vspill, sspill, scratch, occ, lds = rows[k]
        assert scratch == 0 and vspill == 0 and vgpr <= 256 and occ >= 2, (k, rows[k])
    chain = rows['gpmpc::chol_chain_kernel']
    assert chain[2] == 0 and chain[4] == 0, chain
    var = rows['gpmpc::vargemm_persist_kernel']
    assert var[0] <= 120 and var[5] >= 4, var
