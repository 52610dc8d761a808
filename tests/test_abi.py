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
    assert lib.dll.gpmpc_abi_version() == 1
    n = lib.device_count()
    if n == 0:                           # build container: creating a model must fail loudly, not fall back
        import numpy as np
        with pytest.raises(_lib.GpmpcError) as e:
            _lib.Handle(lib, np.zeros((4, 2)), np.zeros((4, 1)))
        assert e.value.code == _lib.EHIP and 'HIP device' in str(e.value)


def test_gfx950_code_object_present(built):
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', built], capture_output=True, text=True)
    txt = out.stdout + out.stderr
    assert 'gfx950' in txt


def test_no_oracle_or_cpu_fallback_in_product():
    """The shipped package must not import the oracle or the emulator."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'gp_mpc_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.inl')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'gp_oracle' not in src and 'import oracle' not in src, f
                if f.endswith('.py'):
                    assert 'libgpmpc_emu' not in src, f
