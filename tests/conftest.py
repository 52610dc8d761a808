"""pytest configuration: `gpu` marker, import paths, golden-fixture helpers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def unpack_lower(P, N, symmetric=False):
    """[Ny, N(N+1)/2] packed row-major lower triangle -> [Ny, N, N]."""
    il = np.tril_indices(N)
    out = np.zeros((P.shape[0], N, N))
    for a in range(P.shape[0]):
        out[a][il] = P[a]
        if symmetric:
            out[a] = out[a] + np.tril(out[a], -1).T
    return out


def load_model(name):
    g = dict(np.load(os.path.join(GOLDEN, f'{name}_model.npz')))
    N = g['X'].shape[0]
    g['chol'] = unpack_lower(g['chol_packed'], N)
    g['invK'] = unpack_lower(g['invK_packed'], N, symmetric=True)
    g['ref_K'] = unpack_lower(g['ref_K_packed'], N, symmetric=True)
    g['normalize'] = bool(g['normalize'])
    if g['normalize']:
        g['meta'] = {k[5:]: v for k, v in g.items() if k.startswith('meta_')}
    return g


@pytest.fixture(scope='session')
def tank():
    return load_model('tank')


@pytest.fixture(scope='session')
def car():
    return load_model('car')


@pytest.fixture(scope='session')
def old_me_pins():
    """Reference-made 'old_ME' outputs (oracle/make_golden.py legacy_pin) at the test points of the two saved models."""
    return {name: dict(np.load(os.path.join(GOLDEN, f'{name}_old_me.npz'))) for name in ('tank', 'car')}


@pytest.fixture(scope='session')
def train_small():
    return dict(np.load(os.path.join(GOLDEN, 'train_small.npz')))


@pytest.fixture(scope='session')
def train_small2():
    return dict(np.load(os.path.join(GOLDEN, 'train_small2.npz')))


@pytest.fixture(scope='session')
def train_small3():
    return dict(np.load(os.path.join(GOLDEN, 'train_small3.npz')))


@pytest.fixture(scope='session')
def ta_pins():
    """Reference-run J / TA outputs (oracle/make_golden.py ta_pin) at the test points of the two saved models."""
    return {name: dict(np.load(os.path.join(GOLDEN, f'{name}_ta.npz'))) for name in ('tank', 'car')}


@pytest.fixture(scope='session')
def em_pins():
    """Reference-run exact moments (oracle/make_golden.py em_pin): model (reference-trained) + quadrature of the reference's
    own predictor, for the ill-conditioned train_small model and the well-conditioned em_model2."""
    return {name: (dict(np.load(os.path.join(GOLDEN, f'{name}.npz'))), dict(np.load(os.path.join(GOLDEN, f'{name}_em.npz'))))
            for name in ('train_small', 'em_model2')}


@pytest.fixture(scope='session')
def ref_written():
    """The model file the reference's own save_model wrote (oracle/make_golden.py ref_written_model) + its outputs."""
    return os.path.join(GOLDEN, 'ref_written_model'), dict(np.load(os.path.join(GOLDEN, 'ref_written_model_outputs.npz')))
