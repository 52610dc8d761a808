"""GPU tier (-m gpu): BASELINE.json's configs C3, C4 and C5 end to end through the C ABI.

  C3  6-output GP, N = 8192, d = 8: 30-step uncertainty propagation (ME / TA / EM) -- `gpmpc_rollout` against
      the host-stepped `gpmpc_predict` loop of gp_class.py:777-804 at full size, and against the ORACLE at a size the
      oracle affords (N = 1024, same Ny / d / T, sn = 0.1).
  C4  N = 4096, d = 6: log-marginal likelihood + gradient, seeded random restarts (optimize.py:433-474), world = 1
      here and a 2-process RCCL variant that runs as soon as a box has two GPUs.
  C5  the call pattern of a CasADi Callback inside IPOPT: Nt = 30 shooting nodes per call, value + mean Jacobian +
      TA covariance (`gpmpc_predict_jac`), on the reference's car model and at C3 size.
The bodies live in tests/parity_cases.py; the emulator tier runs them at toy sizes.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import gp_oracle as go
import parity_cases as pc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def lib():
    from gp_mpc_amd._lib import get_lib
    return get_lib()


@pytest.fixture(scope='module')
def c3(lib):
    """The C3 model (fit once per module: 6 x 8192^2 factors + K^-1)."""
    from gp_mpc_amd._lib import Handle
    p = go.synthetic_problem(8192, 8, 6, 30, seed=1234, sn=1e-2)
    h = Handle(lib, p['X'], p['Y'])
    assert np.all(h.fit(p['hyper'], want_invK=True) == 0)
    yield p, h
    h.close()


def test_c3_rollout_horizon_30_full_size(c3):
    p, h = c3
    pc.check_rollout_vs_host_loop(h, p, Ny=6, d=8, T=30)


def test_c3_rollout_vs_oracle_n1024(lib):
    pc.check_rollout_vs_oracle(lib, N=1024, Ny=6, d=8, T=30)


def test_c3_full_size_step_vs_oracle(c3):
    """N = 8192, d = 8: outputs 1 and 4 of the six, one node, ME / TA / EM against the oracle's own fit of those outputs."""
    import time
    p, h = c3
    t0 = time.time()
    r = pc.check_c3_size_step(h, p, outs=(1, 4), node=3)
    print(f'\n[C3] full-size oracle step took {time.time() - t0:.1f} s; EM bar {r["bar_em"]:.2e}')


def test_c5_pattern_car_fixture(lib, car):
    """Nt = 30 nodes per call on the reference's own car model (N = 200, Ny = 3, d = 5)."""
    from gp_mpc_amd._lib import Handle
    g = car
    X, d = g['X'], g['X'].shape[1]
    h = Handle(lib, X, g['Y'])
    h.set_factors(g['hyper'], g['chol'], g['alpha'], g['invK'])
    rng = np.random.default_rng(3)
    Z = X[rng.integers(0, len(X), 30)] + 0.05 * rng.standard_normal((30, d))
    S = go.synthetic_problem(8, d, 1, 30, seed=6)['Sigma']
    pc.check_callback_pattern(h, X, g['hyper'], g['alpha'], g['chol'], Z, S)
    h.close()


def test_c5_pattern_c3_size(c3):
    """The same call at C3 size against the oracle evaluated on the device's own factors (what `build_gp` would be
    handed), plus the factorisation residual of one output at full size."""
    p, h = c3
    d = 8
    X, H = p['X'], p['hyper']
    f = h.get_factors(chol=True, alpha=True)
    pc.check_callback_pattern(h, X, H, f['alpha'], f['chol'], p['Z'][:30], p['Sigma'][:30], repeats=2)
    a = 4
    K = go.gram(X, H[a, :d], H[a, d] ** 2, H[a, d + 1] ** 2)
    L = f['chol'][a]
    assert np.linalg.norm(L @ L.T - K) / np.linalg.norm(K) <= 1e-14
    assert np.linalg.norm(K @ f['alpha'][a] - p['Y'][:, a]) / (np.linalg.norm(K) * np.linalg.norm(f['alpha'][a])) <= 1e-13


def test_c4_random_restarts_world1(lib):
    p = go.synthetic_problem(4096, 6, 1, 1, seed=1234, sn=1e-2)
    pc.check_random_restarts(lib, p['X'], p['Y'], multistart=16, maxiter=3)


def test_c4_lockstep_batches_are_composition_independent(lib):
    """C4 size: 8 seeded restarts x 3 iterations as batches of 8, of 3 and one point at a time: the same table bit for bit."""
    pc.check_train_lockstep_invariance(lib, N=4096, d=6, nstart=8, max_iter=3, seed=1234)


@pytest.mark.parametrize('optimizer', ['scipy', 'native'])
def test_c4_restart_shard_rccl_two_gpus(lib, tmp_path, optimizer):
    """The RCCL branches of the restart shard -- train.py `_all_gather_rows` on backend nccl ('scipy') and the
    library's own ncclAllGather inside `gpmpc_train_multistart` ('native'): two processes, one GPU each, must
    reproduce the single-process result bitwise.  Skipped on a one-GPU box."""
    if lib.device_count() < 2:
        pytest.skip('needs two GPUs (the driver runs the scaling bench on an 8-GPU node)')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    outs = {}
    for world, port in ((1, 29621), (2, 29622)):
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker_gpu.py'), str(tmp_path), '8', optimizer],
                                  env=dict(env, MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r)))
                 for r in range(world)]
        for pr in procs:
            assert pr.wait(timeout=900) == 0
        outs[world] = [np.load(tmp_path / f'gpu_{optimizer}_rank{r}_of{world}.npz') for r in range(world)]
    one, (r0, r1) = outs[1][0], outs[2]
    for k in ('hyper', 'obj', 'alpha'):
        assert np.array_equal(one[k], r0[k]) and np.array_equal(r0[k], r1[k]), k
    if optimizer == 'scipy':
        assert int(r0['n_eval']) + int(r1['n_eval']) == int(one['n_eval'])


def test_c4_native_training_with_rccl_self_gather(lib):
    """`gpmpc_train_multistart` at C4 size on one GPU with a world = 1 RCCL communicator: the (NLL, theta) table goes
    through ncclAllGather (a self-gather), so librccl's run-time binding and the exchange code execute on every box;
    the result must equal the run without a communicator bitwise, and NLL* the oracle's value at theta*."""
    from gp_mpc_amd._lib import Handle
    from gp_mpc_amd.train import lhs_starts, bounds_ipopt_path
    p = go.synthetic_problem(4096, 6, 1, 1, seed=1234, sn=1e-2)
    X, Y = p['X'], p['Y']
    N, d = X.shape
    lb, ub = bounds_ipopt_path(d)
    starts = lhs_starts(8, lb, ub, 1234)[None]
    res = []
    for use_comm in (False, True):
        h = Handle(lib, X, Y)
        comm = lib.rccl_comm_create(0, 1, 0, lib.rccl_unique_id()) if use_comm else None
        try:
            res.append(h.train_multistart(starts, lb[None], ub[None], max_iter=4, comm=comm))
        finally:
            if comm is not None:
                lib.rccl_comm_destroy(comm)
        if use_comm:
            th = res[-1]['hyper'][0]
            best = float(np.min(res[-1]['obj'][0]))
            ref = go.nll(th, X, Y[:, 0])
            tol = max(1e-10, 50 * np.finfo(float).eps * N * (th[d] ** 2 + th[d + 1] ** 2) / th[d + 1] ** 2)
            assert abs(best - ref) <= tol * (abs(ref) + N), (best, ref, tol)
            assert abs(h.nll(0, th) - best) <= 0.1 * tol * (abs(best) + N)    # (gpmpc_nll: single-matrix execution; the search: batched)
        h.close()
    for k in ('hyper', 'obj', 'theta'):
        assert np.array_equal(res[0][k], res[1][k]), k


def test_two_handles_two_threads(lib):
    pc.check_two_handles_two_threads(lib, N=4096)
