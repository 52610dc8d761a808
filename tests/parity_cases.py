"""Parity checks of the HIP hot path against the oracle, shared by two tiers:

  * tests/test_gpu_parity.py  (-m gpu): the product library libgpmpc_hip.so on a real MI355X,
    called through the C ABI -- the parity tests proper;
  * tests/test_emu_kernels.py (CPU): the SAME kernel sources run under the HIP emulator
    (tests/emu) at small sizes, so indexing/synchronisation bugs are caught without a GPU.

Tolerances: fp64 throughout.
  L:    ||dL||_F / ||L||_F <= 1e-10   (car fixture, cond(K) up to 7e10: 5e-10 -- numpy re-deriving
        its own saved factor on another LAPACK build is already at 8e-11)
  mean: |dmean| / sum_i |ks_i alpha_i| <= 1e-10
  var:  |dvar| / sf^2 <= 1e-10   (plain relative 1e-10 in addition on the sn=0.1 synthetic set)
  NLL:  |dNLL| / (|NLL| + N) <= 1e-10 on well-conditioned data
"""
import numpy as np

import gp_oracle as go
from gp_mpc_amd._lib import Handle, NotPositiveDefinite


def relF(A, B):
    return np.linalg.norm(A - B) / np.linalg.norm(B)


def mean_scale(X, Z, hyper, alpha):
    d = X.shape[1]
    out = np.zeros((Z.shape[0], hyper.shape[0]))
    for a in range(hyper.shape[0]):
        ks = go.cov_se_ard_direct(X, Z, hyper[a, :d], hyper[a, d] ** 2)
        out[:, a] = np.abs(ks).T @ np.abs(alpha[a])
    return out


def check_dgemm(lib, sizes=((70, 33, 50), (128, 128, 64), (200, 130, 96)), seed=0):
    rng = np.random.default_rng(seed)
    for (M, N, K) in sizes:
        for ta in (False, True):
            for tb in (False, True):
                A = rng.standard_normal((K, M) if ta else (M, K))
                B = rng.standard_normal((N, K) if tb else (K, N))   # asymmetric operands: catches transposes
                C0 = rng.standard_normal((M, N))
                ref = 0.7 * (A.T if ta else A) @ (B.T if tb else B) - 0.3 * C0
                C = lib.dgemm(A, B, C0, alpha=0.7, beta=-0.3, transa=ta, transb=tb)
                assert np.abs(C - ref).max() <= 1e-12 * K, (M, N, K, ta, tb)


def check_cholesky(lib, sizes=(64, 100, 192, 256), seed=1):
    rng = np.random.default_rng(seed)
    for n in sizes:
        Q = rng.standard_normal((n, n))
        A = Q @ Q.T + n * np.eye(n)
        L, Li, info = lib.cholesky(A, want_inverse=True)
        Lr = np.linalg.cholesky(A)
        assert info == 0
        assert relF(L, Lr) <= 1e-13
        assert np.all(np.triu(L, 1) == 0.0) and np.all(np.triu(Li, 1) == 0.0)
        assert np.abs(Li @ Lr - np.eye(n)).max() <= 1e-12
    # LAPACK-style info: first non-positive leading minor (1-based)
    A = -np.eye(70)
    A[:30, :30] = 2 * np.eye(30)
    assert lib.cholesky(A)[1] == 31
    A = np.eye(130)
    A[100, 100] = np.nan
    assert lib.cholesky(A)[1] == 101


def check_model_fixture(lib, g, tolL, tol_nll):
    """Saved reference model (tank / car): refit from stored X, hyper and compare with the stored
    chol (written by gp_class.py:693-726) and with the reference's numeric variance GP.covar."""
    X, Y, H = g['X'], g['Y'], g['hyper']
    d = X.shape[1]
    Ny = H.shape[0]
    sf2 = H[:, d] ** 2
    h = Handle(lib, X, Y)
    info = h.fit(H, want_invK=True)
    assert np.all(info == 0)
    f = h.get_factors(invK=True)
    o = go.fit(X, Y, H)
    for a in range(Ny):
        assert relF(f['chol'][a], g['chol'][a]) <= tolL, ('L vs stored', a, relF(f['chol'][a], g['chol'][a]))
        assert relF(f['chol'][a], o['chol'][a]) <= tolL
        assert np.all(np.triu(f['chol'][a], 1) == 0.0)
        K = go.gram(X, H[a, :d], sf2[a], H[a, d + 1] ** 2)
        res = np.linalg.norm(K @ f['alpha'][a] - Y[:, a]) / (np.linalg.norm(K) * np.linalg.norm(f['alpha'][a]))
        assert res <= 1e-13, ('alpha residual', a, res)            # cond-limited quantity -> residual test
        resK = np.linalg.norm(K @ f['invK'][a] - np.eye(len(X))) / (np.linalg.norm(K) * np.linalg.norm(f['invK'][a]))
        assert resK <= 1e-13, ('invK residual', a, resK)
        assert np.allclose(f['invK'][a], f['invK'][a].T, rtol=0, atol=0)
    Z = g['Z']
    mean, var = h.predict_mean_var(Z)
    om, ov, oJ = go.mean_var_jac(Z, X, H, f['alpha'], f['chol'])
    assert np.max(np.abs(mean - om) / mean_scale(X, Z, H, f['alpha'])) <= 1e-10
    assert np.max(np.abs(var - ov) / sf2) <= 1e-10
    assert np.max(np.abs(var.T - g['ref_covar_diag']) / sf2[:, None]) <= 1e-10      # reference GP.covar
    m2, J = h.mean_jac(Z)
    jscale = np.abs(oJ).max(axis=(0, 2), keepdims=True)
    assert np.max(np.abs(J - oJ) / jscale) <= 1e-9
    assert np.array_equal(m2, mean)
    Sig = go.synthetic_problem(8, d, 1, len(Z), seed=5)['Sigma']
    m3, cov = h.predict('TA', Z, Sig)
    oc = go.ta_cov(ov, oJ, Sig)
    assert np.max(np.abs(cov - oc)) <= 1e-9 * np.abs(oc).max()
    m4, cov4 = h.predict('ME', Z)
    assert np.array_equal(np.einsum('baa->ba', cov4), var)
    assert np.all(cov4[:, ~np.eye(Ny, dtype=bool)] == 0.0)
    cv = h.covar(Z[:6])
    assert np.max(np.abs(cv - g['ref_covar'][:, :6, :6]) / sf2[:, None, None]) <= 1e-10
    for a in range(Ny):
        v = h.nll(a, H[a])
        assert abs(v - g['ref_nll'][a]) / (abs(g['ref_nll'][a]) + len(X)) <= tol_nll, (a, v, g['ref_nll'][a])
    # load_model path: stored factors in, no refit (gp_class.py:58-66)
    h2 = Handle(lib, X, Y)
    h2.set_factors(H, g['chol'], g['alpha'], g['invK'])
    mean2, var2 = h2.predict_mean_var(Z)
    om2, ov2, _ = go.mean_var_jac(Z, X, H, g['alpha'], g['chol'], False)
    assert np.max(np.abs(mean2 - om2) / mean_scale(X, Z, H, g['alpha'])) <= 1e-10
    assert np.max(np.abs(var2 - ov2) / sf2) <= 1e-10
    f2 = h2.get_factors(invK=True)
    assert np.array_equal(f2['chol'], g['chol']) and np.array_equal(f2['alpha'], g['alpha'])
    assert np.array_equal(f2['invK'], g['invK'])
    h.close()
    h2.close()


def check_synthetic(lib, N, d, Ny, B, sn, strict_rel):
    """SURVEY 8(d) generator; GPU vs oracle on identical inputs."""
    p = go.synthetic_problem(N, d, Ny, B, seed=1234, sn=sn)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    sf2 = H[:, d] ** 2
    h = Handle(lib, X, Y)
    info = h.fit(H)
    assert np.all(info == 0)
    f = h.get_factors()
    o = go.fit(X, Y, H, want_invK=False)
    for a in range(Ny):
        assert relF(f['chol'][a], o['chol'][a]) <= 1e-10
    mean, var = h.predict_mean_var(Z)
    om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
    assert np.max(np.abs(mean - om) / mean_scale(X, Z, H, o['alpha'])) <= 1e-10
    assert np.max(np.abs(var - ov) / sf2) <= 1e-10
    if strict_rel:
        assert np.max(np.abs(mean - om) / np.abs(om)) <= 1e-10 or np.max(np.abs(mean - om)) <= 1e-10 * np.abs(om).max()
        assert np.max(np.abs(var - ov) / np.abs(ov)) <= 1e-10
        for a in range(Ny):
            v = h.nll(a, H[a])
            ref = go.nll(H[a], X, Y[:, a])
            assert abs(v - ref) / (abs(ref) + N) <= 1e-10
    h.close()
    return dict(X=X, Y=Y, H=H, Z=Z, mean=mean, var=var)


def check_jitter_rule(lib, t):
    """One-shot 1e-8 jitter (optimize.py:345-350): info semantics and NLL on the jittered K."""
    X, Y = t['X'], t['Y']
    h = Handle(lib, X, Y)
    for i, hp in enumerate(t['probes']):
        H = np.tile(hp, (Y.shape[1], 1))
        info = h.fit(H)
        assert np.array_equal(info, t['probe_jitter'][i]), (i, info, t['probe_jitter'][i])
        d = X.shape[1]
        K = go.gram(X, hp[:d], hp[d] ** 2, hp[d + 1] ** 2 + 1e-8 * float(t['probe_jitter'][i].max()))
        tol = max(1e-10, 50 * np.finfo(float).eps * np.linalg.cond(K))   # y^T K^-1 y is cond-limited
        for a in range(Y.shape[1]):
            v = h.nll(a, hp)
            assert h.last_jitter == t['probe_jitter'][i, a]
            assert abs(v - t['probe_nll'][i, a]) <= tol * (abs(t['probe_nll'][i, a]) + len(X))
    # duplicate training rows and (numerically) zero noise: not SPD even after the jitter
    Xd = np.vstack([X[:20], X[:20]])
    Yd = np.vstack([Y[:20], Y[:20]])
    hd = Handle(lib, Xd, Yd)
    bad = np.tile(np.array([1.0, 1.0, 1.0, 1e-12]), (Y.shape[1], 1))
    bad[:, 2] = 1e6          # sf^2 = 1e12 dwarfs the 1e-8 jitter
    try:
        hd.fit(bad)
        raised = False
    except NotPositiveDefinite:
        raised = True
    assert raised and np.all(hd.info < 0)
    h.close()
    hd.close()


def check_nll_gradient(lib, g):
    X, Y = g['X'], g['Y']
    h = Handle(lib, X, Y)
    hp = np.array([12.0, 25.0, 14.0, 18.0, 22.0, 27.0, 2.1, 0.05])
    v, grad = h.nll(1, hp, want_grad=True)
    ov, og = go.nll_grad(hp, X, Y[:, 1])
    assert abs(v - ov) / (abs(ov) + len(X)) <= 1e-10
    assert np.max(np.abs(grad - og) / (np.abs(og) + 1e-3 * np.abs(og).max())) <= 1e-6
    for i in range(len(hp)):      # and against central differences of the reference-pinned NLL
        e = np.zeros_like(hp)
        e[i] = 1e-5 * max(1.0, abs(hp[i]))
        fd = (go.nll(hp + e, X, Y[:, 1]) - go.nll(hp - e, X, Y[:, 1])) / (2 * e[i])
        assert abs(fd - grad[i]) <= 1e-4 * (abs(grad[i]) + 1e-3)
    h.close()
