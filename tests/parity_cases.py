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
import ctypes
import os

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


def check_dgemm_large_tile(lib, seed=5):
    """The 128 x 128 and 64 x 64 tile kernels (DMA-staged operand tiles) in all four operand orientations on
    ragged shapes; GPMPC_DGEMM_TILE pins the tile so that small matrices reach them."""
    import os
    rng = np.random.default_rng(seed)
    try:
        for tile in ('128', '64'):
            os.environ['GPMPC_DGEMM_TILE'] = tile
            for (M, N, K) in ((128, 128, 16), (200, 130, 96), (257, 300, 48), (90, 513, 160)):
                for ta, tb in ((False, True), (False, False), (True, True), (True, False)):
                    A = rng.standard_normal((K, M) if ta else (M, K))
                    B = rng.standard_normal((N, K) if tb else (K, N))
                    C0 = rng.standard_normal((M, N))
                    ref = 1.3 * (A.T if ta else A) @ (B.T if tb else B) + 0.4 * C0
                    C = lib.dgemm(A, B, C0, alpha=1.3, beta=0.4, transa=ta, transb=tb)
                    assert np.abs(C - ref).max() <= 1e-12 * K, (tile, M, N, K, ta, tb)
    finally:
        del os.environ['GPMPC_DGEMM_TILE']


def check_forced_tiles(lib, tank):
    """The large-tile GEMM kernels (DMA-staged) with everything the factorisation asks of them -- triangular K ranges,
    lower-only output, transposed operands, batched nodes of the inverse tree -- on problems small enough for the
    emulator: pin the tile and run the Cholesky / inverse and a model fit with K^-1."""
    try:
        for tile in (64, 128):
            lib.set_tuning('gemm_tile', tile)
            check_cholesky(lib, sizes=(192, 320))
            check_model_fixture(lib, tank, tolL=1e-10, tol_nll=1e-10)
    finally:
        lib.set_tuning('gemm_tile', 0)


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


def mean_bars(mean, om, floor=1e-2):
    """The two plain-relative measures of a predicted mean against the oracle's (per output column, worst column)."""
    mean, om = np.asarray(mean).reshape(len(om), -1), np.asarray(om).reshape(len(om), -1)
    dm, big = np.abs(mean - om), np.abs(om).max(axis=0, keepdims=True)
    return {'mean_rel_to_max': float((dm / big).max()),
            'mean_pointwise_floored': float((dm / np.maximum(np.abs(om), floor * big)).max()),
            'mean_pointwise_raw': float((dm / np.maximum(np.abs(om), 1e-300)).max())}


def longdouble_alpha(X, y, hyper_row, iters=4):
    """alpha = K^-1 y to extended precision (x87 80-bit, eps 1.1e-19; TEST ONLY): the exact SE-ARD kernel in longdouble
    (direct differences), the fp64 Cholesky of its rounding as the preconditioner, `iters` steps of iterative refinement
    with the residual y - K alpha formed in longdouble -- every step gains ~ -log10(cond(K) eps64) digits (cond <= 1e8
    here: >= 8).  Returns (alpha, K) as longdouble arrays."""
    from scipy.linalg import cho_factor, cho_solve
    ld = np.longdouble
    d = X.shape[1]
    Xl = X.astype(ld) / hyper_row[:d].astype(ld)
    D = np.zeros((len(X), len(X)), dtype=ld)
    for k in range(d):
        diff = Xl[:, k:k + 1] - Xl[:, k][None, :]
        D += diff * diff
    K = ld(hyper_row[d]) ** 2 * np.exp(ld(-0.5) * D)
    K[np.diag_indices_from(K)] += ld(hyper_row[d + 1]) ** 2
    cf = cho_factor(K.astype(np.float64), lower=True)
    yl = y.astype(ld)
    alpha = cho_solve(cf, y).astype(ld)
    for _ in range(iters):
        r = yl - K @ alpha
        alpha = alpha + cho_solve(cf, r.astype(np.float64)).astype(ld)
    return alpha, K


def longdouble_mean(X, hyper_row, alpha_ld, Z):
    """ks(z)^T alpha in longdouble for the rows of Z."""
    ld = np.longdouble
    d = X.shape[1]
    Xl, Zl = X.astype(ld) / hyper_row[:d].astype(ld), Z.astype(ld) / hyper_row[:d].astype(ld)
    out = np.zeros(len(Z), dtype=ld)
    for b0 in range(0, len(Z), 256):
        Zb = Zl[b0:b0 + 256]
        D = np.zeros((len(X), len(Zb)), dtype=ld)
        for k in range(d):
            diff = Xl[:, k:k + 1] - Zb[:, k][None, :]
            D += diff * diff
        out[b0:b0 + 256] = (ld(hyper_row[d]) ** 2 * np.exp(ld(-0.5) * D)).T @ alpha_ld
    return out


def check_mean_against_extended_precision(lib, N, d, B, sn, nprobe=1000, seed=1234):
    """VERDICT r05 'mean digits': is the device's mean as close to the TRUE value as numpy's?  Truth = longdouble kernel,
    alpha refined to longdouble accuracy (longdouble_alpha), ks^T alpha in longdouble, at the first `nprobe` test points of the
    C2 generator.  Gate: the device's error (maximum and rms over the points) is at most TWICE the fp64 oracle's own or 1e-10
    max|mean|; the pointwise figures with the floor of 1e-3 max|mean| r04 asked for are printed for both (see below why not gated).  (x87 longdouble has 11 more mantissa bits than fp64: the
    yardstick is ~2000 x finer than what it measures, no more.)  What separates fp64 results from the truth here is the
    rounding of K's entries (expanded form, a1) amplified by cond(K): the same for both, which is the point."""
    p = go.synthetic_problem(N, d, 1, B, seed=seed, sn=sn)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z'][:nprobe]
    al, _ = longdouble_alpha(X, Y[:, 0], H[0])
    truth = longdouble_mean(X, H[0], al, Z).astype(np.float64)
    o = go.fit(X, Y, H, want_invK=False)
    om, _, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
    h = Handle(lib, X, Y)
    h.fit(H)
    gm_big, _ = h.predict_mean_var(p['Z'])              # the 10 000-point route (mean from the variance product's fused sums)
    gm_small, _ = h.predict_mean_var(Z[:40])            # the small-batch route (ks^T alpha)
    h.close()
    big = np.abs(truth).max()
    floor = np.maximum(np.abs(truth), 1e-3 * big)
    e_gpu = np.abs(gm_big[:nprobe, 0] - truth)
    e_small = np.abs(gm_small[:, 0] - truth[:40])
    e_orc = np.abs(om[:, 0] - truth)
    print(f'[mean vs longdouble truth, N={N} sn={sn}] max|err|/max|mean|: device {e_gpu.max() / big:.2e} (small batch {e_small.max() / big:.2e}) '
          f'oracle {e_orc.max() / big:.2e};  pointwise with floor 1e-3: device {(e_gpu / floor).max():.2e} oracle {(e_orc / floor).max():.2e}; '
          f'device vs oracle {np.abs(gm_big[:nprobe, 0] - om[:, 0]).max() / big:.2e}')
    # ("|gpu - truth| <= 2 |oracle - truth| (or 1e-10 of the stated scale)").  Gated on the error's maximum and
    # root mean square over the probe points; the floored pointwise figures are printed for both sides but NOT gated: where
    # |mean| < 1e-2 max|mean| they are the ratio of an absolute rounding error (~1e-11 max|mean| on either side, independent
    # between the two) to a small number -- first MI355X run, sn = 1e-2: device 3.4e-9 / oracle 1.2e-9 pointwise, while the
    # device's largest error (8.0e-12 max|mean|) is HALF the oracle's (1.6e-11).
    rms = lambda e: float(np.sqrt(np.mean(e * e)))
    print(f'   rms error / max|mean|: device {rms(e_gpu) / big:.2e} oracle {rms(e_orc) / big:.2e}')
    assert e_gpu.max() <= max(2.0 * e_orc.max(), 1e-10 * big) and e_small.max() <= max(2.0 * e_orc.max(), 1e-10 * big)
    assert rms(e_gpu) <= max(2.0 * rms(e_orc), 1e-10 * big)
    return dict(device=float((e_gpu / floor).max()), oracle=float((e_orc / floor).max()))


def check_synthetic(lib, N, d, Ny, B, sn, strict_rel):
    """SURVEY 8(d) generator; GPU vs oracle on identical inputs."""
    p = go.synthetic_problem(N, d, Ny, B, seed=1234, sn=sn)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    sf2 = H[:, d] ** 2
    h = Handle(lib, X, Y)
    info = h.fit(H)
    assert np.all(info == 0)
    # (a hand-off that gives up falls back to the single-queue path with the same results: it must not hide behind them)
    if 'GPMPC_SPIN_LIMIT' not in os.environ:
        assert h.counter('handoff_timeouts') == 0, h.counter('handoff_timeouts')
    f = h.get_factors()
    o = go.fit(X, Y, H, want_invK=False)
    for a in range(Ny):
        assert relF(f['chol'][a], o['chol'][a]) <= 1e-10
    mean, var = h.predict_mean_var(Z)
    om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
    assert np.max(np.abs(mean - om) / mean_scale(X, Z, H, o['alpha'])) <= 1e-10
    assert np.max(np.abs(var - ov) / sf2) <= 1e-10
    if strict_rel:
        # north_star's "1e-10 rel" on the well-conditioned set, as TWO named bars that are BOTH gated (r04: an `or`):
        #   mean_rel_to_max         max_i |dmean_i| / max_j |mean_j|                               <= 1e-10
        #   mean_pointwise_floored  max_i |dmean_i| / max(|mean_i|, 1e-2 max_j |mean_j|)            <= 1e-10
        # (a mean that crosses zero has no pointwise relative error below its own size: the floor says from which size on
        #  the pointwise figure is asked for.  Why one hundredth of the largest mean and not less: the mean is a sum of N
        #  terms ks_i alpha_i whose absolute values add up to ~1e3 max|mean| on this set, so two correctly rounded fp64
        #  evaluations in different summation orders -- numpy's and the device's -- differ by ~5e-13 max|mean| in absolute
        #  terms (measured on MI355X at N = 1024 / 4096: rel_to_max 5.4e-13 / 3.7e-13, r05 call 1); with a floor of 1e-3 the
        #  same run sits at 1.004e-10, i.e. the bar would gate the summation order, not the kernels); the variance is
        #  bounded away from 0 (>= sf^2 - ks^T K^-1 ks > 0) and takes the plain pointwise bar.
        bars = mean_bars(mean, om)
        assert bars['mean_rel_to_max'] <= 1e-10, bars
        assert bars['mean_pointwise_floored'] <= 1e-10, bars
        assert np.max(np.abs(var - ov) / np.abs(ov)) <= 1e-10
        for a in range(Ny):
            v = h.nll(a, H[a])
            ref = go.nll(H[a], X, Y[:, a])
            assert abs(v - ref) / (abs(ref) + N) <= 1e-10
    h.close()
    return dict(X=X, Y=Y, H=H, Z=Z, mean=mean, var=var)


def check_variance_persistent(lib, N, d, Ny, B, sn=0.1, seed=4321):
    """The persistent variance product (vargemm_persist.hpp: resident workgroups walking a static tile schedule, the
    slabs of a workgroup's tiles as one stream) against the one-tile-per-workgroup launch of the same tiles: the same bits
    (same per-tile arithmetic, same per-tile partial sums), and against the oracle's variance (gp_functions.py:122-126)."""
    p = go.synthetic_problem(N, d, Ny, B, seed=seed, sn=sn)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    sf2 = H[:, d] ** 2
    h = Handle(lib, X, Y)
    try:
        lib.set_tuning('gemm_tile', 128)
        assert np.all(h.fit(H) == 0)
        lib.set_tuning('vargemm_persist', 0)
        m0, v0 = h.predict_mean_var(Z)
        assert h.counter('persistent_variance_products') == 0
        lib.set_tuning('vargemm_persist', 2)
        m1, v1 = h.predict_mean_var(Z)
        assert h.counter('persistent_variance_products') >= 1
        m2, v2 = h.predict_mean_var(Z[: max(1, B // 2) + 70])      # another shape: the schedule is rebuilt
        # the variance: same bits; the mean comes out of the persistent kernel's fused reduction (L^-1 ks)^T (L^-1 y) instead
        # of ks^T alpha: the same number, another summation
        assert np.array_equal(v0, v1)
        assert np.array_equal(v2, v0[: len(v2)])
        sc = mean_scale(X, Z, H, go.fit(X, Y, H, want_invK=False)['alpha'])
        assert np.max(np.abs(m1 - m0) / sc) <= 1e-12 and np.max(np.abs(m2 - m0[: len(m2)]) / sc[: len(m2)]) <= 1e-12
        o = go.fit(X, Y, H, want_invK=False)
        om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
        assert np.max(np.abs(v1 - ov) / sf2) <= 1e-10
        assert np.max(np.abs(v1 - ov) / np.abs(ov)) <= 1e-10
        assert np.max(np.abs(m1 - om) / sc) <= 1e-10 and np.max(np.abs(m0 - om) / sc) <= 1e-10
    finally:
        lib.set_tuning('gemm_tile', 0)
        lib.set_tuning('vargemm_persist', -1)
        h.close()


def check_set_factors_persistent_mean(lib, N=300, d=4, Ny=1, B=200, seed=77):
    """load_model / checkpoint path (gp_class.py:58-66: hyper, chol, alpha handed over, no training) followed by a large
    batch that takes the persistent variance product: its fused mean reads w = L^-1 y, which `gpmpc_set_factors` must form
    from the imported factor even when the caller supplies alpha (r04 advisor finding: an all-zero mean)."""
    p = go.synthetic_problem(N, d, Ny, B, seed=seed, sn=0.1)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    o = go.fit(X, Y, H, want_invK=False)
    om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
    sc = mean_scale(X, Z, H, o['alpha'])
    try:
        lib.set_tuning('gemm_tile', 128)
        lib.set_tuning('vargemm_persist', 2)
        for with_alpha in (True, False):
            h = Handle(lib, X, Y)
            h.set_factors(H, o['chol'], o['alpha'] if with_alpha else None)
            m, v = h.predict_mean_var(Z)
            assert h.counter('persistent_variance_products') >= 1
            assert np.max(np.abs(m - om) / sc) <= 1e-10, (with_alpha, np.max(np.abs(m - om)), np.abs(om).max())
            assert np.max(np.abs(v - ov) / H[:, d] ** 2) <= 1e-10
            # ... and after a stale w: a fit at other hyper-parameters, then the stored factors again on the same handle
            assert np.all(h.fit(H * 1.3) == 0)
            h.set_factors(H, o['chol'], o['alpha'] if with_alpha else None)
            m, v = h.predict_mean_var(Z)
            assert np.max(np.abs(m - om) / sc) <= 1e-10
            h.close()
    finally:
        lib.set_tuning('gemm_tile', 0)
        lib.set_tuning('vargemm_persist', -1)


class DevArray:
    """A double array in device memory for the device-pointer entry points: hipMalloc / hipMemcpy through the HIP runtime
    the library is linked against (ctypes, no torch: one HIP runtime per process); under the emulator device memory is
    host memory and the array is a numpy array."""

    def __init__(self, lib, src=None, shape=None):
        self.emulated = 'emu' in os.path.basename(lib.path)
        self.shape = tuple(np.shape(src)) if src is not None else tuple(shape)
        self.nbytes = int(np.prod(self.shape)) * 8
        if self.emulated:
            self.host = np.ascontiguousarray(src, dtype=np.float64).copy() if src is not None else np.zeros(self.shape)
            self.ptr = self.host.ctypes.data
            return
        # the runtime the LIBRARY runs on (gpmpc_runtime_info), not whatever 'libamdhip64.so' resolves to: in a process
        # that imported torch earlier (an earlier test of the same pytest run) the library lives on torch's bundled
        # runtime and the bare name would open /opt/rocm's as a second, uninitialised one (hipMalloc fails there)
        self.hip = ctypes.CDLL(lib.runtime_info().get('hip_path', 'libamdhip64.so'))
        p = ctypes.c_void_p()
        assert self.hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(self.nbytes)) == 0
        self.ptr = p.value
        if src is not None:
            a = np.ascontiguousarray(src, dtype=np.float64)
            assert self.hip.hipMemcpy(ctypes.c_void_p(self.ptr), a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(self.nbytes), 1) == 0

    def numpy(self):
        if self.emulated:
            return self.host.copy()
        out = np.empty(self.shape)
        assert self.hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr), ctypes.c_size_t(self.nbytes), 2) == 0
        return out

    def free(self):
        if not self.emulated and self.ptr:
            self.hip.hipFree(ctypes.c_void_p(self.ptr))
            self.ptr = None


def check_predict_behind_tail(lib, N, d, B, sn=0.1, seed=77, strict=True, expect_overlap=True, repeats=2, mean_only_second=False):
    """The first large mean + variance prediction behind a fit runs next to the tail of the triangular inverse
    (api_predict.inl predict_chunk, `behind_tail`: the fit returns when its chain kernel ends; cross-covariances on the
    low-priority queue, alpha and the mean on the workers' queue, the variance product right behind the tail).  Device
    pointers, one output.  Against the oracle, and against the same library's plain route (a second prediction without
    a fit in between runs entirely on the main queue): the same bits for the variance (same tiles, same per-tile partial
    sums) and the mean (same summation order).  mean_only_second: the follow-up call asks for the mean alone -- it must
    order itself behind alpha, which the first call (variance only) left pending on the workers' queue."""
    p = go.synthetic_problem(N, d, 1, B, seed=seed, sn=sn)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    h = Handle(lib, X, Y)
    h.set_pointer_mode(True)
    z = DevArray(lib, Z)
    m1, v1 = DevArray(lib, shape=(B, 1)), DevArray(lib, shape=(B, 1))
    m2, v2 = DevArray(lib, shape=(B, 1)), DevArray(lib, shape=(B, 1))
    o = go.fit(X, Y, H, want_invK=False)
    om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
    for rep in range(repeats):
        assert np.all(h.fit(H) == 0)
        if mean_only_second:
            h.predict_mean_var_dev(B, z.ptr, None, v1.ptr)        # behind the tail, variance only: alpha stays pending
            h.predict_mean_var_dev(B, z.ptr, m1.ptr, None)        # plain route, reads alpha
        else:
            h.predict_mean_var_dev(B, z.ptr, m1.ptr, v1.ptr)      # behind the tail
        h.predict_mean_var_dev(B, z.ptr, m2.ptr, v2.ptr)          # plain route
        h.synchronize()
        mean, var, mean2, var2 = m1.numpy(), v1.numpy(), m2.numpy(), v2.numpy()
        assert np.max(np.abs(mean - om) / mean_scale(X, Z, H, o['alpha'])) <= 1e-10
        assert np.max(np.abs(var - ov) / H[:, d] ** 2) <= 1e-10
        if strict:
            assert np.max(np.abs(var - ov) / np.abs(ov)) <= 1e-10
        assert np.array_equal(var, var2), np.abs(var - var2).max()
        assert np.max(np.abs(mean - mean2)) <= 1e-13 * np.abs(om).max()
    if expect_overlap is not None:
        assert (h.counter('predictions_behind_tail') == repeats) == bool(expect_overlap), h.counter('predictions_behind_tail')
    f = h.get_factors()
    assert relF(f['chol'][0], o['chol'][0]) <= 1e-10
    assert np.max(np.abs(f['alpha'][0] - o['alpha'][0])) <= 1e-9 * np.abs(o["alpha"][0]).max() * (0.1 / sn) ** 2
    for a in (z, m1, v1, m2, v2):
        a.free()
    h.close()


def check_fused_fit_predict(lib, N, d, B, sn=0.1, seed=91, Ny=1, repeats=2, expect_fused=True, jitter_case=False):
    """gpmpc_fit_predict_mean_var (r06: the prediction enqueued before the host waits for the fit's status, cross-covariances
    inside the chain's window) against the two calls it replaces on the same handle: bitwise the same mean and variance, the
    same factors; and against the oracle.  jitter_case: ell = 8, sn = 1e-9 -- the first attempt fails, the fused
    prediction is repeated behind the repeated factorisation (info = 1)."""
    p = go.synthetic_problem(N, d, Ny, B, seed=seed, sn=sn)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    if jitter_case:                      # clearly not positive definite in fp64, clearly so with 1e-8 I (as tests/golden's probe 4)
        H = H.copy()
        H[:, :d] = 8.0
        H[:, d + 1] = 1e-9
    h = Handle(lib, X, Y)
    h.set_pointer_mode(True)
    z = DevArray(lib, Z)
    m1, v1 = DevArray(lib, shape=(B, Ny)), DevArray(lib, shape=(B, Ny))
    m2, v2 = DevArray(lib, shape=(B, Ny)), DevArray(lib, shape=(B, Ny))
    for rep in range(repeats):
        i2 = h.fit(H)
        h.predict_mean_var_dev(B, z.ptr, m2.ptr, v2.ptr)
        h.synchronize()
        f2 = h.get_factors()
        i1 = h.fit_predict_mean_var_dev(H, B, z.ptr, m1.ptr, v1.ptr)
        h.synchronize()
        f1 = h.get_factors()
        assert np.array_equal(i1, i2) and np.all(i1 == (1 if jitter_case else 0)), (i1, i2)
        assert np.array_equal(m1.numpy(), m2.numpy()), np.abs(m1.numpy() - m2.numpy()).max()
        assert np.array_equal(v1.numpy(), v2.numpy()), np.abs(v1.numpy() - v2.numpy()).max()
        assert np.array_equal(f1['chol'], f2['chol']) and np.array_equal(f1['alpha'], f2['alpha'])
    if expect_fused is not None:
        assert (h.counter('fused_fit_predicts') == repeats) == bool(expect_fused), h.counter('fused_fit_predicts')
    if not jitter_case:
        o = go.fit(X, Y, H, want_invK=False)
        om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
        mean, var = m1.numpy(), v1.numpy()
        for a in range(Ny):
            assert np.max(np.abs(mean[:, a] - om[:, a]) / mean_scale(X, Z, H[a:a + 1], o['alpha'][a:a + 1])) <= 1e-10
        assert np.max(np.abs(var - ov) / H[:, d] ** 2) <= 1e-10
    # host pointers: the same entry point runs the two calls (no device buffers to stage)
    h.set_pointer_mode(False)
    i3, m3, v3 = h.fit_predict_mean_var(H, Z)
    # (host-pointer predictions take their mean from the cross-covariance kernel's ks^T alpha, the large device-pointer route
    #  from the variance product's sum_i V_ij w_i: two summation orders of the same N-term sum)
    assert np.array_equal(i3, i1) and np.array_equal(v3, v1.numpy())
    for a in range(Ny):                  # ... compared at the size of that sum's terms
        assert np.max(np.abs(m3[:, a] - m1.numpy()[:, a]) / mean_scale(X, Z, H[a:a + 1], f1['alpha'][a:a + 1])) <= 1e-13
    for a in (z, m1, v1, m2, v2):
        a.free()
    h.close()


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


def _em_scale(invK, X, Y, H, mu, Sigma):
    """Cancellation scale of the exact-moment covariance: t * sum_ij |A_ij Q_ij| per pair
    (gp_functions.py:408-414 sums terms of this size to a result orders of magnitude smaller)."""
    logH = np.log(H)
    Ny, (N, Nx) = len(invK), X.shape
    v = X - mu.reshape(1, Nx)
    beta = np.stack([invK[a] @ Y[:, a] for a in range(Ny)], axis=1)
    log_k = np.stack([2 * logH[a, Nx] - 0.5 * np.sum((v / H[a, :Nx]) ** 2, axis=1) for a in range(Ny)], axis=1)
    scale = np.zeros((Ny, Ny))
    for a in range(Ny):
        ii = v / H[a, :Nx] ** 2
        for b in range(a + 1):
            R = Sigma @ np.diag(1 / H[a, :Nx] ** 2 + 1 / H[b, :Nx] ** 2) + np.eye(Nx)
            t = 1.0 / np.sqrt(abs(np.linalg.det(R)))
            ij = v / H[b, :Nx] ** 2
            Q = np.exp(log_k[:, a][:, None] + log_k[:, b][None, :] + go.maha(ii, -ij, np.linalg.solve(R, Sigma * 0.5)))
            A = np.outer(beta[:, a], beta[:, b])
            if a == b:
                A = A - invK[a]
            scale[a, b] = scale[b, a] = t * np.sum(np.abs(A * Q))
    return scale


def check_moment_methods(lib, g=None):
    """a11 'EM' and a12 'old_ME'/'old_TA' against the oracle restatement (same K^-1 fed to both)."""
    if g is None:
        p = go.synthetic_problem(150, 4, 3, 6, seed=3, sn=0.1)
        X, Y, H, Z, S = p['X'], p['Y'], p['hyper'], p['Z'], p['Sigma'] * 30
        tol = 1e-10
    else:
        X, Y, H, Z = g['X'], g['Y'], g['hyper'], g['Z'][:6]
        S = go.synthetic_problem(8, X.shape[1], 1, 6, seed=5)['Sigma'] * 10
        tol = 1e-10
    d = X.shape[1]
    h = Handle(lib, X, Y)
    h.fit(H, want_invK=True)
    f = h.get_factors(invK=True)
    m, c = h.predict('EM', Z, S)
    m1, c1 = h.predict('old_ME', Z)
    m2, c2 = h.predict('old_TA', Z, S)
    sf2 = H[:, d] ** 2
    for b in range(len(Z)):
        om, oc = go.exact_moment(f['invK'], X, Y, H, Z[b], S[b])
        sc = _em_scale(f['invK'], X, Y, H, Z[b], S[b])
        msc = np.array([np.sum(np.abs(f['invK'][a] @ Y[:, a])) * sf2[a] for a in range(len(H))])
        assert np.max(np.abs(m[b] - om) / msc) <= tol
        assert np.max(np.abs(c[b] - oc) / (sc + sf2.max())) <= 10 * tol, (b, np.abs(c[b] - oc).max(), sc.max())
        assert np.array_equal(c[b], c[b].T)
        o1m, o1c = go.old_me(f['invK'], X, Y, H, Z[b])
        kscale = np.array([np.sum(np.abs(f['invK'][a])) * sf2[a] ** 2 for a in range(len(H))])
        assert np.max(np.abs(m1[b] - o1m) / msc) <= tol
        assert np.max(np.abs(np.diag(c1[b]) - np.diag(o1c)) / kscale) <= tol
        o2m, o2c = go.old_ta(f['invK'], X, Y, H, Z[b], S[b])
        assert np.max(np.abs(m2[b] - o2m) / msc) <= tol
        assert np.max(np.abs(c2[b] - o2c)) <= 1e-6 * max(np.abs(o2c).max(), kscale.max() * 1e-4)
    h.close()


def check_em_chunks(lib, N=300, d=3, Ny=2, seed=12):
    """Exact-moment pair sums with every split of a strip's column sweep (em_kernels.hpp: workgroup = (strip, chunk); chunks
    beyond the diagonal of an a == b strip write zeros, chunks below it count twice): Np = 320 = 5 column tiles, chunk sizes
    1, 2, 3 and the whole strip against the oracle (gp_functions.py:344-418) and against each other at rounding level."""
    p = go.synthetic_problem(N, d, Ny, 3, seed=seed, sn=0.1)
    X, Y, H, Z, S = p['X'], p['Y'], p['hyper'], p['Z'], p['Sigma'] * 30
    h = Handle(lib, X, Y)
    h.fit(H, want_invK=True)
    f = h.get_factors(invK=True)
    sf2 = H[:, d] ** 2
    ref = None
    try:
        # (chunk, ranges per a == b pair: 0 = strips and chunks for those too; 1 / 2 / 4 = one workgroup walks the whole triangle of
        #  15 tiles / ranges that start and end inside strips; 12 with whole strips: more ranges than strip workgroups, the a != b
        #  launch zeroes the slots it does not use; -1 = the default, here one tile per range)
        for chunk, segs in ((1, 0), (2, 0), (3, 0), (1000, 0), (1000, 1), (1000, 2), (2, 4), (1, -1), (3, 7), (1000, 12)):
            lib.set_tuning('em_chunk', chunk)
            lib.set_tuning('em_diag_segs', segs)
            m, c = h.predict('EM', Z, S)
            for b in range(len(Z)):
                om, oc = go.exact_moment(f['invK'], X, Y, H, Z[b], S[b])
                sc = _em_scale(f['invK'], X, Y, H, Z[b], S[b])
                assert np.max(np.abs(c[b] - oc) / (sc + sf2.max())) <= 1e-9, (chunk, segs, b)
                assert np.array_equal(c[b], c[b].T)
            if ref is None:
                ref = c
            else:
                assert np.max(np.abs(c - ref)) <= 1e-12 * max(1.0, np.abs(ref).max()) * 1e3, (chunk, segs)
    finally:
        lib.set_tuning('em_chunk', 0)
        lib.set_tuning('em_diag_segs', -1)
        h.close()


def check_old_me_reference_pin(lib, g, pin, tol=1e-12):
    """a12 'old_ME' (`gp`, gp_functions.py:176-256, alpha=None) against REFERENCE-MADE outputs: the reference's own numpy
    GP.covSEard composed with the K^-1 and Y of its saved model in the graph's order (oracle/make_golden.py legacy_pin).
    The device gets the stored factors (load_model path: gpmpc_set_factors with K^-1) and predicts at the pin's points.
    Bars: |dmean| / (|ks|^T |K^-1| |y|) and |dvar| / (|ks|^T |K^-1| |ks|) <= 1e-12 -- the rounding scale of the two sums
    (cond(K) up to 7e10 on the car model: the raw relative error between the reference's expanded-form ks and the
    direct-difference ks of gp_functions.py:17-22 is already 1e-5 there)."""
    X, Y, H = g['X'], g['Y'], g['hyper']
    h = Handle(lib, X, Y)
    h.set_factors(H, g['chol'], g['alpha'], g['invK'])
    m, c = h.predict('old_ME', pin['Z'])
    h.close()
    var = np.stack([np.diag(cb) for cb in c])                      # [B, Ny]
    em = np.max(np.abs(m - pin['ref_old_me_mean'].T) / pin['mean_scale'].T)
    ev = np.max(np.abs(var - pin['ref_old_me_var'].T) / pin['var_scale'].T)
    assert em <= tol and ev <= tol, (em, ev)
    for cb in c:                                                    # covar = diag(var): gp_functions.py:254
        assert np.array_equal(cb, np.diag(np.diag(cb)))


def check_ta_reference_pin(lib, g, pin, name):
    """a9 mean / J and a10 'TA' against REFERENCE-RUN outputs (oracle/make_golden.py ta_pin: the reference's numpy
    GP.covSEard^T alpha, its complex-step derivative, diag GP.covar, composed by the one line of build_TA_cov
    gp_functions.py:167-171).  The device gets the stored factors (load_model path) and predicts through gpmpc_predict_jac.
    Bars: rounding scale of the sums (1e-13 of sum |ks alpha| resp. sum |ks alpha (x - z)| / l^2), |dvar| <= 1e-10 sf^2,
    and for the matrix MPC factors: |dcov| <= 1e-10 max|cov| (tank, cond 6e7) / 1e-9 (car, cond 7e10)."""
    X, Y, H = g['X'], g['Y'], g['hyper']
    d = X.shape[1]
    h = Handle(lib, X, Y)
    h.set_factors(H, g['chol'], g['alpha'], g['invK'])
    m, c, J = h.predict_jac('TA', pin['Z'], pin['Sigma'])
    m2, c2 = h.predict('TA', pin['Z'], pin['Sigma'])
    mm, cm = h.predict('ME', pin['Z'])
    h.close()
    em = np.max(np.abs(m - pin['ref_mean']) / pin['mean_scale'])
    eJ = np.max(np.abs(J - pin['ref_J']) / pin['J_scale'])
    ev = np.max(np.abs(np.stack([np.diag(x) for x in cm]) - pin['ref_var'])) / (H[:, d] ** 2).max()
    ec = np.max(np.abs(c - pin['ref_ta_cov'])) / np.abs(pin['ref_ta_cov']).max()
    print(f'[TA pin {name}] mean {em:.2e} J {eJ:.2e} (of the sums\' rounding scale)  var/sf2 {ev:.2e}  cov/max|cov| {ec:.2e}')
    assert em <= 1e-13 and eJ <= 1e-13 and ev <= 1e-10, (em, eJ, ev)
    assert ec <= (1e-10 if name == 'tank' else 1e-9), ec
    assert np.array_equal(m, m2) and np.array_equal(c, c2)


def check_em_reference_pin(lib, model, pin, name):
    """a11 'EM' against REFERENCE-RUN moments (oracle/make_golden.py em_pin: Gauss-Hermite quadrature of the reference's own
    numeric predictor on a model its train_gp_numpy produced; factors handed over as the reference returned them).
    em_model2 (cond 4e3): 1e-12 absolute; train_small (cond 7e8): the closed form's K^-1 arithmetic limits every fp64
    evaluation of gp_functions.py:408-414 to ~cond eps sf^2 (the oracle sits 3e-6 from the quadrature there)."""
    tm, tc = {'train_small': (1e-7, 1e-5), 'em_model2': (1e-12, 1e-12)}[name]
    h = Handle(lib, model['X'], model['Y'])
    h.set_factors(model['hyper'], model['chol'], model['alpha'], model['invK'])
    m, c = h.predict('EM', pin['mu'], pin['Sigma'])
    h.close()
    em, ec = np.max(np.abs(m - pin['ref_em_mean'])), np.max(np.abs(c - pin['ref_em_cov']))
    print(f'[EM pin {name}] |dmean| {em:.2e} |dcov| {ec:.2e} (absolute; max|cov| {np.abs(pin["ref_em_cov"]).max():.2e})')
    assert em <= tm and ec <= tc, (em, ec)


def check_reference_written_model(lib, path, out, tmp_path):
    """f2: GP.load_model on a file the REFERENCE's save_model wrote (gp_class.py:693-743), predictions in the model's
    standardised coordinates against the reference's GP.covar / covSEard^T alpha, and our save_model writes the same key
    set with the same values (factors re-exported from the device: exact for what was handed over)."""
    import json
    from gp_mpc_amd.gp import GP
    gp = GP.load_model(path, lib=lib)
    d = json.load(open(path + '.json'))
    H = np.array(d['hyper']['hyper'])
    Ny, Nx = H.shape[0], np.array(d['X']).shape[1]
    assert gp.get_size() == (len(d['X']), Ny, Nx - Ny)
    cv = gp.covar(out['Zs'])
    assert np.max(np.abs(cv[:Ny] - out['ref_covar'])) <= 1e-10 * (H[:, Nx] ** 2).max()
    gp.set_method('ME')
    meta = {k: np.array(v) for k, v in d['meta'].items()}
    Zraw = out['Zs'] * meta['stdZ'] + meta['meanZ']
    alpha = np.array(d['hyper']['alpha'])
    for b in range(len(Zraw)):
        m, c = gp.predict(Zraw[b, :Ny], Zraw[b, Ny:], np.zeros((Nx, Nx)))
        ms = np.array([np.abs(go.cov_se_ard_direct(np.array(d['X']), out['Zs'][b:b + 1], H[a, :Nx], H[a, Nx] ** 2))[:, 0]
                       @ np.abs(alpha[a]) for a in range(Ny)])
        ref = out['ref_mean_std'][b] * meta['stdY'] + meta['meanY']             # inverse_mean, gp_class.py:636-638
        assert np.max(np.abs(m[:, 0] - ref) / (ms * meta['stdY'])) <= 1e-12, (b, m[:, 0], ref)
        assert np.max(np.abs(np.diag(c) - out['ref_covar'][:, b, b])) <= 1e-10 * (H[:, Nx] ** 2).max()
    gp.save_model(str(tmp_path / 'again'))
    d2 = json.load(open(str(tmp_path / 'again') + '.json'))
    assert set(d2) == set(d) and set(d2['hyper']) == set(d['hyper']) and set(d2['meta']) == set(d['meta'])
    for k in ('X', 'Y', 'xlb', 'xub', 'ulb', 'uub', 'mean_func', 'normalize'):
        assert d2[k] == d[k], k
    for k in d['hyper']:
        assert d2['hyper'][k] == d['hyper'][k], k
    for k in d['meta']:
        assert d2['meta'][k] == d['meta'][k], k
    gp.close()


def check_small_batch_chunks(lib, N=600, d=5, Ny=2):
    """Few test points on a larger model: the cross-covariance kernel cuts the training points into chunks and a
    second kernel adds the partial means / Jacobians (the MPC's shooting-node pattern)."""
    p = go.synthetic_problem(N, d, Ny, 64, seed=21, sn=0.1)
    X, Y, H = p['X'], p['Y'], p['hyper']
    h = Handle(lib, X, Y)
    assert np.all(h.fit(H) == 0)
    f = h.get_factors()
    for B in (1, 5, 30, 64):
        Z = p['Z'][:B]
        om, ov, oJ = go.mean_var_jac(Z, X, H, f['alpha'], f['chol'])
        mean, J = h.mean_jac(Z)
        m2, var = h.predict_mean_var(Z)
        ms = mean_scale(X, Z, H, f['alpha'])
        assert np.max(np.abs(mean - om) / ms) <= 1e-10 and np.max(np.abs(m2 - om) / ms) <= 1e-10
        assert np.max(np.abs(var - ov) / H[:, d] ** 2) <= 1e-10
        assert np.max(np.abs(J - oJ) / (ms / H[:, :d].min(axis=1))[..., None]) <= 1e-10
        S = np.tile(np.eye(d) * 1e-3, (B, 1, 1))
        mt, ct = h.predict('TA', Z, S)
        oc = go.ta_cov(ov, oJ, S)
        assert np.max(np.abs(ct - oc)) <= 1e-10 * max(1.0, np.abs(oc).max())
        m3, c3, J3 = h.predict_jac('TA', Z, S)                 # the same in one pass
        assert np.array_equal(m3, mt) and np.array_equal(c3, ct) and np.array_equal(J3, J)
    h.close()


def check_timeout_fallback(lib, N=560, d=4):
    """A hand-off time-out inside the persistent factorisation kernels (forced by a poll budget of 1) must
    leave a correct model behind: the host repeats the factorisation on the single-stream path."""
    import os
    os.environ['GPMPC_SPIN_LIMIT'] = '1'
    try:
        p = go.synthetic_problem(N, d, 1, 20, seed=9, sn=0.1)
        h = Handle(lib, p['X'], p['Y'])
    finally:
        del os.environ['GPMPC_SPIN_LIMIT']
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    for rep in range(5):       # fits 1-3: time-out + repeat on the single-queue path (the chained path is tried again
        assert np.all(h.fit(H) == 0)        # every time); from the third strike on the handle stays single-queue
        assert h.counter('handoff_timeouts') == min(rep + 1, 3), (rep, h.counter('handoff_timeouts'))
        assert h.counter('single_queue_factorisations') == rep + 1 and h.counter('chained_factorisations') == 0
        f = h.get_factors()
        o = go.fit(X, Y, H, want_invK=False)
        assert relF(f['chol'][0], o['chol'][0]) <= 1e-10
        mean, var = h.predict_mean_var(Z)
        om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
        assert np.max(np.abs(mean - om) / mean_scale(X, Z, H, o['alpha'])) <= 1e-10
        assert np.max(np.abs(var - ov) / H[:, d] ** 2) <= 1e-10
    h.close()


def check_append(lib, N0, n, d=4, Ny=2, sn=0.1, seed=5):
    """gpmpc_append (rank-n extension of L, L^-1, alpha with the stored hyper-parameters; the reference's
    update_data_all recomputes from scratch, gp_class.py:474-550) against the oracle's full fit on all points."""
    p = go.synthetic_problem(N0 + n, d, Ny, 25, seed=seed, sn=sn)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    sf2 = H[:, d] ** 2
    h = Handle(lib, X[:N0], Y[:N0])
    assert np.all(h.fit(H) == 0)
    info = h.append(X[N0:], Y[N0:])
    assert np.all(info == 0) and h.N == N0 + n
    f = h.get_factors()
    o = go.fit(X, Y, H, want_invK=False)
    for a in range(Ny):
        assert relF(f['chol'][a], o['chol'][a]) <= 1e-10
        assert np.all(np.triu(f['chol'][a], 1) == 0.0)
    mean, var = h.predict_mean_var(Z)
    om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
    assert np.max(np.abs(mean - om) / mean_scale(X, Z, H, o['alpha'])) <= 1e-10
    assert np.max(np.abs(var - ov) / sf2) <= 1e-10
    # a second append on top, and the NLL path (its own workspace is rebuilt for the new size)
    q = go.synthetic_problem(7, d, Ny, 1, seed=seed + 1, sn=sn)
    h.append(q['X'], q['Y'])
    X2, Y2 = np.vstack([X, q['X']]), np.vstack([Y, q['Y']])
    o2 = go.fit(X2, Y2, H, want_invK=False)
    m2, v2 = h.predict_mean_var(Z)
    om2, ov2, _ = go.mean_var_jac(Z, X2, H, o2['alpha'], o2['chol'], False)
    assert np.max(np.abs(m2 - om2) / mean_scale(X2, Z, H, o2['alpha'])) <= 1e-10 and np.max(np.abs(v2 - ov2) / sf2) <= 1e-10
    ref = go.nll(H[0], X2, Y2[:, 0])
    assert abs(h.nll(0, H[0]) - ref) / (abs(ref) + len(X2)) <= 1e-10
    h.close()


def check_append_series(lib, N0=3000, d=4, seed=21):
    """A run of appends at a size whose N x N blocks (>= 64 MB) go through the size-class free list (gpmpc_api.hip,
    block_alloc): every new workspace after the second is built in blocks the previous append gave back -- while a second
    model keeps fitting and predicting in between, so a block handed out twice would show.  Factors and predictions
    against the oracle's full fit after every step."""
    steps = (64, 10, 64, 30, 20)          # Np = 3072, 3136, 3200, 3200, 3200: one size class (72-82 MB -> 84 MB blocks)
    p = go.synthetic_problem(N0 + sum(steps), d, 1, 20, seed=seed, sn=0.1)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    q = go.synthetic_problem(2980, d, 1, 20, seed=seed + 1, sn=0.1)
    h = Handle(lib, X[:N0], Y[:N0])
    other = Handle(lib, q['X'], q['Y'])
    assert np.all(h.fit(H) == 0) and np.all(other.fit(q['hyper']) == 0)
    om_other, ov_other = other.predict_mean_var(q['Z'])
    reused0, n = h.counter('workspace_blocks_reused'), N0
    for k, m in enumerate(steps):
        assert np.all(h.append(X[n:n + m], Y[n:n + m]) == 0)
        n += m
        assert np.all(other.fit(q['hyper']) == 0)                       # takes and returns nothing, but runs on the same device
        m1, v1 = other.predict_mean_var(q['Z'])
        assert np.array_equal(m1, om_other) and np.array_equal(v1, ov_other)
        o = go.fit(X[:n], Y[:n], H, want_invK=False)
        assert relF(h.get_factors()['chol'][0], o['chol'][0]) <= 1e-10, k
        mean, var = h.predict_mean_var(Z)
        om, ov, _ = go.mean_var_jac(Z, X[:n], H, o['alpha'], o['chol'], False)
        assert np.max(np.abs(mean - om) / mean_scale(X[:n], Z, H, o['alpha'])) <= 1e-10, k
        assert np.max(np.abs(var - ov) / H[:, d] ** 2) <= 1e-10, k
    # K, L and L^-1 of every workspace after the first new one come from the list (the scratch W is below the 64 MB threshold)
    assert h.counter('workspace_blocks_reused') - reused0 >= 3 * (len(steps) - 1), h.counter('workspace_blocks_reused') - reused0
    h.close()
    other.close()


def check_append_after_set_factors(lib, N0=300, n=10, d=4, Ny=2, seed=8):
    """load_model path followed by update_data_all (gp_class.py:58-66, :474-550): `gpmpc_set_factors` never runs the
    jitter rule, so the strip update must find a defined (zero) jitter on the device.  Compared with a full refit."""
    p = go.synthetic_problem(N0 + n, d, Ny, 12, seed=seed, sn=0.1)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    o0 = go.fit(X[:N0], Y[:N0], H, want_invK=False)
    h = Handle(lib, X[:N0], Y[:N0])
    h.set_factors(H, o0['chol'], o0['alpha'])
    assert np.all(h.append(X[N0:], Y[N0:]) == 0) and h.N == N0 + n
    f = h.get_factors()
    o = go.fit(X, Y, H, want_invK=False)
    for a in range(Ny):
        assert relF(f['chol'][a], o['chol'][a]) <= 1e-10
    mean, var = h.predict_mean_var(Z)
    om, ov, _ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], False)
    assert np.max(np.abs(mean - om) / mean_scale(X, Z, H, o['alpha'])) <= 1e-10
    assert np.max(np.abs(var - ov) / H[:, d] ** 2) <= 1e-10
    h.close()


def check_append_rollback(lib, seed=4):
    """gpmpc_append on GPMPC_ENOTPD leaves the model unchanged (include/gpmpc.h), on the strip path and on the refit
    path: appending exact duplicates with sf^2 = 1e12 >> jitter makes the extended K singular."""
    rng = np.random.default_rng(seed)
    for N0 in (40, 300):                      # 40: fewer than 64 old points -> refit path; 300: strip path
        X = rng.standard_normal((N0, 2)) * 3
        Y = rng.standard_normal((N0, 1))
        H = np.array([[0.05, 0.05, 1e6, 1e-12]])
        h = Handle(lib, X, Y)
        assert np.all(h.fit(H) == 0)
        f0 = h.get_factors()
        Z = rng.standard_normal((5, 2))
        m0, v0 = h.predict_mean_var(Z)
        try:
            h.append(X[:8], Y[:8])
            raised = False
        except NotPositiveDefinite:
            raised = True
        assert raised and h.N == N0 and np.all(h.info < 0)
        f1 = h.get_factors()                  # sized by the library's N: a stale size would overrun these buffers
        assert np.array_equal(f0['chol'], f1['chol']) and np.array_equal(f0['alpha'], f1['alpha'])
        m1, v1 = h.predict_mean_var(Z)
        assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
        # and the handle still takes a good append afterwards
        Xn = rng.standard_normal((3, 2)) * 3 + 20.0
        assert np.all(h.append(Xn, rng.standard_normal((3, 1))) == 0) and h.N == N0 + 3
        h.close()


def check_sensitivities(lib, g, nprobe=12):
    """gpmpc_predict_sens (second-order outputs for a casadi Callback) against the oracle's
    closed forms, which tests/test_oracle.py pins by finite differences of the first-order functions."""
    X, Y, H = g['X'], g['Y'], g['hyper']
    d = X.shape[1]
    sf2 = H[:, d] ** 2
    h = Handle(lib, X, Y)
    h.fit(H)
    f = h.get_factors()
    Z = g['Z'][:nprobe]
    mean, var, J, Hm, dvar = h.predict_sens(Z)
    om, ov, oJ = go.mean_var_jac(Z, X, H, f['alpha'], f['chol'])
    oH, odv = go.mean_var_sens(Z, X, H, f['alpha'], f['chol'])
    ms = mean_scale(X, Z, H, f['alpha'])
    assert np.max(np.abs(mean - om) / ms) <= 1e-10 and np.max(np.abs(var - ov) / sf2) <= 1e-10
    ell_min = H[:, :d].min(axis=1)
    # scale of a derivative: the value scale over the shortest length scale (squared for the Hessian)
    assert np.max(np.abs(J - oJ) / (ms / ell_min)[..., None]) <= 1e-10
    assert np.max(np.abs(Hm - oH) / (ms / ell_min ** 2)[..., None, None]) <= 1e-10
    cond = max(np.linalg.cond(f['chol'][a]) ** 2 for a in range(H.shape[0]))
    tol = max(1e-10, 50 * np.finfo(float).eps * cond)                     # u = K^-1 ks is cond-limited
    assert np.max(np.abs(dvar - odv) / (sf2 / ell_min)[None, :, None]) <= tol
    # chunked evaluation (more points than fit one scratch chunk is not reachable here; at least B = 1)
    m1, v1, J1, H1, d1 = h.predict_sens(Z[:1])
    assert np.allclose(H1, Hm[:1], rtol=0, atol=1e-12 * np.abs(Hm).max()) and np.allclose(d1, dvar[:1], rtol=0, atol=1e-12 * np.abs(dvar).max() + 1e-300)
    h.close()


def check_io_pack_boundary(lib, N=100, d=3, Ny=2, seed=51):
    """Host-pointer calls stage small argument sets through one pinned block (IoPack, gpmpc_api.hip: 32768 doubles); just
    below and just above that size the same call takes the two copy routes and must give the same numbers, and both
    must match the oracle."""
    B1, B2 = 1300, 1400            # 'TA' with J: 24 doubles per point (+ padding) -> the limit falls near B = 1365
    p = go.synthetic_problem(N, d, Ny, B2, seed=seed, sn=0.1)
    X, Y, H, Z, S = p['X'], p['Y'], p['hyper'], p['Z'], p['Sigma']
    h = Handle(lib, X, Y)
    assert np.all(h.fit(H) == 0)
    f = h.get_factors()
    sf2 = H[:, d] ** 2
    ma, ca, Ja = h.predict_jac('TA', Z[:B1], S[:B1])
    mb, cb, Jb = h.predict_jac('TA', Z[:B2], S[:B2])
    assert np.array_equal(ma, mb[:B1]) and np.array_equal(Ja, Jb[:B1])
    assert np.allclose(ca, cb[:B1], rtol=0, atol=1e-13 * sf2.max())
    om, ov, oJ = go.mean_var_jac(Z, X, H, f['alpha'], f['chol'])
    oc = go.ta_cov(ov, oJ, S)
    ms = mean_scale(X, Z, H, f['alpha'])
    assert np.max(np.abs(mb - om) / ms) <= 1e-10 and np.max(np.abs(cb - oc)) <= 1e-10 * sf2.max() * max(1.0, np.abs(oc).max())
    m1, v1 = h.predict_mean_var(Z[:1])      # and the smallest call there is
    assert np.array_equal(m1, mb[:1]) and abs(v1[0, 0] - cb[0, 0, 0] + (oJ[0] @ S[0] @ oJ[0].T)[0, 0]) <= 1e-12 * sf2.max()
    h.close()


def check_sensitivities_batches(lib, N=330, d=4, Ny=2, seed=41):
    """gpmpc_predict_sens over the batch sizes that take different routes for V = L^-1 Ks and U = L^-T V (32- and
    64-column streaming tiles with the transposed store; beyond 64 columns the row-major product), on a model with
    several 64-row tiles so that the triangular K ranges of both passes matter.  Against the oracle's closed forms."""
    p = go.synthetic_problem(N, d, Ny, 100, seed=seed, sn=0.1)
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    sf2, ell_min = H[:, d] ** 2, H[:, :d].min(axis=1)
    h = Handle(lib, X, Y)
    assert np.all(h.fit(H) == 0)
    f = h.get_factors()
    oH, odv = go.mean_var_sens(Z, X, H, f['alpha'], f['chol'])
    om, ov, oJ = go.mean_var_jac(Z, X, H, f['alpha'], f['chol'])
    ms = mean_scale(X, Z, H, f['alpha'])
    for B in (3, 33, 64, 70, 100):
        mean, var, J, Hm, dvar = h.predict_sens(Z[:B])
        assert np.max(np.abs(mean - om[:B]) / ms[:B]) <= 1e-10 and np.max(np.abs(var - ov[:B]) / sf2) <= 1e-10, B
        assert np.max(np.abs(J - oJ[:B]) / (ms[:B] / ell_min)[..., None]) <= 1e-10, B
        assert np.max(np.abs(Hm - oH[:B]) / (ms[:B] / ell_min ** 2)[..., None, None]) <= 1e-10, B
        assert np.max(np.abs(dvar - odv[:B]) / (sf2 / ell_min)[None, :, None]) <= 1e-10, (B, np.max(np.abs(dvar - odv[:B])))
        m2, v2 = h.predict_mean_var(Z[:B])                    # the variance of the value path (fused sums, no V kept)
        assert np.array_equal(m2, mean) and np.allclose(v2, var, rtol=0, atol=1e-13 * sf2.max()), B
    h.close()


def check_gp_class(lib, g, tmp_path, em_rollout=True):
    """The Python `GP` surface (reference gp_class.py) against `OracleGP` on a saved reference model."""
    from gp_mpc_amd.gp import GP
    hyper = dict(hyper=g['hyper'], chol=g['chol'], alpha=g['alpha'], invK=g['invK'])
    kw = dict(normalize=g['normalize'], lib=lib)
    if g['normalize']:
        kw.update(meta=g['meta'], xlb=g['xlb'], xub=g['xub'], ulb=g['ulb'], uub=g['uub'])
    gp = GP(g['X'], g['Y'], hyper=hyper, gp_method='TA', **kw)
    og = go.OracleGP(g['X'], g['Y'], g['hyper'], g['chol'], g['alpha'], g['invK'], normalize=g['normalize'],
                     meta=g.get('meta'), gp_method='TA')
    N, Ny, Nu = gp.get_size()
    assert (N, Ny, Nu) == (og.N, og.Ny, og.Nu)
    Nx = Ny + Nu
    if g['normalize']:
        x = g['meta']['meanX'] + 0.3 * g['meta']['stdX']
        u = g['meta']['meanU'] - 0.2 * g['meta']['stdU']
    else:
        x, u = g['X'][3, :Ny] * 1.01, g['X'][3, Ny:] * 0.99
    A = np.random.default_rng(0).standard_normal((Nx, Nx)) * 0.03
    S = A @ A.T + 1e-6 * np.eye(Nx)
    sf2 = g['hyper'][:, Nx] ** 2
    for m in ('ME', 'TA', 'EM', 'old_ME', 'old_TA'):
        gp.set_method(m)
        og.set_method(m)
        mean, cov = gp.predict(x, u, S)
        om, oc = og.predict(x, u, S)
        assert mean.shape == (Ny, 1) and cov.shape == (Ny, Ny)
        # bars on the reference's tank model (cond(K) 6e7): 1e-9 for ME / TA; the moment methods sum N^2 terms of
        # size |alpha|^2 ~ 1e6 against K^-1 and are held to 2e-5 HERE -- check_gp_class_strict repeats all five
        # methods on a well-conditioned model at 1e-10, where no such excuse applies
        rel = 1e-9 if m in ('ME', 'TA') else 2e-5
        assert np.max(np.abs(mean - om)) <= rel * max(1.0, np.abs(om).max()), (m, mean.ravel(), om.ravel())
        assert np.max(np.abs(cov - oc)) <= rel * max(sf2.max(), np.abs(oc).max()), (m, np.abs(cov - oc).max())
    try:
        gp.set_method('XX')
        assert False
    except NameError:
        pass
    gp.set_method('TA')
    og.set_method('TA')
    Ad, Bd = gp.discrete_linearize(x, u, S)
    oA, oB = og.discrete_linearize(x, u, S)
    assert np.allclose(Ad, oA, rtol=1e-8, atol=1e-10 * np.abs(oA).max()) and np.allclose(Bd, oB, rtol=1e-8, atol=1e-10 * np.abs(oB).max())
    assert np.array_equal(gp.noise_variance(), g['hyper'][:, Nx + 1] ** 2)
    hp = gp.get_hyper_parameters()
    assert np.array_equal(hp['length_scale'], g['hyper'][:, :Nx]) and np.array_equal(hp['mean'], g['hyper'][:, Nx + 1:])
    # rollout = numeric loop of predict_compare
    U = np.tile(u, (4, 1))
    mr, vr = gp.rollout(x, U, methods=['TA', 'ME'])
    omr, ovr = og.rollout(x, U, methods=('TA', 'ME'))
    assert np.allclose(mr, omr, rtol=1e-7, atol=1e-9) and np.allclose(vr, np.clip(ovr, 0, None), rtol=1e-5, atol=1e-9 * sf2.max())
    # the exact-moment roll-out too.  EM covariances on the reference's models are cancellation-limited (~1e-5
    # absolute here, alpha ~ 1e3) and the feedback amplifies that noise, so the device roll-out (one call,
    # gpmpc_rollout) is checked against the SAME device predictor driven step by step from the host, as the
    # reference's loop does (gp_class.py:777-804); single EM steps are compared with the oracle elsewhere.
    if em_rollout:
        me, ve = gp.rollout(x, U, methods=['EM'])
        gp.set_method('EM')
        covar = np.eye(Nx) * 1e-6
        covar[:Ny, :Ny] = np.diag(g['hyper'][:, Nx + 1] ** 2)
        mt = np.asarray(x, dtype=np.float64)
        for t in range(1, 5):
            mt, cx = gp.predict(mt, U[t - 1], covar)
            mt = mt.reshape(Ny)
            vt = np.diag(cx) * (g['meta']['stdY'] ** 2 if g['normalize'] else 1.0)
            # (the device maps mean_s -> next x_s in one affine step, the host un-standardises and re-standardises:
            #  one ulp apart, which EM's cancellation turns into ~1e-2 relative in the variance after 4 steps)
            assert np.allclose(me[0, t], mt, rtol=1e-7, atol=1e-7 * np.abs(mt).max())
            assert np.allclose(ve[0, t], np.clip(vt, 0, None), rtol=3e-2, atol=1e-6 * np.abs(vt).max())
            covar[:Ny, :Ny] = cx
        gp.set_method('TA')
    # validate on the training inputs themselves
    Xraw = g['X'] * g['meta']['stdZ'] + g['meta']['meanZ'] if g['normalize'] else g['X']
    Yraw = g['Y'] * g['meta']['stdY'] + g['meta']['meanY'] if g['normalize'] else g['Y']
    smse, mnlp = gp.validate(Xraw[:20], Yraw[:20], verbose=False)
    osmse, omnlp = og.validate(Xraw[:20], Yraw[:20])
    assert np.allclose(smse, osmse, rtol=1e-5, atol=1e-12) and np.allclose(mnlp, omnlp, rtol=1e-6, atol=1e-8)
    # covar / covSEard
    cv = gp.covar(g['Z'][:4])
    assert cv.shape == (Nx, 4, 4)
    assert np.max(np.abs(cv[:Ny] - g['ref_covar'][:, :4, :4]) / sf2[:, None, None]) <= 1e-10
    ks = gp.covSEard(g['X'], g['Z'], g['hyper'][0, :Nx], sf2[0])
    assert np.max(np.abs(ks - g['ref_ks'][0])) <= 1e-14 * sf2[0]
    try:
        gp.covSEard(g['X'], g['Z'][:, :Nx - 1], g['hyper'][0, :Nx], 1.0)
        assert False
    except ValueError:
        pass
    # exact derivatives of predict (what a casadi Callback hands to IPOPT) vs central differences of predict
    for method in ('TA', 'ME'):
        gp.set_method(method)
        m0, c0, D = gp.predict_derivatives(x, u, S)
        mm, cc = gp.predict(x, u, S)
        assert np.array_equal(m0, mm) and np.allclose(c0, cc, rtol=0, atol=1e-13 * sf2.max())
        zraw = np.concatenate([x, u])
        dm = np.concatenate([D['dmean_dx'], D['dmean_du']], axis=1)
        dc = np.concatenate([D['dcov_dx'], D['dcov_du']], axis=2)
        for k in range(Nx):
            e = np.zeros(Nx)
            e[k] = 1e-4 * max(1.0, abs(zraw[k]))
            mp, cp = gp.predict((zraw + e)[:Ny], (zraw + e)[Ny:], S)
            mn, cn = gp.predict((zraw - e)[:Ny], (zraw - e)[Ny:], S)
            assert np.allclose((mp - mn)[:, 0] / (2 * e[k]), dm[:, k], rtol=1e-5, atol=1e-6 * np.abs(dm).max())
            # (the variance carries ~1e-13 sf^2 of cancellation noise, which the difference quotient amplifies)
            assert np.allclose((cp - cn) / (2 * e[k]), dc[:, :, k], rtol=1e-4, atol=1e-5 * np.abs(dc).max() + 1e-12 * sf2.max() / e[k])
        E = np.zeros((Nx, Nx))
        E[0, 1] = 1e-3
        _, cp = gp.predict(x, u, S + E)
        assert np.allclose((cp - cc) / 1e-3, D['dcov_dcov'][:, :, 0, 1], rtol=1e-9, atol=1e-12 * sf2.max() / 1e-3)
    gp.set_method('old_TA')
    try:
        gp.predict_derivatives(x, u, S)
        assert False
    except NotImplementedError:
        pass
    gp.set_method('TA')
    # save / load round trip in the reference's JSON layout
    path = str(tmp_path / 'model')
    gp.save_model(path)
    import json
    d = json.load(open(path + '.json'))
    assert set(d['hyper'].keys()) == {'hyper', 'invK', 'alpha', 'chol', 'length_scale', 'signal_var', 'noise_var', 'mean'}
    gp2 = GP.load_model(path, lib=lib)
    gp2.set_method('TA')
    m2, c2 = gp2.predict(x, u, S)
    m1, c1 = gp.predict(x, u, S)
    assert np.allclose(m1, m2, rtol=1e-12, atol=0) and np.allclose(c1, c2, rtol=0, atol=1e-12 * sf2.max())
    # the same with the matrices in a binary sidecar (what large models use by default)
    gp.save_model(path + '_bin', sidecar=True)
    d = json.load(open(path + '_bin.json'))
    assert d['hyper']['chol'] == {'__sidecar__': 'hyper_chol'} and d['sidecar_file'] == 'model_bin.npz'
    gp3 = GP.load_model(path + '_bin', lib=lib)
    gp3.set_method('TA')
    m3b, c3b = gp3.predict(x, u, S)
    assert np.array_equal(m1, m3b) and np.array_equal(c1, c3b)          # binary round trip is exact
    gp3.close()
    # data replacement keeps the hyper-parameters and refits (gp_class.py:553-626)
    gp2.replace_data_all(Xraw[:40], Yraw[:40])
    assert gp2.get_size()[0] == 40
    Xs = (Xraw[:40] - g['meta']['meanZ']) / g['meta']['stdZ'] if g['normalize'] else Xraw[:40]
    Ys = (Yraw[:40] - g['meta']['meanY']) / g['meta']['stdY'] if g['normalize'] else Yraw[:40]
    o2 = go.OracleGP(Xs, Ys, g['hyper'], normalize=g['normalize'], meta=g.get('meta'), gp_method='ME')
    gp2.set_method('ME')
    m3, c3 = gp2.predict(x, u, S)
    om3, oc3 = o2.predict(x, u, S)
    assert np.allclose(m3, om3, rtol=1e-8, atol=1e-9) and np.allclose(c3, oc3, rtol=0, atol=1e-10 * sf2.max())
    gp2.update_data_all(Xraw[40:50], Yraw[40:50])
    assert gp2.get_size()[0] == 50
    Xs5 = (Xraw[:50] - g['meta']['meanZ']) / g['meta']['stdZ'] if g['normalize'] else Xraw[:50]
    Ys5 = (Yraw[:50] - g['meta']['meanY']) / g['meta']['stdY'] if g['normalize'] else Yraw[:50]
    o5 = go.OracleGP(Xs5, Ys5, g['hyper'], normalize=g['normalize'], meta=g.get('meta'), gp_method='ME')
    m5, c5 = gp2.predict(x, u, S)
    om5, oc5 = o5.predict(x, u, S)
    assert np.allclose(m5, om5, rtol=1e-8, atol=1e-9) and np.allclose(c5, oc5, rtol=0, atol=1e-10 * sf2.max())
    try:
        gp.update_data(Xraw[:2], Yraw[:2])             # documented as not working in the reference (gp_class.py:384-471)
        assert False
    except NotImplementedError:
        pass
    # predict_compare (gp_class.py:746-861) without the figures: the arrays they were drawn from, next to a simulator's
    # trajectory when a model object is handed in (van_der_pol.py:83-85, tank_example.py call it that way)
    class _Sim:                                         # stands in for model_class.Model (casadi + SUNDIALS, out of scope)
        def sampling_time(self): return 0.5
        def sim(self, x0, uu): return np.cumsum(np.ones((len(uu), Ny)), axis=0) + np.asarray(x0).reshape(1, Ny)
    pcmp = gp.predict_compare(x, U, _Sim(), methods=['TA', 'ME'], title='ignored', num_cols=3)
    assert pcmp['methods'] == ['TA', 'ME'] and np.allclose(pcmp['t'], 0.5 * np.arange(5))
    assert np.array_equal(pcmp['mean'], mr) and np.array_equal(pcmp['var'], vr) and np.all(pcmp['var'] >= 0)
    assert pcmp['y_sim'].shape == (5, Ny) and np.array_equal(pcmp['y_sim'][0], np.asarray(x, dtype=np.float64).reshape(Ny))
    assert gp.predict_compare(x, U, methods=['ME'])['y_sim'] is None
    gp.close()
    gp2.close()


def check_training(lib, t):
    """a8: train from the reference's initial point with the reference's bounds; the optimum found
    with device NLL + analytic gradient must be at least as good as train_gp_numpy's and, from the
    same start, land on the same local optimum."""
    from gp_mpc_amd.gp import GP
    X, Y = t['X'], t['Y']
    gp = GP(X, Y, normalize=False, multistart=1, gp_method='ME', lib=lib)
    H = gp.train_info['hyper']
    for a in range(Y.shape[1]):
        ours = go.nll(H[a], X, Y[:, a])
        assert ours <= t['nll'][a] + 1e-6 * abs(t['nll'][a]), (a, ours, t['nll'][a])
        if abs(ours - t['nll'][a]) <= 1e-4 * abs(t['nll'][a]):     # same basin -> same hyper-parameters
            assert np.allclose(H[a], t['hyper'][a], rtol=2e-2), (H[a], t['hyper'][a])
    assert np.allclose(H[0], t['hyper'][0], rtol=2e-2)              # output 0 has a sharp optimum
    # the fitted model predicts like a model built from the trained hyper-parameters
    f = gp.handle.get_factors()
    o = go.fit(X, Y, H, want_invK=False)
    for a in range(Y.shape[1]):
        assert relF(f['chol'][a], o['chol'][a]) <= 1e-9
    gp.close()


def check_edge_cases(lib):
    """Ragged / minimal / maximal shapes and the error behaviour of the C ABI."""
    from gp_mpc_amd._lib import GpmpcError, EINVAL, ENOTFIT
    rng = np.random.default_rng(11)
    for (N, d, Ny) in [(1, 1, 1), (2, 3, 2), (63, 2, 1), (64, 16, 1), (65, 5, 3), (130, 1, 2)]:
        X = rng.standard_normal((N, d))
        Y = rng.standard_normal((N, Ny))
        H = np.hstack([rng.uniform(0.7, 2.0, (Ny, d)), rng.uniform(0.8, 1.5, (Ny, 1)), np.full((Ny, 1), 0.1)])
        h = Handle(lib, X, Y)
        assert np.all(h.fit(H, want_invK=True) == 0)
        o = go.fit(X, Y, H)
        f = h.get_factors(invK=True)
        for a in range(Ny):
            assert relF(f['chol'][a], o['chol'][a]) <= 1e-12 and relF(f['invK'][a], o['invK'][a]) <= 1e-10
            assert relF(f['alpha'][a], o['alpha'][a]) <= 1e-10
        for B in (1, 7, 8, 9, 64, 65):
            Z = rng.standard_normal((B, d))
            mean, var = h.predict_mean_var(Z)
            om, ov, oJ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'])
            assert np.max(np.abs(mean - om)) <= 1e-11 * max(1.0, np.abs(om).max())
            assert np.max(np.abs(var - ov) / (H[:, d] ** 2)) <= 1e-12
            _, J = h.mean_jac(Z)
            assert np.max(np.abs(J - oJ)) <= 1e-10 * max(1.0, np.abs(oJ).max())
        if d <= 8:
            S = go.synthetic_problem(4, d, 1, 3, seed=2)['Sigma'] * 20
            Z3 = rng.standard_normal((3, d))
            m, c = h.predict('EM', Z3, S)
            for b in range(3):
                om, oc = go.exact_moment(f['invK'], X, Y, H, Z3[b], S[b])
                assert np.max(np.abs(m[b] - om)) <= 1e-9 * max(1.0, np.abs(om).max())
                assert np.max(np.abs(c[b] - oc)) <= 1e-8 * max(1.0, np.abs(oc).max(), (H[:, d] ** 2).max())
        h.close()
    # error behaviour: status codes, no silent fallbacks
    X, Y = rng.standard_normal((10, 3)), rng.standard_normal((10, 1))
    h = Handle(lib, X, Y)
    for fn, code in ((lambda: h.predict_mean_var(X[:2]), ENOTFIT),
                     (lambda: h.fit(np.array([[1.0, 1.0, np.nan, 1.0, 0.1]])), EINVAL),
                     (lambda: h.fit(np.array([[1.0, 0.0, 1.0, 1.0, 0.1]])), EINVAL)):
        try:
            fn()
            assert False
        except GpmpcError as e:
            assert e.code == code, (e.code, code)
    h.fit(np.array([[1.0, 1.0, 1.0, 1.0, 0.1]]))
    for fn in (lambda: h.predict(7, X[:2], None), lambda: h.predict('TA', X[:2], None),
               lambda: Handle(lib, rng.standard_normal((5, 17)), rng.standard_normal((5, 1)))):
        try:
            fn()
            assert False
        except GpmpcError as e:
            assert e.code == EINVAL
    h.close()


# ---------------------------------------------------------------------------------------------
# BASELINE configs C3 / C4 / C5 (bodies shared by the GPU tier at full size and the emulator tier at toy size)
# ---------------------------------------------------------------------------------------------
def rollout_inputs(p, Ny, d, T):
    x0 = p['Z'][0, :Ny]
    U = p['Z'][:T, Ny:]
    S0 = np.eye(d) * 1e-6                                           # gp_class.py:764
    S0[:Ny, :Ny] = np.diag(p['hyper'][:, d + 1] ** 2)               # gp_class.py:780
    return x0, U, S0


def check_rollout_vs_host_loop(h, p, Ny, d, T, em_tol=(1e-9, 1e-6)):
    """`gpmpc_rollout` == the reference's loop (gp_class.py:777-804) driven from the host with the same device
    predictor: bitwise for ME and TA (same kernels, same inputs), tight for EM; covariance symmetric positive
    semi-definite at every step."""
    x0, U, S0 = rollout_inputs(p, Ny, d, T)
    z0 = np.concatenate([x0, U[0]])
    m_me = None
    for method in ('ME', 'TA', 'EM'):
        m, c = h.rollout(method, z0, U, S0)
        assert m.shape == (T, Ny) and c.shape == (T, Ny, Ny) and np.all(np.isfinite(m)) and np.all(np.isfinite(c))
        mean_t, S = x0.copy(), S0.copy()
        for t in range(T):
            z = np.concatenate([mean_t, U[t]])
            mm, cc = h.predict(method, z.reshape(1, d), S.reshape(1, d, d))
            if method in ('ME', 'TA'):
                assert np.array_equal(mm[0], m[t]) and np.array_equal(cc[0], c[t]), (method, t)
            else:
                assert np.allclose(mm[0], m[t], rtol=em_tol[0], atol=em_tol[0]), (t, np.abs(mm[0] - m[t]).max())
                assert np.allclose(cc[0], c[t], rtol=em_tol[1], atol=1e-8), (t, np.abs(cc[0] - c[t]).max())
            mean_t = mm[0]
            S[:Ny, :Ny] = cc[0]
            if method == 'TA':     # J Sigma J^T entry by entry (like the reference's matrix product): symmetric to rounding
                assert np.max(np.abs(c[t] - c[t].T)) <= 1e-13 * np.abs(c[t]).max()
            else:
                assert np.array_equal(c[t], c[t].T)
            assert np.linalg.eigvalsh(0.5 * (c[t] + c[t].T)).min() >= -1e-7, (method, t, np.linalg.eigvalsh(c[t]).min())
        if method == 'ME':
            m_me = m
        elif method == 'TA':     # first step: TA's mean IS the GP mean (gp_class.py:216-219)
            assert np.array_equal(m[0], m_me[0])
        else:                    # EM's first-step mean differs at second order in the (small) initial covariance
            assert np.max(np.abs(m[0] - m_me[0])) <= 10 * np.abs(S0).max() * max(1.0, np.abs(m_me[0]).max())


def check_rollout_replay(lib, N=150, Ny=2, d=3, T=6, seed=31):
    """Repeated identical roll-outs go through the captured hipGraph from the third call on (gpmpc_api.hip, rollout_impl):
    the replays must give the bits of the first, plain run; new inputs must be picked up by a replay (they live in the
    staging block, not in the graph); so must a refit with other hyper-parameters (same buffers, new contents) -- checked
    against the host-driven loop again; and a different horizon must not hit the cached loop."""
    p = go.synthetic_problem(N, d, Ny, 8, seed=seed, sn=0.1)
    h = Handle(lib, p['X'], p['Y'])
    assert np.all(h.fit(p['hyper'], want_invK=True) == 0)
    x0, U, S0 = rollout_inputs(p, Ny, d, T)
    z0 = np.concatenate([x0, U[0]])
    for method in ('ME', 'TA', 'EM'):
        first = h.rollout(method, z0, U, S0)
        for _ in range(4):
            again = h.rollout(method, z0, U, S0)
            assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1]), method
        z1 = z0.copy()
        z1[0] += 0.05
        moved = h.rollout(method, z1, 0.5 * U, S0)
        assert not np.array_equal(moved[0], first[0])
        back = h.rollout(method, z0, U, S0)
        assert np.array_equal(first[0], back[0]) and np.array_equal(first[1], back[1]), method
        short = h.rollout(method, z0, U[:T - 2], S0)
        assert np.array_equal(short[0], first[0][:T - 2]) and np.array_equal(short[1], first[1][:T - 2]), method
    H2 = p['hyper'].copy()
    H2[:, :d] *= 1.3
    assert np.all(h.fit(H2, want_invK=True) == 0)
    for _ in range(3):
        check_rollout_vs_host_loop(h, p, Ny, d, T)
    h.close()


def check_far_points(lib, N=150, d=3, Ny=2, seed=9):
    """Test points 1e5 length scales away from every training point (the table exp of the cross-covariance
    kernel took its integer from the low word of a magic-number sum, which wraps beyond |x - z| / ell ~ 1e4 -- ldexp then
    returned inf or garbage instead of 0): k(X, z) = 0, so mean = 0 and var = sf^2 exactly, in the large-batch and the
    small-batch route; and sf = 0 handed in through gpmpc_set_factors gives zeros, not NaN.  The same for the exact moments."""
    p = go.synthetic_problem(N, d, Ny, 80, seed=seed, sn=0.1)
    X, Y, H = p['X'], p['Y'], p['hyper']
    h = Handle(lib, X, Y)
    h.fit(H)
    Z = p['Z'].copy()
    Z[::2] = Z[::2] + 1e5 * H[0, :d].max() * np.array([1.0, -1.0, 1.0][:d])
    Z[1] = 3e8
    for Zq in (Z, Z[:3]):
        m, v = h.predict_mean_var(Zq)
        far = np.arange(len(Zq)) % 2 == 0
        far[1] = True
        assert np.all(np.isfinite(m)) and np.all(np.isfinite(v))
        assert np.all(m[far[:len(Zq)]] == 0.0) and np.all(v[far[:len(Zq)]] == H[:, d] ** 2)
    m, c, J = h.predict_jac('TA', Z[:4], p['Sigma'][:4])
    assert np.all(np.isfinite(c)) and np.all(J[0] == 0.0)
    # exact moments (r06: the pair sums' table exp took the power of two from the low word of its magic-number sum alone and
    # the covariance came out NaN from ~1 200 length scales on; now from both words, and arguments <= -1e9 are clamped):
    # every Q is an exact zero, so mean = 0 and cov = diag(sf^2) exactly, at 1e3, 1e5 length scales and at 3e8
    for scale in (1e3, 1e5):
        Ze = p['Z'][:4].copy()
        Ze[0] = Ze[0] + scale * H[0, :d].max() * np.array([1.0, -1.0, 1.0][:d])
        Ze[2] = 3e8
        me, ce = h.predict('EM', Ze, p['Sigma'][:4] * 1e-3)
        assert np.all(np.isfinite(me)) and np.all(np.isfinite(ce)), scale
        for b in (0, 2):
            assert np.all(me[b] == 0.0) and np.array_equal(ce[b], np.diag(H[:, d] ** 2)), (scale, b)
    f = h.get_factors()
    H0 = H.copy()
    H0[0, d] = 0.0                                                  # sf = 0 for the first output
    h.set_factors(H0, f['chol'], f['alpha'])
    m, v = h.predict_mean_var(p['Z'])
    assert np.all(np.isfinite(m)) and np.all(np.isfinite(v)) and np.all(m[:, 0] == 0.0) and np.all(v[:, 0] == 0.0)
    h.close()


def check_rollout_multi(lib, N=150, Ny=2, d=3, T=5, seed=33, methods=('ME', 'TA', 'EM', 'old_ME')):
    """gpmpc_rollout_multi (SURVEY a17 'batch across trajectories / methods', gp_class.py:777-804) against gpmpc_rollout:
    * a call with ONE trajectory is bitwise the single-trajectory call, for every method;
    * in a mixed call the moment-method trajectories are bitwise the single calls, the 'ME' / 'TA' trajectories agree with
      them to rounding (batched variance kernel: another summation order), scaled bars 1e-10;
    * a trajectory's bits do not depend on its position or on what else is in the call (two compositions with >= 2 'ME' / 'TA').
    (gpmpc_rollout itself is pinned against the oracle in check_rollout_vs_oracle.)"""
    p = go.synthetic_problem(N, d, Ny, T + 2, seed=seed, sn=0.1)
    X, Y, H = p['X'], p['Y'], p['hyper']
    h = Handle(lib, X, Y)
    h.fit(H, want_invK=True)
    Nu = d - Ny
    rng = np.random.default_rng(seed)
    za, zb, zc = p['Z'][0], p['Z'][1], p['Z'][2]
    Ua, Ub = 0.3 * rng.standard_normal((T, Nu)), 0.3 * rng.standard_normal((T, Nu))
    S0 = np.eye(d) * 1e-6
    S0[:Ny, :Ny] = np.diag(H[:, d + 1] ** 2)
    S1 = S0 * 2.0
    sf2 = (H[:, d] ** 2).max()
    single = {m: h.rollout(m, za, Ua, S0) for m in methods}
    for m in methods:                                               # one trajectory: the same launches
        mm, cc = h.rollout_multi([m], za, Ua, S0)
        assert np.array_equal(mm[0], single[m][0]) and np.array_equal(cc[0], single[m][1]), m
    mm, cc = h.rollout_multi(list(methods), za, Ua, S0)              # every method from the same start, as GP.rollout runs them
    for i, m in enumerate(methods):
        if m in ('ME', 'TA'):
            assert np.max(np.abs(mm[i] - single[m][0])) <= 1e-10 * max(1.0, np.abs(single[m][0]).max()), m
            assert np.max(np.abs(cc[i] - single[m][1])) <= 1e-10 * sf2, m
        else:
            assert np.array_equal(mm[i], single[m][0]) and np.array_equal(cc[i], single[m][1]), m
    # composition / position invariance with different starts, controls and input covariances
    m1, c1 = h.rollout_multi(['ME', 'TA'], np.stack([za, zb]), np.stack([Ua, Ub]), np.stack([S0, S1]))
    m2, c2 = h.rollout_multi(['TA', 'ME', 'EM', 'ME'], np.stack([zb, zc, za, za]), np.stack([Ub, Ua, Ua, Ua]), np.stack([S1, S0, S0, S0]))
    assert np.array_equal(m1[0], m2[3]) and np.array_equal(c1[0], c2[3])          # 'ME' from za
    assert np.array_equal(m1[1], m2[0]) and np.array_equal(c1[1], c2[0])          # 'TA' from zb
    if 'EM' in methods:
        assert np.array_equal(m2[2], single['EM'][0]) and np.array_equal(c2[2], single['EM'][1])
    sb = h.rollout('TA', zb, Ub, S1)
    assert np.max(np.abs(m1[1] - sb[0])) <= 1e-10 * max(1.0, np.abs(sb[0]).max()) and np.max(np.abs(c1[1] - sb[1])) <= 1e-10 * sf2
    # a model fitted WITHOUT K^-1: the call forms it on the moment methods' queue, next to the 'ME' / 'TA' group's steps
    if 'EM' in methods:
        h.fit(H)
        m3, c3 = h.rollout_multi(['ME', 'EM', 'TA'], np.stack([za, za, zb]), np.stack([Ua, Ua, Ub]), np.stack([S0, S0, S1]))
        assert np.array_equal(m3[1], single['EM'][0]) and np.array_equal(c3[1], single['EM'][1])
        assert np.array_equal(m3[0], m1[0]) and np.array_equal(c3[0], c1[0]) and np.array_equal(m3[2], m1[1]) and np.array_equal(c3[2], c1[1])
    for bad in (lambda: h.rollout_multi([], za, Ua, S0), lambda: h.rollout_multi([9], za, Ua, S0)):
        try:
            bad()
            assert False
        except Exception:
            pass
    h.close()


def check_rollout_vs_oracle(lib, N, Ny, d, T, seed=77, uscale=0.3):
    """T-step propagation (EM / TA / ME) on the device against the oracle (restatement of gp_class.py:777-804 over
    gp_exact_moment / build_gp / build_TA_cov) on a well-conditioned model (sn = 0.1), in two ways:
      * step by step ALONG THE DEVICE'S TRAJECTORY: the oracle's single-step prediction from the device's own
        (mean_{t-1}, cov_{t-1}) must reproduce the device's (mean_t, cov_t) at the single-step parity bars -- a feedback
        loop of T steps amplifies rounding differences, a per-step comparison does not;
      * end to end against the oracle's own roll-out (`GP.rollout` -> `gpmpc_rollout`) at a bar that allows for that
        amplification."""
    from gp_mpc_amd.gp import GP
    p = go.synthetic_problem(N, d, Ny, T, seed=seed, sn=0.1)
    o = go.fit(p['X'], p['Y'], p['hyper'])
    gp = GP(p['X'], p['Y'], hyper=dict(hyper=p['hyper'], chol=o['chol'], alpha=o['alpha'], invK=o['invK']),
            normalize=False, gp_method='TA', lib=lib)
    og = go.OracleGP(p['X'], p['Y'], p['hyper'], o['chol'], o['alpha'], o['invK'], gp_method='TA')
    x0, U, S0 = rollout_inputs(p, Ny, d, T)
    U = U * uscale
    sf2 = (p['hyper'][:, d] ** 2).max()
    for method in ('EM', 'TA', 'ME'):
        og.set_method(method)
        m, c = gp.handle.rollout(method, np.concatenate([x0, U[0]]), U, S0)
        mean_prev, S = x0.copy(), S0.copy()
        for t in range(T):
            om, oc = og.predict(mean_prev, U[t], S)
            assert np.max(np.abs(m[t] - om[:, 0])) <= 1e-10 * max(1.0, np.abs(om).max()), (method, t, np.abs(m[t] - om[:, 0]).max())
            if method == 'EM':    # pair sums of N^2 terms cancel to the result: the bar of check_moment_methods,
                z = np.concatenate([mean_prev, U[t]])       # 1e-9 of the cancellation scale sum |A o Q|
                bar = 1e-9 * (_em_scale(o['invK'], p['X'], p['Y'], p['hyper'], z, S).max() + sf2)
            else:
                bar = 1e-9 * max(sf2, np.abs(oc).max())
            assert np.max(np.abs(c[t] - oc)) <= bar, (method, t, np.abs(c[t] - oc).max(), bar)
            mean_prev = m[t]
            S[:Ny, :Ny] = c[t]
    m, v = gp.rollout(x0, U, methods=['EM', 'TA', 'ME'])
    om, ov = og.rollout(x0, U, methods=('EM', 'TA', 'ME'))
    assert np.all(np.isfinite(om)) and np.all(np.isfinite(ov))
    ov = np.clip(ov, 0, None)
    assert np.max(np.abs(m - om)) <= 1e-6 * max(1.0, np.abs(om).max()), np.max(np.abs(m - om))
    assert np.max(np.abs(v - ov)) <= 1e-6 * max(1.0, np.abs(ov).max()), np.max(np.abs(v - ov))
    gp.close()


def exact_moment_longdouble(X, Y, H, mu, Sigma):
    """gp_exact_moment (gp_functions.py:344-418) evaluated in longdouble (TEST ONLY: the yardstick for how far an fp64
    evaluation -- the oracle's, the device's -- is from the formula's exact value): K^-1 column by column is too much at
    N = 8192, so the two places it enters are formed without it -- beta = K^-1 y by longdouble_alpha, and
    sum_ij K^-1_ij Q_ij = trace(K^-1 Q) = sum_j (K^-1 Q[:, j])_j with Q[:, j] refined the same way in blocks of columns
    (Q_aa is N x N: the refinement runs on all its columns at once, matrix right-hand side)."""
    from scipy.linalg import cho_factor, cho_solve
    ld = np.longdouble
    Ny, (N, Nx) = len(H), X.shape
    Hl = np.log(H.astype(ld))
    mu = np.asarray(mu, dtype=ld).reshape(1, Nx)
    S = np.asarray(Sigma, dtype=ld)
    v = X.astype(ld) - mu
    eye = np.eye(Nx, dtype=ld)

    def solve_ld(A, B):                                  # small (Nx x Nx) systems: Gauss-Jordan in longdouble
        A, B = A.copy(), B.copy()
        n = len(A)
        for i in range(n):
            piv = i + int(np.argmax(np.abs(A[i:, i])))
            A[[i, piv]], B[[i, piv]] = A[[piv, i]], B[[piv, i]]
            B[i] = B[i] / A[i, i]
            A[i] = A[i] / A[i, i]
            for r in range(n):
                if r != i:
                    B[r] = B[r] - A[r, i] * B[i]
                    A[r] = A[r] - A[r, i] * A[i]
        return B

    def det_ld(A):
        A = A.copy()
        n, det = len(A), ld(1)
        for i in range(n):
            piv = i + int(np.argmax(np.abs(A[i:, i])))
            if piv != i:
                A[[i, piv]] = A[[piv, i]]
                det = -det
            det = det * A[i, i]
            A[i + 1:] = A[i + 1:] - np.outer(A[i + 1:, i] / A[i, i], A[i])
        return abs(det)

    beta, Kl, cfs = [], [], []
    for a in range(Ny):
        al, K = longdouble_alpha(X, Y[:, a], H[a])
        beta.append(al)
        Kl.append(K)
        cfs.append(cho_factor(K.astype(np.float64), lower=True))
    mean = np.zeros(Ny, dtype=ld)
    log_k = np.zeros((N, Ny), dtype=ld)
    for a in range(Ny):
        iLam = np.diag(np.exp(-2 * Hl[a, :Nx]))
        R = S + np.diag(np.exp(2 * Hl[a, :Nx]))
        iR = iLam @ (eye - solve_ld(eye + S @ iLam, S @ iLam))
        T = v @ iR
        c = np.exp(2 * Hl[a, Nx]) / np.sqrt(det_ld(R)) * np.exp(np.sum(Hl[a, :Nx]))
        mean[a] = np.sum(c * np.exp(-np.sum(T * v, axis=1) * ld(0.5)) * beta[a])
        v1 = v / np.exp(Hl[a, :Nx])[None, :]
        log_k[:, a] = 2 * Hl[a, Nx] - np.sum(v1 * v1, axis=1) * ld(0.5)
    cov = np.zeros((Ny, Ny), dtype=ld)
    for a in range(Ny):
        ii = v / np.exp(2 * Hl[a, :Nx])[None, :]
        for b in range(a + 1):
            R = S @ np.diag(np.exp(-2 * Hl[a, :Nx]) + np.exp(-2 * Hl[b, :Nx])) + eye
            t = 1 / np.sqrt(det_ld(R))
            ij = v / np.exp(2 * Hl[b, :Nx])[None, :]
            Q1 = solve_ld(R, S * ld(0.5))
            aQ, bQ = ii @ Q1, (-ij) @ Q1
            maha = np.sum(aQ * ii, axis=1)[:, None] + np.sum(bQ * (-ij), axis=1)[None, :] - 2 * aQ @ (-ij).T
            Q = np.exp(log_k[:, a][:, None] + log_k[:, b][None, :] + maha)
            s = beta[a] @ Q @ beta[b]
            if a == b:                                   # - sum_ij K^-1_ij Q_ij = - trace(K^-1 Q), columns refined in longdouble
                Xs = cho_solve(cfs[a], Q.astype(np.float64)).astype(ld)
                for _ in range(3):
                    Xs = Xs + cho_solve(cfs[a], (Q - Kl[a] @ Xs).astype(np.float64)).astype(ld)
                s = s - np.trace(Xs)
            cov[a, b] = cov[b, a] = t * s
        cov[a, a] += np.exp(2 * Hl[a, Nx])
    cov = cov - np.outer(mean, mean)
    return mean.astype(np.float64), cov.astype(np.float64)


def check_c3_size_step(h, p, outs=(1, 4), node=3):
    """One propagation step at the full C3 size against the ORACLE's own factors: the oracle
    fits `outs` of the model's outputs from scratch (expanded-form K, LAPACK Cholesky, LU solves: optimize.py:303-356,
    :483-494) and evaluates ME / TA / EM at one node (gp_functions.py:72-173, :344-418); the device predicts all outputs
    from its own fit.  EM's covariance between two outputs only involves those two outputs, so the sub-model is exact.
    Bars as at the small sizes: mean 1e-10 of sum |ks alpha|, variances 1e-10 sf^2 (ME), 1e-9 of the largest entry (TA),
    1e-9 of the cancellation scale (EM)."""
    X, Y, H = p['X'], p['Y'], p['hyper']
    d = X.shape[1]
    outs = list(outs)
    Xs, Ys, Hs = X, Y[:, outs], H[outs]
    o = go.fit(Xs, Ys, Hs)                                   # with K^-1 (EM)
    z, S = p['Z'][node], p['Sigma'][node]
    sf2 = Hs[:, d] ** 2
    ms = mean_scale(Xs, z[None], Hs, o['alpha'])[0]
    om, ov, oJ = go.mean_var_jac(z[None], Xs, Hs, o['alpha'], o['chol'])
    # ME
    m, c = h.predict('ME', z[None], S[None])
    assert np.max(np.abs(m[0][outs] - om[0]) / ms) <= 1e-10, np.max(np.abs(m[0][outs] - om[0]) / ms)
    assert np.max(np.abs(np.diag(c[0])[outs] - ov[0]) / sf2) <= 1e-10, np.max(np.abs(np.diag(c[0])[outs] - ov[0]) / sf2)
    # TA
    m, c = h.predict('TA', z[None], S[None])
    oc = go.ta_cov(ov, oJ, S[None])[0]
    got = c[0][np.ix_(outs, outs)]
    assert np.max(np.abs(m[0][outs] - om[0]) / ms) <= 1e-10
    assert np.max(np.abs(got - oc)) <= 1e-9 * max(np.abs(oc).max(), sf2.max()), np.max(np.abs(got - oc))
    # EM
    m, c = h.predict('EM', z[None], S[None])
    em, ec = go.exact_moment(o['invK'], Xs, Ys, Hs, z, S)
    got = c[0][np.ix_(outs, outs)]
    bar = 1e-9 * (_em_scale(o['invK'], Xs, Ys, Hs, z, S).max() + sf2.max())
    assert np.max(np.abs(m[0][outs] - em.reshape(-1))) <= 1e-10 * max(1.0, ms.max()), np.max(np.abs(m[0][outs] - em.reshape(-1)))
    assert np.max(np.abs(got - ec)) <= bar, (np.max(np.abs(got - ec)), bar)
    # how many digits of the matrix MPC factors next (mpc_class.py:345,422) that bar pins: the cancellation scale against
    # |cov| and the device-vs-oracle difference against |cov| (check_em_against_extended_precision supplies the yardstick:
    # how far an fp64 evaluation of this closed form is from its exact value)
    scale = _em_scale(o['invK'], Xs, Ys, Hs, z, S)
    cmax = np.abs(ec).max()
    print(f'\n[C3 EM digits, N = {len(X)}] max|cov| {cmax:.3e}  cancellation scale / |cov| {scale.max() / cmax:.2e}  device vs oracle / |cov| '
          f'{np.abs(got - ec).max() / cmax:.2e} (gated at 1e-9 scale / |cov| = {bar / cmax:.2e})  min eig(cov): device {np.linalg.eigvalsh(got).min():.3e} oracle {np.linalg.eigvalsh(ec).min():.3e}')
    return dict(ms=ms, bar_em=bar, em_scale_over_cov=float(scale.max() / cmax), em_dev_vs_oracle_over_cov=float(np.abs(got - ec).max() / cmax))


def check_em_against_extended_precision(lib, N=1024, d=8, Ny=2, seed=1234, sn=1e-2, nodes=(3, 7)):
    """VERDICT r05 'EM digits': the exact-moment covariance of the C3 generator (d = 8, sn = 1e-2) at a size where the closed form
    can be evaluated in longdouble on the host (exact_moment_longdouble: N^3 longdouble work per output).  Gate: the device is
    within an order of magnitude of the fp64 oracle's own distance from that value, or has eight digits of the covariance --
    what separates either from the exact value is fp64 arithmetic on K^-1 (cond(K) eps) -- and the digits are printed."""
    p = go.synthetic_problem(N, d, Ny, max(nodes) + 1, seed=seed, sn=sn)
    X, Y, H = p['X'], p['Y'], p['hyper']
    h = Handle(lib, X, Y)
    h.fit(H, want_invK=True)
    f = h.get_factors(invK=True)
    o = go.fit(X, Y, H)
    sf2 = H[:, d] ** 2
    out = []
    for node in nodes:
        z, S = p['Z'][node], p['Sigma'][node]
        m, c = h.predict('EM', z[None], S[None])
        om, oc = go.exact_moment(o['invK'], X, Y, H, z, S)
        tm, tc = exact_moment_longdouble(X, Y, H, z, S)
        scale = _em_scale(o['invK'], X, Y, H, z, S).max() + sf2.max()
        cmax = np.abs(tc).max()
        d_dev, d_orc = np.abs(c[0] - tc).max(), np.abs(oc - tc).max()
        print(f'[EM vs longdouble closed form, N={N} node {node}] max|cov| {cmax:.3e} scale/|cov| {scale / cmax:.2e};  |err|/|cov|: device {d_dev / cmax:.2e} '
              f'oracle {d_orc / cmax:.2e}  device vs oracle {np.abs(c[0] - oc).max() / cmax:.2e};  mean |err|: device {np.abs(m[0] - tm).max():.2e} oracle {np.abs(om - tm).max():.2e}')
        # gate: within an order of magnitude of the fp64 oracle's own distance from the exact value, or eight digits of the
        # covariance (MI355X, N = 1024, sn = 1e-2: device 1.9e-9, oracle 3.4e-10 of max|cov| at a cancellation scale of 8e8 |cov|;
        # "1e-10 of the cancellation scale", the bar of the device-vs-oracle checks, would allow 8e-2 here and says nothing
        # about the matrix MPC factors next)
        assert d_dev <= max(10.0 * d_orc, 1e-8 * cmax), (d_dev, d_orc, cmax)
        assert np.abs(m[0] - tm).max() <= max(2.0 * np.abs(om - tm).max(), 1e-10 * max(1.0, np.abs(tm).max()))
        out.append((float(d_dev / cmax), float(d_orc / cmax)))
    h.close()
    return out


def check_callback_pattern(h, X, H, alpha, chol, Z, S, repeats=5):
    """One NLP-callback evaluation: Nt nodes, value + mean Jacobian + TA covariance from one `gpmpc_predict_jac` call,
    against the oracle evaluated on the given factors."""
    d = X.shape[1]
    m, c, J = h.predict_jac('TA', Z, S)
    om, ov, oJ = go.mean_var_jac(Z, X, H, alpha, chol)
    oc = go.ta_cov(ov, oJ, S)
    ms = mean_scale(X, Z, H, alpha)
    assert np.max(np.abs(m - om) / ms) <= 1e-10
    assert np.max(np.abs(J - oJ) / (ms / H[:, :d].min(axis=1))[..., None]) <= 1e-10
    assert np.max(np.abs(np.einsum('baa->ba', c) - np.einsum('baa->ba', oc)) / H[:, d] ** 2) <= 1e-9
    assert np.max(np.abs(c - oc)) <= 1e-9 * max(np.abs(oc).max(), (H[:, d] ** 2).max())
    for it in range(repeats):                  # repeated calls (an NLP iteration loop) are deterministic
        m2, c2, J2 = h.predict_jac('TA', Z, S)
        assert np.array_equal(m, m2) and np.array_equal(c, c2) and np.array_equal(J, J2)


def check_random_restarts(lib, X, Y, multistart=16, maxiter=3, min_finite=8):
    """C4 at world = 1: seeded Latin-hypercube restarts with an iteration cap, arg-min as optimize.py:474.
    NLL* is checked against the oracle's `calc_NLL_numpy` restatement at theta*, per-restart objectives are
    reproduced bitwise by a second run, and the arg-min objective is the device NLL at theta*."""
    from gp_mpc_amd.train import train_gp
    N, d = X.shape
    runs = []
    for rep in range(2):
        h = Handle(lib, X, Y)
        opt = train_gp(h, X, Y, multistart=multistart, random_restarts=True, seed=1234, numpy_path_conventions=False,
                       optimizer_opts={'maxiter': maxiter})
        runs.append((opt, h.get_factors(chol=False)['alpha'].copy()))
        if rep == 1:
            for a in range(Y.shape[1]):
                th = opt['hyper'][a]
                best = float(np.min(opt['obj'][a]))
                # arg-min objective == device NLL at theta* (the search evaluates through the batched execution, gpmpc_nll
                # through the single-matrix one: same number to rounding)
                ref = go.nll(th, X, Y[:, a])                                  # calc_NLL_numpy restatement, host
                sf2, sn2 = th[d] ** 2, th[d + 1] ** 2
                tol = max(1e-10, 50 * np.finfo(float).eps * N * (sf2 + sn2) / sn2)   # y^T K^-1 y is cond-limited
                assert abs(h.nll(a, th) - best) <= 0.1 * tol * (abs(best) + len(X))
                assert abs(best - ref) <= tol * (abs(ref) + N), (best, ref, tol)
                assert np.isfinite(opt['obj'][a]).sum() >= min_finite
        h.close()
    (o1, a1), (o2, a2) = runs
    assert np.array_equal(o1['hyper'], o2['hyper']) and np.array_equal(o1['obj'], o2['obj']) and np.array_equal(a1, a2)


def check_two_handles_two_threads(lib, N, d=6, B=256, reps=6):
    """include/gpmpc.h: calls on different handles are thread-safe.  Two models fitted and queried from two threads on
    ONE GPU: the persistent-kernel factorisation wants the whole chip, so the two fits compete for it."""
    import threading
    probs = [go.synthetic_problem(N, d, 1, B, seed=s, sn=1e-2) for s in (1, 2)]
    refs = []
    for p in probs:                                     # sequential reference results, one handle at a time
        h = Handle(lib, p['X'], p['Y'])
        assert np.all(h.fit(p['hyper']) == 0)
        refs.append(h.predict_mean_var(p['Z']))
        assert h.counter('handoff_timeouts') == 0 and h.counter('chained_factorisations') == 1   # alone: no time-out
        h.close()
    handles = [Handle(lib, p['X'], p['Y']) for p in probs]
    errs = []

    def work(i):
        try:
            for rep in range(reps):
                assert np.all(handles[i].fit(probs[i]['hyper']) == 0)
                m, v = handles[i].predict_mean_var(probs[i]['Z'])
                assert np.max(np.abs(m - refs[i][0])) <= 1e-9 * np.abs(refs[i][0]).max()
                assert np.max(np.abs(v - refs[i][1])) <= 1e-10
        except BaseException as e:                       # noqa: BLE001 (reported to the main thread)
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    stats = [(hh.counter('chained_factorisations'), hh.counter('single_queue_factorisations'), hh.counter('handoff_timeouts'))
             for hh in handles]
    print('\n[two handles] (chained, single-queue, time-outs) per handle:', stats)
    for hh in handles:
        hh.close()
    assert not errs, errs
    # the factorisations take turns on the device, so the persistent path must have served most of them even with the
    # other handle's prediction kernels in flight
    assert all(st[0] >= reps // 2 for st in stats), stats


def check_mean_functions(lib, N=150, d=3, Ny=2, seed=31):
    """f4: prior mean functions 'const' / 'linear' / 'polynomial' (gp_functions.py:25-69) through fit, NLL (+ gradient),
    prediction with and without m(z) (build_gp's meanFunc argument vs GP.__init__'s call), data update, training."""
    from gp_mpc_amd._lib import GpmpcError, EINVAL
    from gp_mpc_amd.gp import GP
    p = go.synthetic_problem(N, d, Ny, 9, seed=seed, sn=0.1)
    X, Y, Hk, Z, S = p['X'], p['Y'], p['hyper'], p['Z'], p['Sigma']
    Y = Y + 0.4 + 0.3 * X[:, :1]                       # something for a mean function to explain
    rng = np.random.default_rng(seed)
    h0 = Handle(lib, X, Y)
    h0.fit(Hk, want_invK=True)
    em0 = h0.predict('EM', Z[:3], S[:3] * 20)
    h0.close()
    for func in ('const', 'linear', 'polynomial'):
        hm = go.mean_param_count(func, d)
        H = np.hstack([Hk, rng.uniform(-0.3, 0.3, (Ny, hm))])
        o = go.fit_mean(X, Y, H, func)
        h = Handle(lib, X, Y)
        h.set_mean_func(func, False)
        assert h.nh == d + 2 + hm
        assert np.all(h.fit(H) == 0)
        f = h.get_factors()
        assert np.array_equal(f['hyper'], H)
        for a in range(Ny):
            assert relF(f['chol'][a], o['chol'][a]) <= 1e-10 and relF(f['alpha'][a], o['alpha'][a]) <= 1e-9
            v, g = h.nll(a, H[a], want_grad=True)
            ov, og = go.nll_mean_grad(H[a], X, Y[:, a], func)
            assert abs(v - go.nll_mean(H[a], X, Y[:, a], func)) / (abs(ov) + N) <= 1e-10
            assert np.max(np.abs(g - og) / (np.abs(og) + 1e-3 * np.abs(og).max())) <= 1e-6, (func, g, og)
        if func == 'linear':      # Gaussian hyper-priors of calc_NLL (optimize.py:77-97), value and gradient
            prior = dict(ell_mean=10.0, ell_std=10.0, sf_mean=10.0, sf_std=10.0, sn_mean=1e-5, sn_std=1e-2)   # (:158-165)
            h.set_hyper_prior(prior)
            v, g = h.nll(0, H[0], want_grad=True)
            ref = go.nll_prior(H[0], X, Y[:, 0], prior, func)
            assert abs(v - ref) / (abs(ref) + N) <= 1e-10
            for k in range(d + 2):
                e = np.zeros(h.nh)
                e[k] = 1e-6 * max(1.0, abs(H[0, k]))
                fd = (go.nll_prior(H[0] + e, X, Y[:, 0], prior, func) - go.nll_prior(H[0] - e, X, Y[:, 0], prior, func)) / (2 * e[k])
                assert abs(fd - g[k]) <= 1e-5 * (abs(g[k]) + 1e-3 * np.abs(g).max()), (k, fd, g[k])
            h.set_hyper_prior(None)
            assert abs(h.nll(0, H[0]) - go.nll_mean(H[0], X, Y[:, 0], func)) / (abs(ref) + N) <= 1e-10
        # GP.__init__'s predictor (build_gp without meanFunc): ks^T alpha only
        ms = mean_scale(X, Z, Hk, o['alpha'])
        mean, var = h.predict_mean_var(Z)
        om, ov_, oJ = go.mean_var_jac(Z, X, Hk, o['alpha'], o['chol'])
        assert np.max(np.abs(mean - om) / ms) <= 1e-10 and np.max(np.abs(var - ov_)) <= 1e-10
        # build_gp(meanFunc=func): + m(z), Jacobian + dm/dz, TA covariance through that Jacobian
        h.set_mean_func(func, True)
        h.fit(H)
        mean, cov, J = h.predict_jac('TA', Z, S)
        om, ov_, oJ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'], mean_func=func)
        assert np.max(np.abs(mean - om) / (ms + 1.0)) <= 1e-10
        assert np.max(np.abs(J - oJ)) <= 1e-10 * max(1.0, np.abs(oJ).max())
        assert np.max(np.abs(cov - go.ta_cov(ov_, oJ, S))) <= 1e-10
        m1, v1, J1, Hm, dvar = h.predict_sens(Z[:2])
        oH, odv = go.mean_var_sens(Z[:2], X, Hk, o['alpha'], o['chol'])
        if func == 'polynomial':
            for a in range(Ny):
                oH[:, a] += np.diag(2 * H[a, d + 2:d + 2 + d])
        assert np.max(np.abs(Hm - oH)) <= 1e-9 * max(1.0, np.abs(oH).max()) and np.max(np.abs(dvar - odv)) <= 1e-9
        assert np.array_equal(m1, mean[:2]) and np.array_equal(J1, J[:2])
        # the moment methods ignore the mean function like the reference (beta = K^-1 y, gp_functions.py:383)
        em = h.predict('EM', Z[:3], S[:3] * 20)
        assert np.allclose(em[0], em0[0], rtol=0, atol=1e-12) and np.allclose(em[1], em0[1], rtol=0, atol=1e-12)
        try:
            h.predict('old_TA', Z[:1], S[:1])
            assert False
        except GpmpcError as e:
            assert e.code == EINVAL
        # load_model path without alpha: recomputed from y - m(X)
        h2 = Handle(lib, X, Y)
        h2.set_mean_func(func, True)
        h2.set_factors(H, o['chol'])
        m2, _ = h2.predict_mean_var(Z)
        assert np.max(np.abs(m2 - om) / (ms + 1.0)) <= 1e-10
        h2.close()
        # data update keeps the mean function
        q = go.synthetic_problem(6, d, Ny, 1, seed=seed + 1, sn=0.1)
        h.append(q['X'], q['Y'])
        X2, Y2 = np.vstack([X, q['X']]), np.vstack([Y, q['Y']])
        o2 = go.fit_mean(X2, Y2, H, func, want_invK=False)
        m3, v3 = h.predict_mean_var(Z)
        om3, ov3, _ = go.mean_var_jac(Z, X2, H, o2['alpha'], o2['chol'], False, mean_func=func)
        assert np.max(np.abs(m3 - om3) / (ms + 1.0)) <= 1e-9 and np.max(np.abs(v3 - ov3)) <= 1e-10
        h.close()
    # training, IPOPT-path conventions (train_gp optimize.py:100-294): mean parameters are decision variables
    gp = GP(X[:60], Y[:60], mean_func='const', normalize=False, optimize_nummeric=False, gp_method='ME', lib=lib,
            optimizer_opts={'maxiter': 40})
    Ht = gp.train_info['hyper']
    assert Ht.shape == (Ny, d + 3)
    for a in range(Ny):
        h_init = np.concatenate([np.std(X[:60], 0), [np.std(Y[:60, a]), 1e-5, 0.0]])
        assert go.nll_mean(Ht[a], X[:60], Y[:60, a], 'const') < go.nll_mean(h_init, X[:60], Y[:60, a], 'const')
        assert -1e2 <= Ht[a, -1] <= 1e2 and abs(Ht[a, -1]) > 1e-3           # the constant moved off its start
    assert gp.get_hyper_parameters()['mean'].shape == (Ny, 2)                # off-by-one slice: [sn, c], gp_class.py:142
    gp.close()
    # numpy-path conventions: calc_NLL_numpy ignores the mean parameters, they stay 0 (optimize.py:377-379)
    gp = GP(X[:40], Y[:40], mean_func='linear', normalize=False, optimize_nummeric=True, gp_method='ME', lib=lib,
            optimizer_opts={'maxiter': 15})
    Hn = gp.train_info['hyper']
    assert Hn.shape == (Ny, d + 2 + d + 1) and np.all(Hn[:, d + 2:] == 0.0)
    gp.close()
    try:
        GP(X[:10], Y[:10], mean_func='cubic', lib=lib)
        assert False
    except NameError:
        pass


def check_feedback_rollout(lib, g, T=5):
    """a17 with feedback=True (gp_class.py:772-803): LQR gain from the model's own linearisation, u_t = K (mean_t -
    x_ref), control blocks of the input covariance; `GP.rollout` -> `gpmpc_rollout_feedback` against OracleGP.rollout,
    on a saved reference model (standardised, ME / TA) and on a well-conditioned synthetic one (all three methods)."""
    from gp_mpc_amd.gp import GP
    hyper = dict(hyper=g['hyper'], chol=g['chol'], alpha=g['alpha'], invK=g['invK'])
    kw = dict(normalize=g['normalize'], lib=lib)
    if g['normalize']:
        kw.update(meta=g['meta'], xlb=g['xlb'], xub=g['xub'], ulb=g['ulb'], uub=g['uub'])
    gp = GP(g['X'], g['Y'], hyper=hyper, gp_method='TA', **kw)
    og = go.OracleGP(g['X'], g['Y'], g['hyper'], g['chol'], g['alpha'], g['invK'], normalize=g['normalize'],
                     meta=g.get('meta'), gp_method='TA')
    N, Ny, Nu = gp.get_size()
    x = g['meta']['meanX'] + 0.3 * g['meta']['stdX'] if g['normalize'] else g['X'][3, :Ny]
    u = g['meta']['meanU'] - 0.2 * g['meta']['stdU'] if g['normalize'] else g['X'][3, Ny:]
    x_ref = x * 0.9
    U = np.tile(u, (T, 1))
    rng = np.random.default_rng(2)
    Kgain = 0.05 * rng.standard_normal((Nu, Ny))
    for Kin in (Kgain, None):                 # a given gain, and the LQR gain of the linearised model
        m, v, c = gp.rollout(x, U, methods=['TA', 'ME'], feedback=True, x_ref=x_ref, K=Kin, return_controls=True)
        om, ov = og.rollout(x, U, methods=('TA', 'ME'), feedback=True, x_ref=x_ref, K=Kin)
        sc = max(1.0, np.abs(om).max())
        assert np.allclose(c, og.controls, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(og.controls).max())), np.abs(c - og.controls).max()
        assert np.allclose(m, om, rtol=1e-7, atol=1e-8 * sc), np.abs(m - om).max()
        assert np.allclose(v, np.clip(ov, 0, None), rtol=1e-5, atol=1e-9 * max(1.0, np.abs(ov).max())), np.abs(v - ov).max()
    gp.close()
    p = go.synthetic_problem(120, 5, 3, T, seed=21, sn=0.1)
    o = go.fit(p['X'], p['Y'], p['hyper'])
    gp = GP(p['X'], p['Y'], hyper=dict(hyper=p['hyper'], chol=o['chol'], alpha=o['alpha'], invK=o['invK']),
            normalize=False, gp_method='EM', lib=lib)
    og = go.OracleGP(p['X'], p['Y'], p['hyper'], o['chol'], o['alpha'], o['invK'], gp_method='EM')
    x0, U = p['Z'][0, :3], p['Z'][:T, 3:] * 0.3
    m, v, c = gp.rollout(x0, U, methods=['EM', 'TA', 'ME'], feedback=True, Q=np.eye(3) * 2.0, R=np.eye(2) * 0.5,
                         return_controls=True)
    om, ov = og.rollout(x0, U, methods=('EM', 'TA', 'ME'), feedback=True, Q=np.eye(3) * 2.0, R=np.eye(2) * 0.5)
    assert np.max(np.abs(c - og.controls)) <= 1e-8 * max(1.0, np.abs(og.controls).max())
    assert np.max(np.abs(m - om)) <= 1e-8 * max(1.0, np.abs(om).max()) and np.max(np.abs(v - np.clip(ov, 0, None))) <= 1e-8
    assert np.max(np.abs(c[0] - U)) > 1e-3          # the controls really come from the feedback law, not from U
    gp.close()


def check_em_sens(lib, N=150, d=4, Ny=3, B=5, seed=13, tol=1e-9):
    """f1: value and Jacobians of 'EM' (`gpmpc_predict_em_sens`) against the oracle's closed forms, which
    tests/test_oracle.py pins by complex-step differentiation of the exact-moment formulas."""
    p = go.synthetic_problem(N, d, Ny, B, seed=seed, sn=0.1)
    X, Y, H, Z, S = p['X'], p['Y'], p['hyper'], p['Z'] * 0.5, p['Sigma'] * 40
    H = H.copy()
    H[:, :d] *= np.linspace(0.6, 1.3, d)[None, :]
    h = Handle(lib, X, Y)
    h.fit(H, want_invK=True)
    f = h.get_factors(invK=True)
    mean, cov, dm_dz, dm_dS, dc_dz, dc_dS = h.predict_em_sens(Z, S)
    m0, c0 = h.predict('EM', Z, S)
    assert np.array_equal(mean, m0) and np.array_equal(cov, c0)
    nv = h.predict_em_sens(Z, S, want_cov=False)        # Jacobian-only call: no covariance value, same derivatives
    assert nv[1] is None and np.array_equal(nv[0], mean)
    assert all(np.array_equal(a, b) for a, b in zip(nv[2:], (dm_dz, dm_dS, dc_dz, dc_dS)))
    sf2 = (H[:, d] ** 2).max()
    for b in range(B):
        o1, o2, o3, o4 = go.exact_moment_sens(f['invK'], X, Y, H, Z[b], S[b])
        sc = _em_scale(f['invK'], X, Y, H, Z[b], S[b]).max() + sf2          # cancellation scale of the pair sums
        ell = H[:, :d].min()
        assert np.max(np.abs(dm_dz[b] - o1)) <= tol * max(1.0, np.abs(o1).max()), (b, np.abs(dm_dz[b] - o1).max())
        assert np.max(np.abs(dm_dS[b] - o2)) <= tol * max(1.0, np.abs(o2).max()), (b, np.abs(dm_dS[b] - o2).max())
        assert np.max(np.abs(dc_dz[b] - o3)) <= tol * max(sc / ell, np.abs(o3).max()), (b, np.abs(dc_dz[b] - o3).max(), np.abs(o3).max())
        assert np.max(np.abs(dc_dS[b] - o4)) <= tol * max(sc / ell ** 2, np.abs(o4).max()), (b, np.abs(dc_dS[b] - o4).max(), np.abs(o4).max())
    # a direct end-to-end check as well: central differences of the device's own 'EM' value.  The pair sums cancel from
    # ~8e5 down to ~0.1 here, so a covariance value carries ~1e-10 of summation noise: with e = 1e-5 the difference
    # quotient was within 2.5x of the tolerance in the worst case (and failed once after an unrelated change moved K by
    # an ulp); e = 1e-4 leaves a factor 25, its truncation error is ~1e-9.
    e = 1e-4
    kc = min(1, d - 1)
    Zp, Zm = Z[:1].copy(), Z[:1].copy()
    Zp[0, kc] += e
    Zm[0, kc] -= e
    mp, cp = h.predict('EM', Zp, S[:1])
    mm, cm = h.predict('EM', Zm, S[:1])
    assert np.allclose((mp - mm)[0] / (2 * e), dm_dz[0][:, kc], rtol=1e-5, atol=1e-6)
    assert np.allclose((cp - cm)[0] / (2 * e), dc_dz[0][:, :, kc], rtol=1e-4, atol=1e-5)
    h.close()


def check_callback_blocks(lib, N=120, Ny=3, Nu=2, seed=17):
    """The numeric core of the casadi Callback (gp_mpc_amd/casadi_callback.py::jacobian_blocks): six Jacobian blocks
    in CasADi's column-major vec layout, for every propagation method, against central differences of `GP.predict`
    taken in that layout, on a standardised model (so the chain rule through GP.predict's scaling is exercised)."""
    from gp_mpc_amd.gp import GP
    from gp_mpc_amd.casadi_callback import jacobian_blocks, jacobian_dense
    Nx = Ny + Nu
    p = go.synthetic_problem(N, Nx, Ny, 2, seed=seed, sn=0.1)
    rng = np.random.default_rng(seed)
    meta = dict(meanY=rng.standard_normal(Ny), stdY=rng.uniform(0.5, 2.0, Ny), meanZ=rng.standard_normal(Nx),
                stdZ=rng.uniform(0.5, 2.0, Nx))
    meta.update(meanX=meta['meanZ'][:Ny], stdX=meta['stdZ'][:Ny], meanU=meta['meanZ'][Ny:], stdU=meta['stdZ'][Ny:])
    o = go.fit(p['X'], p['Y'], p['hyper'])
    gp = GP(p['X'], p['Y'], hyper=dict(hyper=p['hyper'], chol=o['chol'], alpha=o['alpha'], invK=o['invK']), normalize=True,
            meta=meta, xlb=np.zeros(Ny), xub=np.ones(Ny), ulb=np.zeros(Nu), uub=np.ones(Nu), lib=lib)
    z = meta['meanZ'] + 0.4 * meta['stdZ'] * rng.standard_normal(Nx)
    x, u = z[:Ny], z[Ny:]
    A = rng.standard_normal((Nx, Nx)) * 0.15
    S = A @ A.T + 1e-3 * np.eye(Nx)

    def both(zv, Sv):
        m, c = gp.predict(zv[:Ny], zv[Ny:], Sv)
        return np.concatenate([np.array(m).reshape(-1), np.array(c).reshape(-1, order='F')])
    for method in ('ME', 'TA', 'EM', 'old_ME', 'old_TA'):
        gp.set_method(method)
        blocks = jacobian_blocks(gp, x, u, S)
        shapes = [(Ny, Ny), (Ny, Nu), (Ny, Nx * Nx), (Ny * Ny, Ny), (Ny * Ny, Nu), (Ny * Ny, Nx * Nx)]
        assert [b.shape for b in blocks] == shapes
        # the single stacked Jacobian of CasADi 3.4 / 3.5 (README.md:18-19): the same numbers, side by side
        Jd = jacobian_dense(gp, x, u, S)
        assert Jd.shape == (Ny + Ny * Ny, Ny + Nu + Nx * Nx)
        cuts_r, cuts_c = [0, Ny, Ny + Ny * Ny], [0, Ny, Ny + Nu, Ny + Nu + Nx * Nx]
        for o in range(2):
            for i in range(3):
                assert np.allclose(Jd[cuts_r[o]:cuts_r[o + 1], cuts_c[i]:cuts_c[i + 1]], blocks[3 * o + i], rtol=1e-9, atol=1e-12)
        Jz = np.zeros((Ny + Ny * Ny, Nx))
        for k in range(Nx):
            e = np.zeros(Nx)
            e[k] = 1e-4          # (the EM covariance carries ~1e-10 of summation noise: smaller steps drown in it)
            Jz[:, k] = (both(z + e, S) - both(z - e, S)) / 2e-4
        # The input covariance is perturbed SYMMETRICALLY, entries (p, q) and (q, p) together -- the only kind of
        # perturbation an NLP built on a covariance matrix produces, and the only one the device's value path is
        # defined for (it exploits Sigma = Sigma^T) -- and compared with the sum of the two Jacobian columns.
        JS = np.zeros((Ny + Ny * Ny, Nx * Nx))
        fold = lambda Bk: np.stack([Bk[:, pp + Nx * qq] + (Bk[:, qq + Nx * pp] if pp != qq else 0.0)
                                    for qq in range(Nx) for pp in range(Nx)], axis=1)
        for qq in range(Nx):
            for pp in range(Nx):
                E = np.zeros((Nx, Nx))
                E[pp, qq] = E[qq, pp] = 1e-4
                JS[:, pp + Nx * qq] = (both(z, S + E) - both(z, S - E)) / 2e-4
        ref = [Jz[:Ny, :Ny], Jz[:Ny, Ny:], JS[:Ny], Jz[Ny:, :Ny], Jz[Ny:, Ny:], JS[Ny:]]
        got = [blocks[0], blocks[1], fold(blocks[2]), blocks[3], blocks[4], fold(blocks[5])]
        for i, (b, r) in enumerate(zip(got, ref)):
            assert np.allclose(b, r, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(r).max())), (method, i, np.abs(b - r).max())
        if method in ('ME', 'TA'):
            assert np.all(blocks[2] == 0.0)
        if method == 'EM':
            assert np.abs(blocks[2]).max() > 1e-4          # the EM mean does depend on Sigma
    gp.close()


def check_callback_batched(lib, N=120, Ny=3, Nu=2, Nt=4, seed=23):
    """The numeric core of the batched casadi Callback (all Nt shooting nodes of mpc_class.py:361-423 in one call):
    values against GP.predict node by node, the block-diagonal Jacobian -- triplets in CasADi's column-major vec layout of
    (X[Ny x Nt], U[Nu x Nt], C[Nx x Nx Nt]) -> (M[Ny x Nt], V[Ny x Ny Nt]) and its dense stacked form -- against central
    differences of the batched values in that layout, every propagation method, standardised model."""
    from gp_mpc_amd.gp import GP
    from gp_mpc_amd import casadi_callback as cb
    Nx = Ny + Nu
    p = go.synthetic_problem(N, Nx, Ny, 2, seed=seed, sn=0.1)
    rng = np.random.default_rng(seed)
    meta = dict(meanY=rng.standard_normal(Ny), stdY=rng.uniform(0.5, 2.0, Ny), meanZ=rng.standard_normal(Nx),
                stdZ=rng.uniform(0.5, 2.0, Nx))
    meta.update(meanX=meta['meanZ'][:Ny], stdX=meta['stdZ'][:Ny], meanU=meta['meanZ'][Ny:], stdU=meta['stdZ'][Ny:])
    o = go.fit(p['X'], p['Y'], p['hyper'])
    gp = GP(p['X'], p['Y'], hyper=dict(hyper=p['hyper'], chol=o['chol'], alpha=o['alpha'], invK=o['invK']), normalize=True,
            meta=meta, xlb=np.zeros(Ny), xub=np.ones(Ny), ulb=np.zeros(Nu), uub=np.ones(Nu), lib=lib)
    Zn = meta['meanZ'][None, :] + 0.4 * meta['stdZ'][None, :] * rng.standard_normal((Nt, Nx))
    X, U = Zn[:, :Ny].T.copy(), Zn[:, Ny:].T.copy()                      # [Ny x Nt], [Nu x Nt]
    Cs = []
    for t in range(Nt):
        A = rng.standard_normal((Nx, Nx)) * 0.15
        Cs.append(A @ A.T + 1e-3 * np.eye(Nx))
    C = np.concatenate(Cs, axis=1)                                       # [Nx x Nx Nt]

    def stacked(Xv, Uv, Cv):
        M, V = cb.batched_values(gp, Xv, Uv, Cv)
        return np.concatenate([M.reshape(-1, order='F'), V.reshape(-1, order='F')])
    nin = [Ny * Nt, Nu * Nt, Nx * Nx * Nt]
    for method in ('ME', 'TA', 'EM', 'old_ME', 'old_TA'):
        gp.set_method(method)
        M, V = cb.batched_values(gp, X, U, C)
        assert M.shape == (Ny, Nt) and V.shape == (Ny, Ny * Nt)
        for t in range(Nt):
            m1, c1 = gp.predict(X[:, t], U[:, t], Cs[t])
            assert np.allclose(M[:, t], np.array(m1).reshape(-1), rtol=1e-9, atol=1e-12), (method, t)
            assert np.allclose(V[:, Ny * t:Ny * (t + 1)], c1, rtol=1e-8, atol=1e-12 * max(1.0, np.abs(c1).max())), (method, t)
        Jd = cb.batched_jacobian_dense(gp, X, U, C)
        assert Jd.shape == (Ny * Nt + Ny * Ny * Nt, sum(nin))
        trip = cb.batched_jacobian_triplets(gp, X, U, C)
        assert len(trip) == 6 and [t_[3] for t_ in trip] == [(r, c) for r in (Ny * Nt, Ny * Ny * Nt) for c in nin]
        assert sum(len(t_[2]) for t_ in trip) == Nt * (Ny + Ny * Ny) * (Ny + Nu + Nx * Nx)      # block diagonal: Nt dense blocks
        # central differences in the vec layout; the covariance blocks are perturbed symmetrically (see check_callback_blocks)
        ref = np.zeros_like(Jd)
        base = [X.reshape(-1, order='F'), U.reshape(-1, order='F'), C.reshape(-1, order='F')]
        shp = [X.shape, U.shape, C.shape]
        col = 0
        for i in range(2):
            for k in range(nin[i]):
                v = [b.copy() for b in base]
                w = [b.copy() for b in base]
                v[i][k] += 1e-4
                w[i][k] -= 1e-4
                ref[:, col] = (stacked(*[a.reshape(s_, order='F') for a, s_ in zip(v, shp)])
                               - stacked(*[a.reshape(s_, order='F') for a, s_ in zip(w, shp)])) / 2e-4
                col += 1
        got = Jd.copy()
        for t in range(Nt):
            for qq in range(Nx):
                for pp in range(Nx):
                    E = np.zeros((Nx, Nx * Nt))
                    E[pp, Nx * t + qq] = E[qq, Nx * t + pp] = 1e-4
                    k = col + pp + Nx * (Nx * t + qq)
                    ref[:, k] = (stacked(X, U, C + E) - stacked(X, U, C - E)) / 2e-4
                    k2 = col + qq + Nx * (Nx * t + pp)
                    if pp != qq:
                        got[:, k] = Jd[:, k] + Jd[:, k2]         # the two columns a symmetric perturbation moves together
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max())), (method, np.abs(got - ref).max())
        # node t's outputs do not depend on node s's inputs
        mask = np.zeros_like(Jd, dtype=bool)
        roff, coff = [0, Ny * Nt], [0, Ny * Nt, Ny * Nt + Nu * Nt]
        for kk, (rows, cols, vals, shape) in enumerate(trip):
            mask[rows + roff[kk // 3], cols + coff[kk % 3]] = True
        assert np.all(Jd[~mask] == 0.0)
    gp.close()


def _callback_model(lib, N, Ny, Nu, seed):
    """A standardised GP (so the chain rule through GP.predict's scaling is exercised) and its oracle twin."""
    from gp_mpc_amd.gp import GP
    Nx = Ny + Nu
    p = go.synthetic_problem(N, Nx, Ny, 2, seed=seed, sn=0.1)
    rng = np.random.default_rng(seed)
    meta = dict(meanY=rng.standard_normal(Ny), stdY=rng.uniform(0.5, 2.0, Ny), meanZ=rng.standard_normal(Nx),
                stdZ=rng.uniform(0.5, 2.0, Nx))
    meta.update(meanX=meta['meanZ'][:Ny], stdX=meta['stdZ'][:Ny], meanU=meta['meanZ'][Ny:], stdU=meta['stdZ'][Ny:])
    o = go.fit(p['X'], p['Y'], p['hyper'])
    gp = GP(p['X'], p['Y'], hyper=dict(hyper=p['hyper'], chol=o['chol'], alpha=o['alpha'], invK=o['invK']), normalize=True,
            meta=meta, xlb=np.zeros(Ny), xub=np.ones(Ny), ulb=np.zeros(Nu), uub=np.ones(Nu), lib=lib)
    og = go.OracleGP(p['X'], p['Y'], p['hyper'], o['chol'], o['alpha'], o['invK'], normalize=True, meta=meta)
    return gp, og, meta, rng


def _oracle_vec(og, x, u, S):
    m, c = og.predict(x, u, S)
    return np.concatenate([np.asarray(m).reshape(-1), np.asarray(c).reshape(-1, order='F')])     # CasADi's column-major vec


def _oracle_jacobian(og, x, u, S, Ny, Nu, h=1e-3):
    """d [mean; vec(cov)] / d [x; u; vec(covar)] of the ORACLE's predict by fourth-order central differences (the exact-moment
    covariance is a difference of O(1) terms, good to ~1e-10 absolute: with the two-point stencil at h = 1e-5 that noise alone
    is 1e-5 in the quotient -- seen on the GPU tier at N = 200; the five-point stencil at h = 1e-3 keeps both error terms below
    1e-6); the input covariance is perturbed symmetrically (the only perturbation the value path is defined for, see
    check_callback_blocks), so the columns of the covariance block are to be compared folded: col(p, q) + col(q, p)."""
    Nx = Ny + Nu
    z = np.concatenate([x, u])
    J = np.zeros((Ny + Ny * Ny, Nx + Nx * Nx))

    def stencil(f):
        return (-f(2.0) + 8.0 * f(1.0) - 8.0 * f(-1.0) + f(-2.0)) / (12.0 * h)

    for k in range(Nx):
        e = np.zeros(Nx)
        e[k] = h
        J[:, k] = stencil(lambda s: _oracle_vec(og, (z + s * e)[:Ny], (z + s * e)[Ny:], S))
    for qq in range(Nx):
        for pp in range(Nx):
            E = np.zeros((Nx, Nx))
            E[pp, qq] = E[qq, pp] = h
            J[:, Nx + pp + Nx * qq] = stencil(lambda s: _oracle_vec(og, x, u, S + s * E))
    return J


def _fold_cov_columns(J, Nx, off):
    """columns off + p + Nx q of a Jacobian w.r.t. vec(covar): add the mirror column (what a symmetric perturbation moves)"""
    out = J.copy()
    for qq in range(Nx):
        for pp in range(Nx):
            if pp != qq:
                out[:, off + pp + Nx * qq] = J[:, off + pp + Nx * qq] + J[:, off + qq + Nx * pp]
    return out


def check_callback_classes(lib, ca, version, N=60, Ny=2, Nu=1, Nt=3, seed=31, methods=('ME', 'TA', 'EM', 'old_ME', 'old_TA')):
    """The casadi.Callback classes of gp_mpc_amd/casadi_callback.py EXECUTED through the Callback protocol (`ca`: a module
    with CasADi's API -- tests/stub_casadi.py here, casadi itself where it is installed), for the Jacobian convention of
    CasADi `version` (3.4 / 3.5: one stacked Jacobian, the reference's version README.md:18-19; >= 3.6: one block per
    (output, input) pair): construct, evaluate with CasADi-ordered arguments, ask for the Jacobian function and evaluate
    it.  Values against OracleGP.predict (gp_class.py:245-263, the call MPC makes at mpc_class.py:412-413), Jacobians
    against central differences of the ORACLE (not of the HIP path)."""
    from gp_mpc_amd import casadi_callback as cb
    Nx = Ny + Nu
    ca.set_version(version) if hasattr(ca, 'set_version') else None
    keep = cb.ca
    cb.ca = ca
    gp, og, meta, rng = _callback_model(lib, N, Ny, Nu, seed)
    try:
        layout = cb.jacobian_layout()
        assert layout == ('dense' if cb.casadi_version(version) < (3, 6) else 'blocks')
        Zn = meta['meanZ'][None, :] + 0.4 * meta['stdZ'][None, :] * rng.standard_normal((Nt, Nx))
        Cs = []
        for t in range(Nt):
            A = rng.standard_normal((Nx, Nx)) * 0.15
            Cs.append(A @ A.T + 1e-3 * np.eye(Nx))
        for method in methods:
            gp.set_method(method)
            og.set_method(method)
            vtol = 1e-9 if method in ('ME', 'TA') else 1e-7
            # ---- one node per call: the signature of GP.__predict (gp_class.py:212-224) ----
            f = cb.make_predict_callback(gp, name='gp_hip_' + method)
            assert (f.n_in(), f.n_out()) == (3, 2)
            assert [f.size_in(i) for i in range(3)] == [(Ny, 1), (Nu, 1), (Nx, Nx)] and [f.size_out(i) for i in range(2)] == [(Ny, 1), (Ny, Ny)]
            x, u, S = Zn[0, :Ny], Zn[0, Ny:], Cs[0]
            mean, cov = f(ca.DM(x), ca.DM(u), ca.DM(S))
            om, oc = og.predict(x, u, S)
            assert np.array(mean).shape == (Ny, 1) and np.array(cov).shape == (Ny, Ny)
            assert np.allclose(np.array(mean), om, rtol=vtol, atol=vtol), (method, np.abs(np.array(mean) - om).max())
            assert np.allclose(np.array(cov), oc, rtol=0, atol=vtol * max(1.0, np.abs(oc).max())), (method, np.abs(np.array(cov) - oc).max())
            Jf = f.jacobian()
            assert Jf.n_in() == 5 and Jf.n_out() == (1 if layout == 'dense' else 6)
            res = Jf(ca.DM(x), ca.DM(u), ca.DM(S), mean, cov)
            if layout == 'dense':
                Jd = np.array(res)
            else:
                b = [np.array(r) for r in res]
                assert [r.shape for r in b] == [(Ny, Ny), (Ny, Nu), (Ny, Nx * Nx), (Ny * Ny, Ny), (Ny * Ny, Nu), (Ny * Ny, Nx * Nx)]
                Jd = np.block([[b[0], b[1], b[2]], [b[3], b[4], b[5]]])
            assert Jd.shape == (Ny + Ny * Ny, Nx + Nx * Nx)
            ref = _oracle_jacobian(og, x, u, S, Ny, Nu)
            got = _fold_cov_columns(Jd, Nx, Nx)
            assert np.allclose(got, ref, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(ref).max())), (method, layout, np.abs(got - ref).max())
            # ---- all shooting nodes of mpc_class.py:361-423 in one call ----
            g = cb.make_batched_predict_callback(gp, Nt, name='gp_hip_nodes_' + method)
            X, U, C = Zn[:, :Ny].T.copy(), Zn[:, Ny:].T.copy(), np.concatenate(Cs, axis=1)
            assert [g.size_in(i) for i in range(3)] == [(Ny, Nt), (Nu, Nt), (Nx, Nx * Nt)]
            M, V = g(ca.DM(X), ca.DM(U), ca.DM(C))
            M, V = np.array(M), np.array(V)
            assert M.shape == (Ny, Nt) and V.shape == (Ny, Ny * Nt)
            for t in range(Nt):
                om, oc = og.predict(X[:, t], U[:, t], Cs[t])
                assert np.allclose(M[:, t:t + 1], om, rtol=vtol, atol=vtol), (method, t)
                assert np.allclose(V[:, Ny * t:Ny * (t + 1)], oc, rtol=0, atol=vtol * max(1.0, np.abs(oc).max())), (method, t)
            Jg = g.jacobian()
            res = Jg(ca.DM(X), ca.DM(U), ca.DM(C), ca.DM(M), ca.DM(V))
            nin, nout = [Ny * Nt, Nu * Nt, Nx * Nx * Nt], [Ny * Nt, Ny * Ny * Nt]
            if layout == 'dense':
                assert res.sparsity().nnz() == Nt * (Ny + Ny * Ny) * (Nx + Nx * Nx)          # declared block diagonal
                Jb = np.array(res)
            else:
                bb = [np.array(r) for r in res]
                assert [r.shape for r in bb] == [(r_, c_) for r_ in nout for c_ in nin]
                assert sum(r.sparsity().nnz() for r in res) == Nt * (Ny + Ny * Ny) * (Nx + Nx * Nx)
                Jb = np.block([[bb[0], bb[1], bb[2]], [bb[3], bb[4], bb[5]]])
            assert Jb.shape == (sum(nout), sum(nin))
            # node t's block against the oracle's single-node Jacobian; everything off the block diagonal is exactly zero
            seen = np.zeros_like(Jb, dtype=bool)
            for t in range(Nt):
                ref = _oracle_jacobian(og, X[:, t], U[:, t], Cs[t], Ny, Nu)
                rows = np.concatenate([Ny * t + np.arange(Ny), nout[0] + Ny * Ny * t + np.arange(Ny * Ny)])
                cols = np.concatenate([Ny * t + np.arange(Ny), nin[0] + Nu * t + np.arange(Nu),
                                       nin[0] + nin[1] + Nx * Nx * t + np.arange(Nx * Nx)])
                blk = Jb[np.ix_(rows, cols)]
                got = _fold_cov_columns(blk, Nx, Nx)
                assert np.allclose(got, ref, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(ref).max())), (method, layout, t, np.abs(got - ref).max())
                seen[np.ix_(rows, cols)] = True
            assert np.all(Jb[~seen] == 0.0)
            assert f.n_eval >= 1 and Jf.n_eval == 1 and g.n_eval == 1 and Jg.n_eval == 1 if hasattr(f, 'n_eval') else True
        # the Callback refuses what CasADi would refuse: wrong argument shapes
        bad = False
        try:
            f(ca.DM(np.zeros(Ny + 1)), ca.DM(u), ca.DM(S))
        except RuntimeError:
            bad = True
        assert bad
        cb.ca = None                       # without casadi the factories say so (the GP itself keeps working)
        for make in (lambda: cb.make_predict_callback(gp), lambda: cb.make_batched_predict_callback(gp, Nt)):
            try:
                make()
                raised = False
            except ImportError:
                raised = True
            assert raised
    finally:
        cb.ca = keep
        gp.close()


def check_training_active_bound(lib, t):
    """a8 on the second reference-made fixture (train_small2.npz): the optimum lies ON the upper bound of sn.  Both
    optimisers over the device's NLL + gradient -- scipy SLSQP through the GP class and the native projected L-BFGS behind
    `gpmpc_train_multistart` -- must end at least as low as the reference's optimum, on the bound, at its length scales."""
    from gp_mpc_amd.train import train_gp
    check_training(lib, t)
    X, Y, d = t['X'], t['Y'], t['X'].shape[1]
    h = Handle(lib, X, Y)
    opt = train_gp(h, X, Y, multistart=1, numpy_path_conventions=True, optimizer='native')
    H = opt['hyper']
    ours = go.nll(H[0], X, Y[:, 0])
    assert ours <= t['nll'][0] + 1e-6 * abs(t['nll'][0]), (ours, t['nll'][0])
    assert abs(H[0, d + 1] - 1e-2) <= 1e-6 and np.allclose(H[0], t['hyper'][0], rtol=5e-2), (H[0], t['hyper'][0])
    h.close()


def check_training_beats_failed_reference_search(lib, t):
    """a8 on the third reference-made fixture (train_small3.npz): where the reference's finite-difference SLSQP ends on a
    bound or never leaves its start, both optimisers over the device's NLL + analytic gradient must end at least as low --
    far lower for the two outputs the reference leaves at NLL 8e5 and 3e7 -- with finite factors."""
    from gp_mpc_amd.train import train_gp
    X, Y, d = t['X'], t['Y'], t['X'].shape[1]
    for optimizer in ('native', 'scipy'):
        h = Handle(lib, X, Y)
        opt = train_gp(h, X, Y, multistart=1, numpy_path_conventions=True, optimizer=optimizer)
        H = opt['hyper']
        for a in range(Y.shape[1]):
            ours = go.nll(H[a], X, Y[:, a])
            assert ours <= t['nll'][a] + 1e-6 * abs(t['nll'][a]), (optimizer, a, ours, t['nll'][a])
        if optimizer == 'native':
            assert go.nll(H[1], X, Y[:, 1]) < 1e3 and go.nll(H[2], X, Y[:, 2]) < 1e3      # it does leave the start
        f = h.get_factors()
        assert np.all(np.isfinite(f['chol'])) and np.all(np.isfinite(f['alpha']))
        h.close()


def check_training_never_worse(lib, n_cases=6, seed=5):
    """Seeded data sets (standardised, 1-3 inputs, noise 3e-4 .. 0.1) from the reference's single start: the native search
    behind `gpmpc_train_multistart` -- the default of `GP` / `train_gp` -- must end at or below BOTH the restated reference
    (`go.train`: SLSQP with finite differences, which reproduces `train_gp_numpy` on the fixtures) and SLSQP with the
    device's analytic gradient.  (Found with this sweep: the latter can end far above the reference, and both SLSQP variants
    stay at the start, NLL ~1e9, when the noise exceeds the sn bound.)"""
    from gp_mpc_amd.train import train_gp
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        N, d = int(rng.integers(25, 60)), int(rng.integers(1, 4))
        X = rng.standard_normal((N, d))
        w = rng.standard_normal((d,))
        noise = 10 ** rng.uniform(-3.5, -1)
        y = np.sin(X @ w) + 0.4 * np.cos(X[:, 0] * 1.7) + noise * rng.standard_normal(N)
        Y = ((y - y.mean()) / y.std())[:, None]
        res = {}
        for optimizer in ('native', 'scipy'):
            h = Handle(lib, X, Y)
            opt = train_gp(h, X, Y, multistart=1, numpy_path_conventions=True, optimizer=optimizer)
            res[optimizer] = go.nll(opt['hyper'][0], X, Y[:, 0])
            h.close()
        ref = go.nll(go.train(X, Y, multistart=1)['hyper'][0], X, Y[:, 0])
        best = min(ref, res['scipy'])
        assert res['native'] <= best + 1e-4 * (abs(best) + N), (case, N, d, noise, ref, res)


def check_training_native(lib, t):
    """a8 behind the C ABI (`gpmpc_train_multistart`): from the reference's initial point inside the reference's box
    (both conventions) the native projected L-BFGS must reach an NLL at least as good as `train_gp_numpy`'s SLSQP
    optimum frozen in train_small.npz, report obj == device NLL at theta*, and leave the model fitted there."""
    from gp_mpc_amd.train import train_gp
    X, Y = t['X'], t['Y']
    Ny, d = Y.shape[1], X.shape[1]
    for numpy_path in (True, False):
        h = Handle(lib, X, Y)
        opt = train_gp(h, X, Y, multistart=2, numpy_path_conventions=numpy_path, optimizer='native')
        H = opt['hyper']
        for a in range(Ny):
            ours = go.nll(H[a], X, Y[:, a])
            assert ours <= t['nll'][a] + 1e-6 * abs(t['nll'][a]), (numpy_path, a, ours, t['nll'][a])
            K = go.gram(X, H[a, :d], H[a, d] ** 2, H[a, d + 1] ** 2)
            tol = max(1e-10, 50 * np.finfo(float).eps * np.linalg.cond(K))      # y^T K^-1 y is cond-limited
            assert abs(opt['obj'][a].min() - ours) <= tol * (abs(ours) + len(X)), (opt['obj'][a].min(), ours, tol)
            assert opt['obj'][a, 0] == opt['obj'][a, 1]                  # identical starts -> identical restarts
        assert np.allclose(H[0], t['hyper'][0], rtol=5e-2)               # output 0 has a sharp optimum
        f = h.get_factors()
        o = go.fit(X, Y, H, want_invK=False)
        for a in range(Ny):
            assert relF(f['chol'][a], o['chol'][a]) <= 1e-9
        h.close()
    # with a mean function on the IPOPT-path conventions, and the error path: an empty box is refused
    h = Handle(lib, X, Y)
    opt = train_gp(h, X, Y, multistart=1, numpy_path_conventions=False, optimizer='native', mean_func='const')
    for a in range(Ny):
        assert go.nll_mean(opt['hyper'][a], X, Y[:, a], 'const') <= t['nll'][a] + 1e-6 * abs(t['nll'][a])
    from gp_mpc_amd._lib import GpmpcError, EINVAL
    try:
        h.train_multistart(np.ones((Ny, 1, h.nh)), np.ones((Ny, h.nh)), np.zeros((Ny, h.nh)))
        assert False
    except GpmpcError as e:
        assert e.code == EINVAL
    h.close()


def check_train_lockstep_invariance(lib, N, d, nstart, max_iter, seed=5, mean_func='zero'):
    """The lock-step restart search (api_train.inl): a rank's restarts advance together and their evaluation points go
    through the device as batches.  A point's value and gradient must not depend on what else is in its batch: the same
    search with batches of any size and with one point at a time (`train_batch_cap` = 1: the same kernels on a batch of
    one) returns the same table of (NLL, theta) bit for bit -- which is what keeps the restart shard world-size
    invariant.  NLL* is the oracle's NLL at theta*."""
    from gp_mpc_amd.train import lhs_starts, bounds_ipopt_path
    p = go.synthetic_problem(N, d, 1, 1, seed=seed, sn=0.05)
    X, Y = p['X'], p['Y']
    res = []
    # (cap, vargemm_persist): the K^-1 product of the gradient runs as one persistent launch over a static schedule when the
    # batch is large, as one tile per workgroup when it is small -- forced both ways here: same bits
    for cap, persist in ((0, -1), (1, -1), (3, -1), (0, 2), (0, 0)):
        h = Handle(lib, X, Y)
        if mean_func != 'zero':
            h.set_mean_func(mean_func, add_to_prediction=False)
        lb, ub = bounds_ipopt_path(d)
        nm = h.nh - (d + 2)
        lb, ub = np.concatenate([lb, -5.0 * np.ones(nm)]), np.concatenate([ub, 5.0 * np.ones(nm)])
        starts = lhs_starts(nstart, lb, ub, 1234)
        if nm:
            starts[:, d + 2:] = 0.1
        lib.set_tuning('train_batch_cap', cap)
        lib.set_tuning('vargemm_persist', persist)
        try:
            res.append(h.train_multistart(starts[None], lb[None], ub[None], max_iter=max_iter))
        finally:
            lib.set_tuning('train_batch_cap', 0)
            lib.set_tuning('vargemm_persist', -1)
        if cap == 0 and persist == -1:
            th = res[-1]['hyper'][0]
            best = float(np.min(res[-1]['obj'][0]))
            ref = go.nll(th, X, Y[:, 0]) if mean_func == 'zero' else go.nll_mean(th, X, Y[:, 0], mean_func)
            tol = max(1e-10, 50 * np.finfo(float).eps * N * (th[d] ** 2 + th[d + 1] ** 2) / th[d + 1] ** 2)
            assert abs(best - ref) <= tol * (abs(ref) + N), (best, ref, tol)
            assert np.isfinite(res[-1]['obj']).sum() >= max(1, nstart // 2)
        h.close()
    for k in ('hyper', 'obj', 'theta'):
        for other in res[1:]:
            assert np.array_equal(res[0][k], other[k], equal_nan=True), k
    assert len({r['evaluations'] for r in res}) == 1


def check_gp_class_strict(lib, N=400, Ny=3, Nu=2, seed=29):
    """`GP.predict` / `discrete_linearize` / `validate` for all five methods against OracleGP on a WELL-CONDITIONED
    standardised model (sn = 0.1, cond(K) ~ 1e5) at the plain bars: 1e-10 relative to the value scale (the moment
    methods relative to their cancellation scale, which is O(1) here)."""
    from gp_mpc_amd.gp import GP
    Nx = Ny + Nu
    p = go.synthetic_problem(N, Nx, Ny, 8, seed=seed, sn=0.1)
    rng = np.random.default_rng(seed)
    meta = dict(meanY=rng.standard_normal(Ny), stdY=rng.uniform(0.5, 2.0, Ny), meanZ=rng.standard_normal(Nx),
                stdZ=rng.uniform(0.5, 2.0, Nx))
    meta.update(meanX=meta['meanZ'][:Ny], stdX=meta['stdZ'][:Ny], meanU=meta['meanZ'][Ny:], stdU=meta['stdZ'][Ny:])
    o = go.fit(p['X'], p['Y'], p['hyper'])
    gp = GP(p['X'], p['Y'], hyper=dict(hyper=p['hyper'], chol=o['chol'], alpha=o['alpha'], invK=o['invK']), normalize=True,
            meta=meta, xlb=np.zeros(Ny), xub=np.ones(Ny), ulb=np.zeros(Nu), uub=np.ones(Nu), lib=lib)
    og = go.OracleGP(p['X'], p['Y'], p['hyper'], o['chol'], o['alpha'], o['invK'], normalize=True, meta=meta)
    sf2 = (p['hyper'][:, Nx] ** 2).max()
    for b in range(4):
        z = meta['meanZ'] + meta['stdZ'] * p['Z'][b] * 0.7
        x, u, S = z[:Ny], z[Ny:], p['Sigma'][b] * 30
        zs = (z - meta['meanZ']) / meta['stdZ']
        for m in ('ME', 'TA', 'EM', 'old_ME', 'old_TA'):
            gp.set_method(m)
            og.set_method(m)
            mean, cov = gp.predict(x, u, S)
            om, oc = og.predict(x, u, S)
            msc = mean_scale(p['X'], zs[None], p['hyper'], o['alpha'])[0] * meta['stdY']      # size of the terms of the mean sum
            assert np.max(np.abs(mean[:, 0] - om[:, 0]) / msc) <= 1e-10, (m, b, np.abs(mean - om).max())
            csc = _em_scale(o['invK'], p['X'], p['Y'], p['hyper'], zs, S).max() + sf2 if m == 'EM' else max(sf2, np.abs(oc).max())
            assert np.max(np.abs(cov - oc)) <= 1e-9 * csc, (m, b, np.abs(cov - oc).max(), csc)
        for m in ('TA', 'EM'):              # EM: Jacobian of the exact-moment mean (depends on the input covariance)
            gp.set_method(m)
            og.set_method(m)
            A, Bm = gp.discrete_linearize(x, u, S)
            oA, oB = og.discrete_linearize(x, u, S)
            assert np.max(np.abs(A - oA)) <= 1e-9 * max(1.0, np.abs(oA).max()) and np.max(np.abs(Bm - oB)) <= 1e-9 * max(1.0, np.abs(oB).max()), m
        ta = gp.discrete_linearize(x, u, S * 0.0 + np.eye(Nx) * 1e-12)
        gp.set_method('TA')
        assert np.allclose(ta[0], gp.discrete_linearize(x, u, S)[0], rtol=1e-6, atol=1e-8)      # EM(Sigma -> 0) = GP mean Jacobian
    Xraw = p['X'] * meta['stdZ'] + meta['meanZ']
    Yraw = p['Y'] * meta['stdY'] + meta['meanY']
    smse, mnlp = gp.validate(Xraw[:50], Yraw[:50], verbose=False)
    osmse, omnlp = og.validate(Xraw[:50], Yraw[:50])
    assert np.allclose(smse, osmse, rtol=1e-9, atol=0) and np.allclose(mnlp, omnlp, rtol=1e-9, atol=1e-12)
    gp.close()


def check_gp_rollout_lockstep(lib, N=200, Ny=2, Nu=1, T=5, seed=37, normalize=True):
    """`GP.rollout` through gpmpc_rollout_multi (what it does from GP.ROLLOUT_MULTI_MIN_N training points on) against its
    per-method route (gp_class.py:777-804: one method after the other): the moment methods bitwise, 'ME' / 'TA' to rounding,
    un-standardisation and clipping included."""
    from gp_mpc_amd.gp import GP
    d = Ny + Nu
    p = go.synthetic_problem(N, d, Ny, 4, seed=seed, sn=0.1)
    rng = np.random.default_rng(seed)
    Xraw = p['X'] * np.array([2.0, 0.5, 3.0][:d]) + np.array([0.5, -1.0, 0.2][:d])
    Yraw = p['Y'] * np.array([1.5, 0.7][:Ny]) + np.array([0.3, -0.2][:Ny])
    f = go.fit(p['X'], p['Y'], p['hyper'])
    hyper = dict(hyper=p['hyper'], chol=f['chol'], alpha=f['alpha'], invK=f['invK'])
    kw = dict(normalize=False, lib=lib)
    Xin, Yin = p['X'], p['Y']
    if normalize:
        meta = dict(meanY=Yraw.mean(0), stdY=Yraw.std(0), meanZ=Xraw.mean(0), stdZ=Xraw.std(0), meanX=Xraw[:, :Ny].mean(0),
                    stdX=Xraw[:, :Ny].std(0), meanU=Xraw[:, Ny:].mean(0), stdU=Xraw[:, Ny:].std(0))
        kw = dict(normalize=True, lib=lib, meta=meta, xlb=[-9.0] * Ny, xub=[9.0] * Ny, ulb=[-9.0] * Nu, uub=[9.0] * Nu)
    gp = GP(Xin, Yin, hyper=hyper, gp_method='TA', **kw)
    x0 = (Xraw if normalize else p['X'])[3, :Ny]
    U = 0.2 * rng.standard_normal((T, Nu)) + (Xraw[:, Ny:].mean(0) if normalize else 0.0)
    methods = ['EM', 'TA', 'ME', 'old_ME']
    keep = GP.ROLLOUT_MULTI_MIN_N
    try:
        GP.ROLLOUT_MULTI_MIN_N = 10 ** 9
        m1, v1 = gp.rollout(x0, U, methods=methods)
        GP.ROLLOUT_MULTI_MIN_N = 0
        m2, v2 = gp.rollout(x0, U, methods=methods)
    finally:
        GP.ROLLOUT_MULTI_MIN_N = keep
    for i, m in enumerate(methods):
        if m in ('EM', 'old_ME'):
            assert np.array_equal(m1[i], m2[i]) and np.array_equal(v1[i], v2[i]), m
        else:
            assert np.allclose(m1[i], m2[i], rtol=1e-10, atol=1e-12) and np.allclose(v1[i], v2[i], rtol=1e-8, atol=1e-12 * np.abs(v1[i]).max()), m
    gp.close()


def check_wide_inputs(lib, N=130, Ny=2, B=9, seed=77):
    """Input dimensions 9 .. 16: every path is instantiated up to d = 16 -- fit, mean / var / Jacobian, TA covariance,
    second-order outputs, the legacy methods, NLL + gradient, and the exact moments (gp_exact_moment,
    gp_functions.py:344-418, is dimension-generic: a second instantiation of the pair kernels with a 16-deep cross
    term; its derivative outputs likewise: check_em_sens is run at these dimensions by the callers)."""
    for d in (9, 12, 16):
        p = go.synthetic_problem(N, d, Ny, B, seed=seed + d, sn=0.1)
        X, Y, Z, S = p['X'], p['Y'], p['Z'], p['Sigma']
        H = p['hyper'].copy()
        H[:, :d] *= np.linspace(1.5, 3.0, d)[None, :]          # d is large: keep the kernel from going diagonal
        h = Handle(lib, X, Y)
        assert np.all(h.fit(H, want_invK=True) == 0)
        o = go.fit(X, Y, H)
        f = h.get_factors(invK=True)
        sf2, ell_min = H[:, d] ** 2, H[:, :d].min(axis=1)
        for a in range(Ny):
            assert relF(f['chol'][a], o['chol'][a]) <= 1e-11, d
        mean, cov, J = h.predict_jac('TA', Z, S)
        om, ov, oJ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'])
        ms = mean_scale(X, Z, H, o['alpha'])
        assert np.max(np.abs(mean - om) / ms) <= 1e-10 and np.max(np.abs(J - oJ) / (ms / ell_min)[..., None]) <= 1e-10, d
        assert np.max(np.abs(cov - go.ta_cov(ov, oJ, S))) <= 1e-10 * max(1.0, sf2.max()), d
        m2, v2, J2, Hm, dvar = h.predict_sens(Z)
        oH, odv = go.mean_var_sens(Z, X, H, o['alpha'], o['chol'])
        assert np.max(np.abs(Hm - oH) / (ms / ell_min ** 2)[..., None, None]) <= 1e-10, d
        assert np.max(np.abs(dvar - odv) / (sf2 / ell_min)[None, :, None]) <= 1e-10, d
        for method in ('old_ME', 'old_TA'):
            mm, cc = h.predict(method, Z, S)
            for b in range(B):
                rm, rc = go.old_ta(f['invK'], X, Y, H, Z[b], S[b]) if method == 'old_TA' else go.old_me(f['invK'], X, Y, H, Z[b])
                assert np.allclose(mm[b], rm, rtol=0, atol=1e-9 * max(1.0, np.abs(rm).max())), (d, method)
                assert np.allclose(cc[b], rc, rtol=0, atol=1e-9 * max(1.0, sf2.max())), (d, method)
        for a in range(Ny):
            v, g = h.nll(a, H[a], want_grad=True)
            ref = go.nll(H[a], X, Y[:, a])
            assert abs(v - ref) / (abs(ref) + N) <= 1e-10, d
            assert np.allclose(g, go.nll_grad(H[a], X, Y[:, a])[1], rtol=1e-8, atol=1e-8 * (abs(ref) + N)), d
        me, ce = h.predict('EM', Z[:3], S[:3])
        for b in range(3):
            em, ec = go.exact_moment(f['invK'], X, Y, H, Z[b], S[b])
            sc = _em_scale(f['invK'], X, Y, H, Z[b], S[b]).max() + sf2.max()
            assert np.allclose(me[b], em, rtol=0, atol=1e-9 * max(1.0, np.abs(em).max())), (d, np.abs(me[b] - em).max())
            assert np.max(np.abs(ce[b] - ec)) <= 1e-9 * sc, (d, np.max(np.abs(ce[b] - ec)), sc)
            assert np.allclose(ce[b], ce[b].T)
        h.close()


def check_random_shapes(lib, n_cases=10, seed=2024, nmax=220):
    """Seeded sweep over ragged shapes (N not a multiple of anything, d = 1..8, Ny = 1..3, B = 1..70): fit, mean / var /
    Jacobian, TA and EM covariances, NLL + gradient against the oracle -- the indexing of every kernel with padding in
    all of its dimensions at once."""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        N = int(rng.integers(1, nmax))
        d = int(rng.integers(1, 9))
        Ny = int(rng.integers(1, 4))
        B = int(rng.integers(1, 71))
        p = go.synthetic_problem(N, d, Ny, B, seed=int(rng.integers(1 << 30)), sn=0.1)
        X, Y, Z, S = p['X'], p['Y'], p['Z'], p['Sigma'] * 30
        H = np.hstack([rng.uniform(0.6, 2.5, (Ny, d)), rng.uniform(0.7, 1.6, (Ny, 1)), np.full((Ny, 1), 0.1)])
        h = Handle(lib, X, Y)
        assert np.all(h.fit(H, want_invK=True) == 0), (case, N, d, Ny)
        o = go.fit(X, Y, H)
        f = h.get_factors(invK=True)
        tag = (case, N, d, Ny, B)
        for a in range(Ny):
            assert relF(f['chol'][a], o['chol'][a]) <= 1e-11, tag
            assert relF(f['invK'][a], o['invK'][a]) <= 1e-9, tag
        mean, cov, J = h.predict_jac('TA', Z, S)
        om, ov, oJ = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'])
        ms = mean_scale(X, Z, H, o['alpha'])
        assert np.max(np.abs(mean - om) / ms) <= 1e-10, tag
        assert np.max(np.abs(J - oJ) / (ms / H[:, :d].min(axis=1))[..., None]) <= 1e-10, tag
        assert np.max(np.abs(cov - go.ta_cov(ov, oJ, S))) <= 1e-10 * max(1.0, (H[:, d] ** 2).max()), tag
        nb = min(B, 2)
        me, ce = h.predict('EM', Z[:nb], S[:nb])
        for b in range(nb):
            em, ec = go.exact_moment(f['invK'], X, Y, H, Z[b], S[b])
            sc = _em_scale(f['invK'], X, Y, H, Z[b], S[b]).max() + (H[:, d] ** 2).max()
            assert np.max(np.abs(me[b] - em)) <= 1e-9 * max(1.0, np.abs(em).max()), tag
            assert np.max(np.abs(ce[b] - ec)) <= 1e-9 * sc, tag
        a = int(rng.integers(0, Ny))
        v, g = h.nll(a, H[a], want_grad=True)
        ovv, og = go.nll_grad(H[a], X, Y[:, a])
        assert abs(v - ovv) <= 1e-10 * (abs(ovv) + N), tag
        assert np.max(np.abs(g - og)) <= 1e-7 * (np.abs(og).max() + 1e-3), tag
        h.close()
