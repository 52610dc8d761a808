"""CPU tier: pin the oracle (oracle/gp_oracle.py) to the reference.

Golden vectors in tests/golden/ were produced by oracle/make_golden.py, which
RUNS the reference's numpy code (calc_cov_matrix, calc_NLL_numpy, GP.covar,
GP.covSEard, train_gp_numpy) and reads its two saved models.  Functions that
only exist as CasADi graphs in the reference (a9-a12) are pinned through
identities."""
import numpy as np
import pytest

import gp_oracle as go


def relF(A, B):
    return np.linalg.norm(A - B) / np.linalg.norm(B)


@pytest.mark.parametrize('name', ['tank', 'car'])
def test_kernel_matrix_matches_reference(name, request):
    g = request.getfixturevalue(name)
    d = g['X'].shape[1]
    for a in range(g['hyper'].shape[0]):
        K = go.cov_se_ard(g['X'], g['X'], g['hyper'][a, :d], g['hyper'][a, d] ** 2)
        assert np.max(np.abs(K - g['ref_K'][a])) <= 1e-15 * g['hyper'][a, d] ** 2   # a1, same op order
        ks = go.cov_se_ard(g['X'], g['Z'], g['hyper'][a, :d], g['hyper'][a, d] ** 2)
        assert np.max(np.abs(ks - g['ref_ks'][a])) <= 1e-15 * g['hyper'][a, d] ** 2
        # a2 direct-difference form agrees with the expanded form to rounding
        kd = go.cov_se_ard_direct(g['X'], g['Z'], g['hyper'][a, :d], g['hyper'][a, d] ** 2)
        assert np.max(np.abs(kd - ks)) <= 1e-11 * g['hyper'][a, d] ** 2


@pytest.mark.parametrize('name,tol_alpha', [('tank', 1e-6), ('car', 1e-4)])
def test_refit_matches_saved_model(name, tol_alpha, request):
    """Stored chol/alpha/invK (written by gp_class.py:693-726) are re-derived
    from stored X, hyper.  L: 1e-10 rel-F.  alpha/invK are cond-limited
    (cond(K) up to 7e10) -> residual test."""
    g = request.getfixturevalue(name)
    f = go.fit(g['X'], g['Y'], g['hyper'])
    for a in range(g['hyper'].shape[0]):
        assert relF(f['chol'][a], g['chol'][a]) <= 1e-10
        assert np.all(np.triu(f['chol'][a], 1) == 0.0)
        assert relF(f['alpha'][a], g['alpha'][a]) <= tol_alpha
        d = g['X'].shape[1]
        K = go.gram(g['X'], g['hyper'][a, :d], g['hyper'][a, d] ** 2, g['hyper'][a, d + 1] ** 2)
        res = np.linalg.norm(K @ f['alpha'][a] - g['Y'][:, a]) / (np.linalg.norm(K) * np.linalg.norm(f['alpha'][a]))
        assert res <= 1e-14
    assert np.all(f['info'] == 0)
    # derived hyper fields, incl. the off-by-one `mean` slice gp_class.py:139-142
    d = g['X'].shape[1]
    assert np.allclose(g['length_scale'], g['hyper'][:, :d], rtol=0, atol=0)
    assert np.allclose(g['signal_var'], g['hyper'][:, d] ** 2, rtol=1e-15)
    assert np.allclose(g['noise_var'], g['hyper'][:, d + 1] ** 2, rtol=1e-15)
    assert np.allclose(g['hyper_mean'], g['hyper'][:, d + 1:], rtol=0, atol=0)


@pytest.mark.parametrize('name', ['tank', 'car'])
def test_nll_matches_reference(name, request):
    g = request.getfixturevalue(name)
    N = g['X'].shape[0]
    for a in range(g['hyper'].shape[0]):
        v = go.nll(g['hyper'][a], g['X'], g['Y'][:, a])
        assert abs(v - g['ref_nll'][a]) / (abs(g['ref_nll'][a]) + N) <= 1e-12


@pytest.mark.parametrize('name', ['tank', 'car'])
def test_variance_matches_reference_covar(name, request):
    """a14 `GP.covar` is the only numeric variance code in the reference."""
    g = request.getfixturevalue(name)
    gp = go.OracleGP(g['X'], g['Y'], g['hyper'], g['chol'], g['alpha'], g['invK'])
    cv = gp.covar(g['Z'])[:gp.Ny]
    d = gp.Nx
    for a in range(gp.Ny):
        sf2 = g['hyper'][a, d] ** 2
        assert np.max(np.abs(cv[a] - g['ref_covar'][a])) <= 1e-12 * sf2
    # a9 restatement (direct-form ks, triangular solve) == diag(covar)
    mean, var, _ = go.mean_var_jac(g['Z'], g['X'], g['hyper'], g['alpha'], g['chol'])
    for a in range(gp.Ny):
        sf2 = g['hyper'][a, d] ** 2
        assert np.max(np.abs(var[:, a] - g['ref_covar_diag'][a])) <= 1e-10 * sf2
        ks = go.cov_se_ard(g['X'], g['Z'], g['hyper'][a, :d], sf2)
        ref_mean = ks.T @ g['alpha'][a]
        scale = np.abs(ks).T @ np.abs(g['alpha'][a])
        assert np.max(np.abs(mean[:, a] - ref_mean) / scale) <= 1e-10


def test_jitter_rule(train_small):
    t = train_small
    for i, h in enumerate(t['probes']):
        for a in range(t['Y'].shape[1]):
            d = t['X'].shape[1]
            K = go.gram(t['X'], h[:d], h[d] ** 2, h[d + 1] ** 2)
            _, info = go.chol_jitter(K)
            assert info == t['probe_jitter'][i, a]
            v = go.nll(h, t['X'], t['Y'][:, a])
            assert abs(v - t['probe_nll'][i, a]) <= 1e-9 * (abs(t['probe_nll'][i, a]) + 40)


def test_train_reproduces_reference_optimum(train_small):
    """a8: same SLSQP call as optimize.py:466-467 on the restated NLL."""
    t = train_small
    opt = go.train(t['X'], t['Y'], multistart=1)
    assert np.allclose(opt['hyper'], t['hyper'], rtol=1e-5, atol=1e-9)
    for a in range(2):
        assert relF(opt['chol'][a], t['chol'][a]) <= 1e-4
    f = go.fit(t['X'], t['Y'], t['hyper'])
    for a in range(2):
        assert relF(f['chol'][a], t['chol'][a]) <= 1e-10
        assert relF(f['invK'][a], t['invK'][a]) <= 1e-5
        assert abs(go.nll(t['hyper'][a], t['X'], t['Y'][:, a]) - t['nll'][a]) <= 1e-9


def test_train_reproduces_reference_optimum_with_an_active_bound(train_small2):
    """a8 once more, on the second fixture the reference's own train_gp_numpy produced (oracle/make_golden.py::synthetic2):
    three inputs of different scale and a noise level above the reference's upper bound on sn, so that the optimum sits ON
    the bound (sn = 1e-2, optimize.py:441-442) -- the restated bounds and SLSQP call must land there too."""
    t = train_small2
    X, Y, d = t['X'], t['Y'], t['X'].shape[1]
    assert abs(t['hyper'][0, d + 1] - 1e-2) <= 1e-7          # SLSQP stops a hair inside the bound
    opt = go.train(X, Y, multistart=1)
    assert np.allclose(opt['hyper'], t['hyper'], rtol=1e-4, atol=1e-9), (opt['hyper'], t['hyper'])
    assert relF(opt['chol'][0], t['chol'][0]) <= 1e-3
    f = go.fit(X, Y, t['hyper'])
    assert relF(f['chol'][0], t['chol'][0]) <= 1e-10 and relF(f['invK'][0], t['invK'][0]) <= 1e-6
    assert relF(f['alpha'][0], t['alpha'][0]) <= 1e-7
    assert abs(go.nll(t['hyper'][0], X, Y[:, 0]) - t['nll'][0]) <= 1e-9 * (abs(t['nll'][0]) + len(X))
    K = go.cov_se_ard(X, X, t['hyper'][0, :d], t['hyper'][0, d] ** 2)
    assert np.max(np.abs(K - t['ref_K'][0])) <= 1e-14 * t['hyper'][0, d] ** 2
    _, var, _ = go.mean_var_jac(t['Z'], X, t['hyper'], t['alpha'], t['chol'], False)
    assert np.max(np.abs(var[:, 0] - np.diag(t['ref_covar'][0]))) <= 1e-10 * t['hyper'][0, d] ** 2


def test_train_reproduces_reference_when_its_search_fails(train_small3):
    """a8, third fixture (oracle/make_golden.py::synthetic3): the reference's SLSQP run ends with sn on its LOWER bound for
    the first output and does not leave the starting point for the other two (NLL 8e5 and 3e7 there).  The restated
    training must do exactly the same -- same bounds, same start, same finite-difference SLSQP call -- and the factors at
    whatever it returns must match."""
    t = train_small3
    X, Y = t['X'], t['Y']
    opt = go.train(X, Y, multistart=1)
    assert np.allclose(opt['hyper'], t['hyper'], rtol=1e-6, atol=1e-12), (opt['hyper'], t['hyper'])
    assert np.array_equal(opt['hyper'][1:], t['hyper'][1:])              # the untouched starts, bit for bit
    for a in range(3):
        assert abs(go.nll(t['hyper'][a], X, Y[:, a]) - t['nll'][a]) <= 1e-9 * (abs(t['nll'][a]) + len(X))
    f = go.fit(X, Y, t['hyper'])
    for a in (1, 2):                                                     # (output 0 has sn = 1e-10: cond(K) ~ 1e12)
        assert relF(f['chol'][a], t['chol'][a]) <= 1e-9 and relF(f['alpha'][a], t['alpha'][a]) <= 1e-6


def test_nll_gradient_vs_finite_differences(tank):
    g = tank
    X, y = g['X'], g['Y'][:, 1]
    h = np.array([12.0, 25.0, 14.0, 18.0, 22.0, 27.0, 2.1, 0.05])
    v, grad = go.nll_grad(h, X, y)
    assert abs(v - go.nll(h, X, y)) <= 1e-9 * (abs(v) + 60)
    for i in range(len(h)):
        e = np.zeros_like(h)
        e[i] = 1e-5 * max(1.0, abs(h[i]))
        fd = (go.nll(h + e, X, y) - go.nll(h - e, X, y)) / (2 * e[i])
        assert abs(fd - grad[i]) <= 1e-5 * (abs(grad[i]) + 1e-3), (i, fd, grad[i])


def test_mean_jacobian_vs_finite_differences(tank):
    g = tank
    z = g['Z'][:3]
    _, _, J = go.mean_var_jac(z, g['X'], g['hyper'], g['alpha'], g['chol'])
    eps = 1e-6
    for dd in range(z.shape[1]):
        e = np.zeros(z.shape[1])
        e[dd] = eps
        mp, _, _ = go.mean_var_jac(z + e, g['X'], g['hyper'], g['alpha'], g['chol'], False)
        mm, _, _ = go.mean_var_jac(z - e, g['X'], g['hyper'], g['alpha'], g['chol'], False)
        assert np.allclose((mp - mm) / (2 * eps), J[:, :, dd], rtol=1e-4, atol=1e-5)


def _well_conditioned(seed=3, N=50, d=3, Ny=2):
    p = go.synthetic_problem(N, d, Ny, 4, seed=seed, sn=0.1)
    p['hyper'][:, :d] = [[1.2, 0.9, 1.5], [0.8, 1.4, 1.1]]
    f = go.fit(p['X'], p['Y'], p['hyper'])
    return p, f


def test_exact_moment_identities():
    """a11 pins: beta==alpha; EM(Sigma->0) == ME; Monte-Carlo
    moments of the ME predictor under z~N(mu,Sigma) match EM."""
    p, f = _well_conditioned()
    X, Y, H = p['X'], p['Y'], p['hyper']
    d = X.shape[1]
    mu = np.array([0.3, -0.2, 0.5])
    for a in range(2):
        assert np.allclose(f['invK'][a] @ Y[:, a], f['alpha'][a], rtol=1e-9, atol=1e-11)
    m0, c0 = go.exact_moment(f['invK'], X, Y, H, mu, np.eye(d) * 1e-12)
    mean, var, _ = go.mean_var_jac(mu, X, H, f['alpha'], f['chol'], False)
    assert np.allclose(m0, mean[0], rtol=1e-8)
    assert np.allclose(np.diag(c0), var[0], rtol=1e-5, atol=1e-9)
    A = np.array([[0.3, 0.0, 0.0], [0.1, 0.2, 0.0], [-0.05, 0.1, 0.25]])
    Sigma = A @ A.T
    m, c = go.exact_moment(f['invK'], X, Y, H, mu, Sigma)
    rng = np.random.default_rng(0)
    zs = mu + rng.standard_normal((40000, d)) @ A.T
    ms, vs, _ = go.mean_var_jac(zs, X, H, f['alpha'], f['chol'], False)
    mc_mean = ms.mean(0)
    mc_cov = np.cov(ms.T) + np.diag(vs.mean(0))
    assert np.allclose(m, mc_mean, atol=4 * ms.std(0).max() / np.sqrt(len(zs)) + 1e-3)
    assert np.allclose(c, mc_cov, atol=0.02 * np.abs(mc_cov).max() + 1e-3)
    assert np.allclose(c, c.T, rtol=0, atol=0)
    # TA is a first-order approximation of the same thing
    _, var1, J = go.mean_var_jac(mu, X, H, f['alpha'], f['chol'])
    ta = go.ta_cov(var1, J, (Sigma * 1e-4)[None])[0]
    _, em = go.exact_moment(f['invK'], X, Y, H, mu, Sigma * 1e-4)
    assert np.allclose(ta, em, rtol=5e-2, atol=1e-7)


def test_legacy_methods_agree_with_me_on_well_conditioned_data():
    p, f = _well_conditioned()
    X, Y, H = p['X'], p['Y'], p['hyper']
    z = np.array([0.1, 0.4, -0.3])
    mean, var, _ = go.mean_var_jac(z, X, H, f['alpha'], f['chol'], False)
    m1, c1 = go.old_me(f['invK'], X, Y, H, z)
    assert np.allclose(m1, mean[0], rtol=1e-9)
    assert np.allclose(np.diag(c1), var[0], rtol=1e-7, atol=1e-10)
    m2, c2 = go.old_ta(f['invK'], X, Y, H, z, np.zeros((3, 3)))
    assert np.allclose(m2, mean[0], rtol=1e-9)
    assert np.allclose(np.diag(c2), var[0], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize('name', ['tank', 'car'])
def test_old_me_matches_reference_made_pin(name, request, old_me_pins):
    """a12 'old_ME' pinned by reference outputs: oracle `old_me` (restating gp_functions.py:176-256) against the reference's
    own GP.covSEard composed with its stored K^-1 and Y (oracle/make_golden.py legacy_pin), on the rounding scale of the sums."""
    g, pin = request.getfixturevalue(name), old_me_pins[name]
    assert np.array_equal(pin['Z'], g['Z'])
    for b, z in enumerate(pin['Z']):
        m, c = go.old_me(g['invK'], g['X'], g['Y'], g['hyper'], z)
        assert np.max(np.abs(m - pin['ref_old_me_mean'][:, b]) / pin['mean_scale'][:, b]) <= 1e-14
        assert np.max(np.abs(np.diag(c) - pin['ref_old_me_var'][:, b]) / pin['var_scale'][:, b]) <= 1e-14
        assert np.array_equal(c, np.diag(np.diag(c)))
    # the pin is not vacuous: the sums cancel to variances far below sf^2 (on the car model, cond(K) ~ 7e10, to values of either
    # sign -- what the reference's formulation gives there, kept as is)
    assert np.all(np.abs(pin['ref_old_me_var']) < 1e-3 * g['hyper'][:, g['X'].shape[1]][:, None] ** 2)


def test_predict_standardisation_and_rollout(tank):
    g = tank
    gp = go.OracleGP(g['X'], g['Y'], g['hyper'], g['chol'], g['alpha'], g['invK'],
                     normalize=True, meta=g['meta'], gp_method='ME')
    x = g['meta']['meanX'] + 0.1 * g['meta']['stdX']
    u = g['meta']['meanU'] - 0.2 * g['meta']['stdU']
    mean, cov = gp.predict(x, u, np.eye(6) * 1e-6)
    assert mean.shape == (4, 1) and cov.shape == (4, 4)
    zs = np.concatenate([(x - g['meta']['meanX']) / g['meta']['stdX'],
                         (u - g['meta']['meanU']) / g['meta']['stdU']])
    m, v, _ = go.mean_var_jac(zs, g['X'], g['hyper'], g['alpha'], g['chol'], False)
    assert np.allclose(mean[:, 0], m[0] * g['meta']['stdY'] + g['meta']['meanY'])
    assert np.allclose(np.diag(cov), v[0])        # covariance stays standardised (gp_class.py:262)
    with pytest.raises(NameError):
        gp.set_method('XX')
    U = np.tile(u, (5, 1))
    mr, vr = gp.rollout(x, U, methods=('EM', 'TA', 'ME'))
    assert mr.shape == (3, 6, 4) and np.all(np.isfinite(mr)) and np.all(np.isfinite(vr))
    # first step: the three methods agree closely because the input covariance is tiny
    assert np.allclose(mr[0, 1], mr[2, 1], rtol=1e-3, atol=1e-3)
    A, Bm = gp.discrete_linearize(x, u, np.eye(6) * 1e-6)
    assert A.shape == (4, 4) and Bm.shape == (4, 2)


def test_sensitivities_match_finite_differences(tank):
    """mean_var_sens / ta_cov_sens (closed forms behind gpmpc_predict_sens; no reference function --
    CasADi AD does this inside IPOPT) against central differences of mean_var_jac / ta_cov."""
    X, Y, H = tank['X'], tank['Y'], tank['hyper']
    d = X.shape[1]
    o = go.fit(X, Y, H, want_invK=False)
    Z = tank['Z'][:3]
    mean, var, J = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'])
    Hm, dvar = go.mean_var_sens(Z, X, H, o['alpha'], o['chol'])
    eps = 1e-4            # alpha ~ 1e3 on this model: smaller steps drown in cancellation noise
    rng = np.random.default_rng(0)
    A = rng.standard_normal((d, d))
    S = A @ A.T * 1e-3 + 1e-6 * np.eye(d)
    cov0 = go.ta_cov(var, J, np.tile(S, (len(Z), 1, 1)))
    for p in range(d):
        dz = np.zeros(d); dz[p] = eps
        mp, vp, Jp = go.mean_var_jac(Z + dz, X, H, o['alpha'], o['chol'])
        mm, vm, Jm = go.mean_var_jac(Z - dz, X, H, o['alpha'], o['chol'])
        assert np.allclose((Jp - Jm) / (2 * eps), Hm[..., p], rtol=1e-6, atol=1e-7 * np.abs(Hm).max())
        assert np.allclose((vp - vm) / (2 * eps), dvar[..., p], rtol=1e-5, atol=1e-5 * np.abs(dvar).max())
        cp = go.ta_cov(vp, Jp, np.tile(S, (len(Z), 1, 1)))
        cm = go.ta_cov(vm, Jm, np.tile(S, (len(Z), 1, 1)))
        for b in range(len(Z)):
            dcz, dcS = go.ta_cov_sens(var[b], J[b], Hm[b], dvar[b], S)
            assert np.allclose((cp[b] - cm[b]) / (2 * eps), dcz[..., p], rtol=1e-5, atol=1e-5 * np.abs(dcz).max())
    # cov is linear in Sigma: d cov / d Sigma_de = J[:, d] J[:, e]^T exactly
    dcz, dcS = go.ta_cov_sens(var[0], J[0], Hm[0], dvar[0], S)
    E = np.zeros((d, d)); E[1, 2] = 1.0
    c1 = go.ta_cov(var[:1], J[:1], (S + E)[None])[0]
    assert np.allclose(c1 - cov0[0], dcS[:, :, 1, 2], rtol=1e-12, atol=1e-14)


# ---------------------------------------------------------------------------------------------
# Stronger pins for the CasADi-only functions (a9-a12).  Everything below is built on functions that
# ARE pinned to the reference's numpy code: `cov_se_ard` (bit-exact vs GP.covSEard / calc_cov_matrix)
# and the Cholesky-based variance of GP.covar.
# ---------------------------------------------------------------------------------------------
def _pinned_mean_var(Zq, X, H, alpha, chol):
    """mean_a(z) = covSEard(X, z)^T alpha_a and var_a(z) = sf2 - |L^-1 ks|^2 exactly as the reference's numeric
    code evaluates them (gp_class.py:314-350 kernel, :377-380 variance), for complex or real z."""
    from scipy.linalg import solve_triangular
    d = X.shape[1]
    Ny = H.shape[0]
    mean = np.zeros((len(Zq), Ny), dtype=Zq.dtype)
    var = np.zeros((len(Zq), Ny), dtype=Zq.dtype)
    for a in range(Ny):
        sf2 = H[a, d] ** 2
        dist = 0
        for i in range(d):                      # calc_cov_matrix / covSEard operation order, complex-safe
            x1 = X[:, i].reshape(-1, 1)
            x2 = Zq[:, i].reshape(-1, 1)
            dist = ((x1 ** 2).sum(1).reshape(-1, 1) + (x2 ** 2).sum(1) - 2 * (x1 @ x2.T)) / H[a, i] ** 2 + dist
        ks = sf2 * np.exp(-.5 * dist)
        mean[:, a] = ks.T @ alpha[a]
        v = solve_triangular(chol[a].astype(Zq.dtype), ks, lower=True)
        var[:, a] = sf2 - np.sum(v * v, axis=0)
    return mean, var


def test_mean_jacobian_and_sensitivities_by_complex_step(tank):
    """a9's J (CasADi AD in the reference, gp_functions.py:146-147) and the second-order closed forms: the pinned
    mean / variance are analytic in z, so Im f(z + i h e_p) / h is their derivative to rounding -- a pin at 1e-12
    instead of the 1e-4 of central differences."""
    g = tank
    X, H = g['X'], g['hyper']
    d = X.shape[1]
    o = go.fit(X, g['Y'], H, want_invK=False)
    Z = g['Z'][:4]
    mean, var, J = go.mean_var_jac(Z, X, H, o['alpha'], o['chol'])
    Hm, dvar = go.mean_var_sens(Z, X, H, o['alpha'], o['chol'])
    h = 1e-30
    ms = np.abs(go.cov_se_ard_direct(X, Z, H[0, :d], 1.0)).T @ np.abs(o['alpha'][0])     # size of the sum's terms
    for p in range(d):
        Zc = Z.astype(complex)
        Zc[:, p] += 1j * h
        mc, vc = _pinned_mean_var(Zc, X, H, o['alpha'], o['chol'])
        assert np.max(np.abs(mc.real - mean)) <= 1e-9 * ms.max()
        assert np.max(np.abs(mc.imag / h - J[:, :, p])) <= 1e-11 * np.abs(J).max() + 1e-12 * ms.max()
        assert np.max(np.abs(vc.imag / h - dvar[:, :, p])) <= 1e-9 * np.abs(dvar).max()
    # Hessian of the mean: complex step through the analytic Jacobian (same formula, complex arithmetic)
    for p in range(d):
        for b in range(len(Z)):
            zc = Z[b].astype(complex)
            zc[p] += 1j * h
            for a in range(H.shape[0]):
                ell2 = H[a, :d] ** 2
                ks = H[a, d] ** 2 * np.exp(-.5 * np.sum((X - zc) ** 2 / ell2, axis=1))
                Jc = ((ks * o['alpha'][a])[:, None] * (X - zc) / ell2).sum(axis=0)
                assert np.max(np.abs(Jc.imag / h - Hm[b, a, :, p])) <= 1e-11 * np.abs(Hm[b, a]).max()


def _gauss_hermite_moments(mu, Sigma, X, H, alpha, chol, n=64):
    """E[mean(z)], Cov[mean(z)] + diag(E[var(z)]) under z ~ N(mu, Sigma) by tensor Gauss-Hermite quadrature of the
    PINNED predictor (d <= 2)."""
    d = len(mu)
    t, w = np.polynomial.hermite_e.hermegauss(n)        # weight exp(-t^2/2)
    w = w / np.sqrt(2 * np.pi)
    A = np.linalg.cholesky(Sigma)
    grids = np.meshgrid(*([t] * d), indexing='ij')
    T = np.stack([g.ravel() for g in grids], axis=1)
    W = np.ones(len(T))
    for k in range(d):
        W = W * w[np.stack([g.ravel() for g in np.meshgrid(*([np.arange(n)] * d), indexing='ij')], axis=1)[:, k]]
    Zq = mu + T @ A.T
    m, v = _pinned_mean_var(Zq, X, H, alpha, chol)
    Em = W @ m
    Ev = W @ v
    C = (m * W[:, None]).T @ m - np.outer(Em, Em) + np.diag(Ev)
    return Em, C


@pytest.mark.parametrize('d', [1, 2])
def test_exact_moment_vs_gauss_hermite_quadrature(d):
    """a11 `gp_exact_moment` (gp_functions.py:344-418): mean, variances AND the cross-covariance between outputs
    against quadrature of the pinned ME predictor, to 1e-9 (this replaces a 2 % Monte-Carlo check)."""
    Ny = 2
    p = go.synthetic_problem(40, d, Ny, 4, seed=11 + d, sn=0.1)
    X, Y, H = p['X'], p['Y'], p['hyper']
    H[:, :d] = [[1.3, 0.9][:d], [0.8, 1.6][:d]]
    H[:, d] = [1.2, 0.7]
    f = go.fit(X, Y, H)
    for (mu, Sigma) in ((np.array([0.3, -0.4][:d]), np.array([[0.09, 0.03], [0.03, 0.16]])[:d, :d]),
                        (np.array([-1.1, 0.6][:d]), np.array([[0.5, -0.2], [-0.2, 0.3]])[:d, :d])):
        em_mean, em_cov = go.exact_moment(f['invK'], X, Y, H, mu, Sigma)
        q_mean, q_cov = _gauss_hermite_moments(mu, Sigma, X, H, f['alpha'], f['chol'], n=80)
        q2_mean, q2_cov = _gauss_hermite_moments(mu, Sigma, X, H, f['alpha'], f['chol'], n=64)
        assert np.max(np.abs(q_mean - q2_mean)) <= 1e-12 and np.max(np.abs(q_cov - q2_cov)) <= 1e-11   # converged
        assert np.max(np.abs(em_mean - q_mean)) <= 1e-9 * max(1.0, np.abs(q_mean).max())
        assert np.max(np.abs(em_cov - q_cov)) <= 1e-9 * (H[:, d] ** 2).max(), (em_cov, q_cov)
        assert abs(em_cov[0, 1]) > 1e-4          # the cross-covariance is not trivially zero in this set-up


def _old_ta_scalar_loops(invK, X, Y, H, z, S):
    """`gp_taylor_approx(diag=True)` gp_functions.py:259-340 evaluated entry by entry with explicit loops and CasADi's
    indexing semantics spelt out (an evaluation independent of the vectorised `go.old_ta`):
      * `v[e]` on the N x Nx matrix v is linear (column-major) element e, i.e. v[e, 0]           (:327-328)
      * `var` has Nx entries and is filled output by output, so `var[d]` with d > a is still 0     (:283,:320,:331)
      * covar_temp keeps only its [0, 0] entry = inputcovar[a, a]                                   (:334-335)"""
    Ny = len(invK)
    N, Nx = X.shape
    v = X - z.reshape(1, Nx)
    var = [0.0] * Nx
    d_mean = [0.0] * Ny
    mean = [0.0] * Ny
    cov = np.zeros((Ny, Ny))
    dd = [[0.0] * Ny for _ in range(Ny)]
    for a in range(Ny):
        w = [1.0 / H[a, k] ** 2 for k in range(Nx)]
        sf2 = H[a, Nx] ** 2
        ks = [sf2 * np.exp(-0.5 * sum((X[i, k] - z[k]) ** 2 / H[a, k] ** 2 for k in range(Nx))) for i in range(N)]
        alpha = [sum(invK[a][i, j] * Y[j, a] for j in range(N)) for i in range(N)]
        invKks = [sum(invK[a][i, j] * ks[j] for j in range(N)) for i in range(N)]
        mean[a] = sum(ks[i] * alpha[i] for i in range(N))
        var[a] = sf2 - sum(ks[i] * invKks[i] for i in range(N))
        d_mean[a] = sum(w[a] * v[i, a] * ks[i] * alpha[i] for i in range(N))
        for d in range(Ny):
            for e in range(Ny):
                ve = v[e, 0]                      # v[e]: linear index
                vd = v[d, 0]                      # v[d]
                t1a = [sum(v[i, d] * ks[i] * invK[a][i, j] for i in range(N)) for j in range(N)]
                t1b = sum(t1a[j] * ve * ks[j] for j in range(N))
                t2 = sum(vd * ve * ks[i] * invKks[i] for i in range(N))
                dd[d][e] = -2 * w[d] * w[e] * (t1b + t2)
                if d == e:
                    dd[d][e] += 2 * w[d] * (sf2 - var[d])
        cov[a, a] = var[a] + S[a, a] * (0.5 * dd[0][0] + d_mean[0] * d_mean[0])    # trace(covar_temp @ (...))
    return np.array(mean), cov


def test_old_ta_with_input_covariance():
    """a12 'old_TA' with Sigma != 0: the whole dd_var block, which the Sigma = 0 test never reaches."""
    p = go.synthetic_problem(30, 4, 3, 3, seed=9, sn=0.1)
    X, Y, H = p['X'], p['Y'], p['hyper']
    H[:, :4] = [[1.2, 0.9, 1.5, 1.1], [0.8, 1.4, 1.1, 0.7], [1.0, 1.3, 0.6, 1.9]]
    f = go.fit(X, Y, H)
    from scipy.linalg import solve_triangular
    for b in range(3):
        z = p['Z'][b]
        S = p['Sigma'][b] * 200
        m, c = go.old_ta(f['invK'], X, Y, H, z, S)
        m2, c2 = _old_ta_scalar_loops(f['invK'], X, Y, H, z, S)
        assert np.allclose(m, m2, rtol=1e-12, atol=1e-13)
        assert np.allclose(c, c2, rtol=1e-10, atol=1e-13)
        assert np.all(np.abs(np.diag(c) - np.diag(go.old_me(f['invK'], X, Y, H, z)[1])) > 1e-8)   # the Sigma term is live
    # building blocks of :325-331 against derivatives of the pinned variance: -2 w_d (v_d * ks)^T K^-1 ks = d var_a/dz_d
    z = p['Z'][0]
    for a in range(3):
        ell2 = H[a, :4] ** 2
        ks = go.cov_se_ard_direct(X, z, H[a, :4], H[a, 4] ** 2)[:, 0]
        for dd_ in range(4):
            t1a = ((X[:, dd_] - z[dd_]) * ks) @ f['invK'][a]
            zc = z.astype(complex)
            zc[dd_] += 1e-30j
            kc = H[a, 4] ** 2 * np.exp(-.5 * np.sum((X - zc) ** 2 / ell2, axis=1))
            vc = solve_triangular(f['chol'][a].astype(complex), kc, lower=True)
            dvar = (-(vc * vc).sum()).imag / 1e-30
            assert abs(-2 / ell2[dd_] * (t1a @ ks) - dvar) <= 1e-8 * max(1.0, abs(dvar))


def test_em_rollout_is_stable_on_well_conditioned_model():
    """The EM roll-out of OracleGP (the comparison target of the device roll-out in the GPU tier) on a model where
    K^-1 cancellation does not dominate: repeated runs agree and the covariance stays symmetric PSD."""
    p = go.synthetic_problem(60, 3, 2, 2, seed=5, sn=0.1)
    gp = go.OracleGP(p['X'], p['Y'], p['hyper'], gp_method='EM')
    U = np.tile(np.array([[0.2]]), (6, 1))
    m1, v1 = gp.rollout(np.array([0.1, -0.2]), U, methods=('EM', 'TA'))
    assert np.all(np.isfinite(m1)) and np.all(v1 >= 0)
    assert np.allclose(m1[0, 1], m1[1, 1], rtol=0, atol=2e-2)     # EM ~ TA after one step (Sigma_x = sn^2 I = 0.01 I)


def _exact_moment_complex(invK, X, Y, H, mu, Sigma):
    """Transcription of go.exact_moment that is analytic in (mu, Sigma): determinants by np.linalg.det instead of
    |prod diag(R_qr)| (equal for the positive arguments used here).  Only for complex-step differentiation."""
    Ny, (N, Nx) = len(invK), X.shape
    lh = np.log(H)
    v = X - mu.reshape(1, Nx)
    eye = np.eye(Nx)
    mean = np.zeros(Ny, dtype=complex)
    beta = np.stack([invK[a] @ Y[:, a] for a in range(Ny)], axis=1)
    log_k = np.zeros((N, Ny), dtype=complex)
    cov = np.zeros((Ny, Ny), dtype=complex)
    for a in range(Ny):
        iL = np.diag(np.exp(-2 * lh[a, :Nx]))
        R = Sigma + np.diag(np.exp(2 * lh[a, :Nx]))
        iR = iL @ (eye - np.linalg.solve(eye + Sigma @ iL, Sigma @ iL))
        c = np.exp(2 * lh[a, Nx]) / np.sqrt(np.linalg.det(R)) * np.exp(np.sum(lh[a, :Nx]))
        mean[a] = np.sum(c * np.exp(-np.sum((v @ iR) * v, axis=1) * 0.5) * beta[:, a])
        log_k[:, a] = 2 * lh[a, Nx] - np.sum((v / np.exp(lh[a, :Nx])) ** 2, axis=1) * 0.5
    for a in range(Ny):
        ii = v / np.exp(2 * lh[a, :Nx])
        for b in range(a + 1):
            R = Sigma @ np.diag(np.exp(-2 * lh[a, :Nx]) + np.exp(-2 * lh[b, :Nx])) + eye
            t = 1.0 / np.sqrt(np.linalg.det(R))
            ij = v / np.exp(2 * lh[b, :Nx])
            S = np.linalg.solve(R, Sigma * 0.5)
            aQ, bQ = ii @ S, (-ij) @ S
            mh = np.sum(aQ * ii, axis=1)[:, None] + np.sum(bQ * (-ij), axis=1)[None, :] - 2 * aQ @ (-ij).T
            A = np.outer(beta[:, a], beta[:, b]) - (invK[a] if a == b else 0.0)
            cov[a, b] = cov[b, a] = t * np.sum(A * np.exp(log_k[:, a][:, None] + log_k[:, b][None, :] + mh))
        cov[a, a] += np.exp(2 * lh[a, Nx])
    return mean, cov - np.outer(mean, mean)


def test_exact_moment_sensitivities_by_complex_step():
    """exact_moment_sens (the derivative outputs of 'EM' for a casadi Callback) against complex-step derivatives of
    the exact-moment formulas, every entry of d/d mu and d/d Sigma (entries of Sigma independent)."""
    p, f = _well_conditioned(seed=8, N=40, d=3, Ny=2)
    X, Y, H = p['X'], p['Y'], p['hyper']
    d, Ny = 3, 2
    mu = np.array([0.2, -0.3, 0.4])
    A = np.array([[0.3, 0.0, 0.0], [0.1, 0.25, 0.0], [-0.05, 0.1, 0.2]])
    Sg = A @ A.T
    m0, c0 = go.exact_moment(f['invK'], X, Y, H, mu, Sg)
    mc, cc = _exact_moment_complex(f['invK'], X, Y, H, mu.astype(complex), Sg.astype(complex))
    assert np.max(np.abs(mc.real - m0)) <= 1e-13 and np.max(np.abs(cc.real - c0)) <= 1e-13   # same function
    dm_dz, dm_dS, dc_dz, dc_dS = go.exact_moment_sens(f['invK'], X, Y, H, mu, Sg)
    h = 1e-30
    for k in range(d):
        muc = mu.astype(complex)
        muc[k] += 1j * h
        mc, cc = _exact_moment_complex(f['invK'], X, Y, H, muc, Sg.astype(complex))
        assert np.max(np.abs(mc.imag / h - dm_dz[:, k])) <= 1e-11 * max(1.0, np.abs(dm_dz).max())
        assert np.max(np.abs(cc.imag / h - dc_dz[:, :, k])) <= 1e-11 * max(1.0, np.abs(dc_dz).max())
        for l in range(d):
            Sc = Sg.astype(complex)
            Sc[k, l] += 1j * h
            mc, cc = _exact_moment_complex(f['invK'], X, Y, H, mu.astype(complex), Sc)
            assert np.max(np.abs(mc.imag / h - dm_dS[:, k, l])) <= 1e-11 * max(1.0, np.abs(dm_dS).max()), (k, l)
            assert np.max(np.abs(cc.imag / h - dc_dS[:, :, k, l])) <= 1e-11 * max(1.0, np.abs(dc_dS).max()), (k, l)


# ---------------------------------------------------------------------------------------------
# r06: REFERENCE-RUN pins for the CasADi-only rows (oracle/make_golden.py ta_pin / em_pin / ref_written_model): the
# reference's own numpy GP.covSEard / GP.covar / train_gp_numpy / save_model produced every number compared with below.
# ---------------------------------------------------------------------------------------------
EM_PIN_TOL = {'train_small': (1e-7, 1e-5), 'em_model2': (1e-12, 1e-12)}     # (mean, cov) absolute; see em_pin's docstring


@pytest.mark.parametrize('name', ['tank', 'car'])
def test_mean_jacobian_and_ta_match_reference_run_pin(name, request, ta_pins):
    """a9 mean / J and a10 'TA' (gp_functions.py:114-147,152-173) against the reference-run composition: bars are the
    rounding scale of the sums (1e-14 of sum|ks alpha| resp. sum|ks alpha (x - z)| / l^2) and, for the quantity MPC consumes,
    |dcov| <= 1e-10 max|cov| on the tank model (cond 6e7); the car model (cond 7e10: alpha ~ 1e6) carries 1e-9."""
    g = request.getfixturevalue(name)
    t = ta_pins[name]
    m, v, J = go.mean_var_jac(t['Z'], g['X'], g['hyper'], g['alpha'], g['chol'])
    cov = go.ta_cov(v, J, t['Sigma'])
    d = g['X'].shape[1]
    assert np.max(np.abs(m - t['ref_mean']) / t['mean_scale']) <= 1e-14
    assert np.max(np.abs(J - t['ref_J']) / t['J_scale']) <= 1e-14
    assert np.max(np.abs(v - t['ref_var'])) <= 1e-10 * (g['hyper'][:, d] ** 2).max()
    tol = 1e-10 if name == 'tank' else 1e-9
    assert np.max(np.abs(cov - t['ref_ta_cov'])) <= tol * np.abs(t['ref_ta_cov']).max()


@pytest.mark.parametrize('name', ['train_small', 'em_model2'])
def test_exact_moment_matches_reference_run_quadrature(name, em_pins):
    """a11 'EM' (gp_functions.py:344-430) against Gauss-Hermite quadrature OF THE REFERENCE'S OWN numeric predictor on models
    the reference trained.  em_model2 (cond 4e3): 1e-12 absolute on mean and covariance (quadrature converged to 2e-15).
    train_small (cond 7e8): the closed form's own K^-1 arithmetic limits it to cond * eps * sf^2 ~ 1e-6."""
    model, pin = em_pins[name]
    tm, tc = EM_PIN_TOL[name]
    assert pin['quad_convergence'].max() <= 5e-12
    for i in range(len(pin['mu'])):
        m, c = go.exact_moment(model['invK'], model['X'], model['Y'], model['hyper'], pin['mu'][i], pin['Sigma'][i])
        assert np.max(np.abs(m - pin['ref_em_mean'][i])) <= tm, (i, m, pin['ref_em_mean'][i])
        assert np.max(np.abs(c - pin['ref_em_cov'][i])) <= tc, (i, c, pin['ref_em_cov'][i])
    assert np.abs(pin['ref_em_cov'][:, 0, 1]).max() > 1e-3          # cross-covariances are exercised


def test_reference_written_model_file(ref_written):
    """f2: the JSON the reference's GP.save_model wrote (gp_class.py:693-734): full key set, derived entries, and the oracle
    GP built from it reproduces the reference's GP.covar / covSEard^T alpha in the model's standardised coordinates."""
    import json
    path, out = ref_written
    d = json.load(open(path + '.json'))
    assert set(d) == {'X', 'Y', 'hyper', 'mean_func', 'normalize', 'xlb', 'xub', 'ulb', 'uub', 'meta'}
    assert set(d['hyper']) == {'hyper', 'invK', 'alpha', 'chol', 'length_scale', 'signal_var', 'noise_var', 'mean'}
    assert set(d['meta']) == {'meanY', 'stdY', 'meanZ', 'stdZ', 'meanX', 'stdX', 'meanU', 'stdU'}
    X, Y, H = np.array(d['X']), np.array(d['Y']), np.array(d['hyper']['hyper'])
    Nx = X.shape[1]
    assert np.allclose(X, (out['X_raw'] - d['meta']['meanZ']) / d['meta']['stdZ'], rtol=1e-15, atol=1e-15)
    assert np.array_equal(np.array(d['hyper']['signal_var']), H[:, Nx] ** 2)
    assert np.array_equal(np.array(d['hyper']['mean']), H[:, Nx + 1:])          # off by one, includes sn (gp_class.py:142)
    f = go.fit(X, Y, H)
    assert relF(f['chol'], np.array(d['hyper']['chol'])) <= 1e-12
    og = go.OracleGP(X, Y, H, np.array(d['hyper']['chol']), np.array(d['hyper']['alpha']), np.array(d['hyper']['invK']),
                     normalize=True, meta={k: np.array(v) for k, v in d['meta'].items()}, gp_method='ME')
    m, v, _ = go.mean_var_jac(out['Zs'], X, H, og.alpha, og.chol, want_jac=False)
    ms = np.stack([np.abs(go.cov_se_ard_direct(X, out['Zs'], H[a, :Nx], H[a, Nx] ** 2)).T @ np.abs(og.alpha[a])
                   for a in range(len(H))], axis=1)
    assert np.max(np.abs(m - out['ref_mean_std']) / ms) <= 1e-14
    assert np.max(np.abs(v - np.stack([np.diag(c) for c in out['ref_covar']], axis=1))) <= 1e-10 * (H[:, Nx] ** 2).max()


@pytest.mark.parametrize('name', ['train_small', 'em_model2'])
def test_exact_moment_closed_form_in_longdouble_matches_reference_run_quadrature(name, em_pins):
    """The same closed form (gp_functions.py:344-418) evaluated in longdouble (tests/parity_cases.exact_moment_longdouble: K^-1 y
    and trace(K^-1 Q) refined to extended precision) against the reference-run quadrature: on train_small (cond 7e8), where
    every fp64 evaluation sits 3e-6 away, it agrees to 1e-8 -- the 3e-6 is fp64 arithmetic on K^-1, not the formula."""
    import parity_cases as pc
    model, pin = em_pins[name]
    tm, tc = {'train_small': (1e-9, 1e-8), 'em_model2': (1e-12, 1e-12)}[name]
    for i in range(len(pin['mu'])):
        m, c = pc.exact_moment_longdouble(model['X'], model['Y'], model['hyper'], pin['mu'][i], pin['Sigma'][i])
        assert np.max(np.abs(m - pin['ref_em_mean'][i])) <= tm and np.max(np.abs(c - pin['ref_em_cov'][i])) <= tc, (i, m, c)


def test_mean_truth_helpers_on_a_small_problem():
    """longdouble_alpha / longdouble_mean (the yardstick of the GPU tier's mean-digits test) against the fp64 oracle where fp64 is
    accurate (cond 1e5): they agree to cond * eps, and the refinement has converged (a fifth step moves alpha by < 1 % of fp64's error)."""
    import parity_cases as pc
    p = go.synthetic_problem(300, 4, 1, 20, seed=5, sn=0.1)
    a4, K = pc.longdouble_alpha(p['X'], p['Y'][:, 0], p['hyper'][0], iters=4)
    a5, _ = pc.longdouble_alpha(p['X'], p['Y'][:, 0], p['hyper'][0], iters=5)
    o = go.fit(p['X'], p['Y'], p['hyper'], want_invK=False)
    e64 = np.max(np.abs(o['alpha'][0] - a4.astype(np.float64)))
    assert e64 <= 1e-10 * np.abs(o['alpha'][0]).max()
    assert np.max(np.abs(a4 - a5)) <= 1e-2 * e64            # the yardstick is >= 100 x finer than fp64's own error (11 more mantissa bits)
    t = pc.longdouble_mean(p['X'], p['hyper'][0], a4, p['Z']).astype(np.float64)
    om, _, _ = go.mean_var_jac(p['Z'], p['X'], p['hyper'], o['alpha'], o['chol'], False)
    assert np.max(np.abs(om[:, 0] - t)) <= 1e-11 * np.abs(t).max()
