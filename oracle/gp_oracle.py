"""CPU oracle for the GP-regression hot path of helgeanl/GP-MPC.

TEST INFRASTRUCTURE ONLY.  This module is a numpy (fp64) restatement of the
reference algorithm.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker --
never as the thing measured as "the product" and never as a fallback: the
shipped path (`gp_mpc_amd`) raises when its HIP library is missing.

Parity status
-------------
* PINNED against the reference's own numpy code (imported in the build
  container by `oracle/make_golden.py`, outputs frozen in `tests/golden/`):
  a1 `calc_cov_matrix` / `GP.covSEard`, a3-a5+a7 `calc_NLL_numpy`,
  a6/a8 `train_gp_numpy`, a14 `GP.covar`, and against the two saved models
  `examples/models/gp_{tank,car}_example.json` (chol / alpha / invK);
  (r05) a12 'old_ME' (`gp`, gp_functions.py:176-256, alpha=None): its two
  matrix products composed from the reference's own `GP.covSEard` output and
  the K^-1 / Y of its saved models (`make_golden.py legacy`,
  tests/golden/{tank,car}_old_me.npz).
* (r06) PINNED BY REFERENCE-RUN COMPOSITIONS (`make_golden.py ta | em | refmodel`; casadi is not installable, so the
  CasADi graphs themselves cannot run, but every number below was produced by the reference's own numpy functions):
  a9 mean and its Jacobian J: `GP.covSEard(X, z)^T alpha` (gp_class.py:314-350) and its complex-step derivative
  (h = 1e-30: exact to rounding) on the two saved models -> tests/golden/{tank,car}_ta.npz; oracle vs pin 2e-16 of the
  sums' rounding scale;  a10 'TA': diag(`GP.covar`) + J Sigma J^T (the one line of build_TA_cov, gp_functions.py:167-171)
  -> same files; oracle vs pin 7e-14 (tank) / 1.4e-10 (car, cond 7e10) of max|cov|;
  a11 'EM' (`gp_exact_moment`): tensor Gauss-Hermite quadrature (80^2 nodes, converged to 2e-15 / 2e-12) of that same
  reference predictor on two models the reference's `train_gp_numpy` produced -> tests/golden/{em_model2,train_small}_em.npz;
  oracle vs pin 7e-14 absolute on em_model2 (cond 4e3); on train_small (cond 7e8) 3e-6 = cond * eps * sf^2, the closed
  form's own K^-1 arithmetic (the quadrature is the more accurate side there);
  f2 model file: tests/golden/ref_written_model.json was written by the reference's `GP.optimize` + `GP.save_model`.
* STILL "PARITY UNPINNED": a12 `gp_taylor_approx` ('old_TA') only -- its graph relies on CasADi's linear indexing of an
  MX matrix (gp_functions.py:325-331, self-documented bug) which no numpy function of the reference reproduces; restated
  twice independently (vectorised here, scalar loops in tests/test_oracle.py) and the two agree.

All `file:line` citations are into /root/reference/gp_mpc/.
Conventions: hyper[a] = [ell_1..ell_d, sf, sn] with sf, sn
STANDARD DEVIATIONS; arrays are C-order fp64; chol[a] is dense lower
triangular with explicit zeros above the diagonal.
"""
from __future__ import annotations

import numpy as np

JITTER = 1e-8  # optimize.py:349 / gp_class.py:528


# --------------------------------------------------------------------------
# a1 / a2  SE-ARD kernel
# --------------------------------------------------------------------------
def cov_se_ard(X, Z, ell, sf2):
    """a1: `GP.covSEard` gp_class.py:314-350 and `calc_cov_matrix`
    optimize.py:303-319 -- the EXPANDED form x^2 + z^2 - 2 x z accumulated
    per input dimension, exactly in the reference's operation order."""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(1, -1)
    if Z.ndim == 1:
        Z = Z.reshape(1, -1)
    n1, D = X.shape
    n2, D2 = Z.shape
    if D != D2:  # gp_class.py:342-344
        raise ValueError('Input dimensions are not the same! D_x=' + str(D)
                         + ', D_z=' + str(D2))
    dist = 0
    for i in range(D):
        x1 = X[:, i].reshape(n1, 1)
        x2 = Z[:, i].reshape(n2, 1)
        dist = (np.sum(x1 ** 2, 1).reshape(-1, 1) + np.sum(x2 ** 2, 1)
                - 2 * np.dot(x1, x2.T)) / ell[i] ** 2 + dist
    return sf2 * np.exp(-.5 * dist)


def cov_se_ard_direct(X, Z, ell, sf2):
    """a2: CasADi `covSEard` gp_functions.py:17-22 -- DIRECT difference form
    sum_d (x_d - z_d)^2 / ell_d^2 (what `build_gp` :114-117 evaluates)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
    ell = np.asarray(ell, dtype=np.float64)
    diff = X[:, None, :] - Z[None, :, :]
    dist = np.sum(diff ** 2 / ell ** 2, axis=2)
    return sf2 * np.exp(-.5 * dist)


# --------------------------------------------------------------------------
# a3 / a4  noise, symmetrise, Cholesky with the one-shot jitter rule
# --------------------------------------------------------------------------
def gram(X, ell, sf2, sn2):
    """a1+a3: K = k(X,X) + sn^2 I, symmetrised.  optimize.py:341-344."""
    n = X.shape[0]
    K = cov_se_ard(X, X, ell, sf2)
    K = K + sn2 * np.eye(n)
    K = (K + K.T) * 0.5
    return K


def chol_jitter(K):
    """a4: optimize.py:345-350 (same rule at :483-488, gp_class.py:524-529).
    Returns (L, info): info = 0 plain success, 1 = jitter 1e-8*I was added
    once.  A second failure propagates as LinAlgError like the reference."""
    try:
        return np.linalg.cholesky(K), 0
    except np.linalg.LinAlgError:
        K = K + np.eye(K.shape[0]) * JITTER
        return np.linalg.cholesky(K), 1


# --------------------------------------------------------------------------
# a5 / a6  alpha and the explicit inverse
# --------------------------------------------------------------------------
def alpha_from_chol(L, y):
    """a5: alpha = solve(L^T, solve(L, y)) with GENERAL LU solves, as the
    reference does (optimize.py:353-354, :494)."""
    return np.linalg.solve(L.T, np.linalg.solve(L, y))


def inv_from_chol(L):
    """a6: invL = solve(L, I); invK = solve(L^T, invL). optimize.py:489-490."""
    n = L.shape[0]
    invL = np.linalg.solve(L, np.eye(n))
    return np.linalg.solve(L.T, invL)


def fit_output(X, y, hyper_row, want_invK=True):
    """a1,a3-a6 at a fixed hyper row (optimize.py:476-494 after arg-min;
    gp_class.py:514-537 in update_data_all).  Zero mean."""
    d = X.shape[1]
    ell = hyper_row[:d]
    sf2 = hyper_row[d] ** 2
    sn2 = hyper_row[d + 1] ** 2
    K = gram(X, ell, sf2, sn2)
    L, info = chol_jitter(K)
    alpha = alpha_from_chol(L, y)
    invK = inv_from_chol(L) if want_invK else None
    return dict(L=L, alpha=alpha, invK=invK, info=info)


def fit(X, Y, hyper, want_invK=True):
    """All outputs: returns chol[Ny,N,N], alpha[Ny,N], invK[Ny,N,N], info[Ny]."""
    N, Ny = Y.shape
    chol = np.zeros((Ny, N, N))
    alpha = np.zeros((Ny, N))
    invK = np.zeros((Ny, N, N)) if want_invK else None
    info = np.zeros(Ny, dtype=np.int32)
    for a in range(Ny):
        r = fit_output(X, Y[:, a], hyper[a], want_invK)
        chol[a], alpha[a], info[a] = r['L'], r['alpha'], r['info']
        if want_invK:
            invK[a] = r['invK']
    return dict(chol=chol, alpha=alpha, invK=invK, info=info)


# --------------------------------------------------------------------------
# a7  negative log marginal likelihood (+ analytic gradient)
# --------------------------------------------------------------------------
def nll(hyper, X, y):
    """a7: `calc_NLL_numpy` optimize.py:322-356.  NLL = 0.5 y^T alpha +
    sum_i log|L_ii| (no N/2 log 2pi term, zero mean)."""
    n, D = X.shape
    ell = hyper[:D]
    sf2 = hyper[D] ** 2
    lik = hyper[D + 1] ** 2
    K = gram(X, ell, sf2, lik)
    L, _ = chol_jitter(K)
    logK = 2 * np.sum(np.log(np.abs(np.diag(L))))
    invLy = np.linalg.solve(L, y)
    alpha = np.linalg.solve(L.T, invLy)
    return 0.5 * np.dot(y.T, alpha) + 0.5 * logK


def nll_grad(hyper, X, y):
    """Analytic gradient of a7 w.r.t. hyper = [ell.., sf, sn] (raw, not log).
    NO reference function (optimize.py:371-375 says explicit gradients "should
    be implemented", SLSQP uses finite differences; the CasADi path :174-194
    gets it from AD).  Rasmussen & Williams (2006) eq. 5.9:
        dNLL/dtheta = 0.5 tr((K^-1 - alpha alpha^T) dK/dtheta).
    Checked against central differences of `nll` in tests."""
    n, D = X.shape
    ell = np.asarray(hyper[:D], dtype=np.float64)
    sf = hyper[D]
    sn = hyper[D + 1]
    Kse = cov_se_ard(X, X, ell, sf ** 2)
    K = Kse + sn ** 2 * np.eye(n)
    K = (K + K.T) * 0.5
    L, info = chol_jitter(K)
    alpha = alpha_from_chol(L, y)
    invK = inv_from_chol(L)
    W = invK - np.outer(alpha, alpha)
    g = np.zeros(D + 2)
    for dd in range(D):
        diff2 = (X[:, dd:dd + 1] - X[:, dd:dd + 1].T) ** 2
        g[dd] = 0.5 * np.sum(W * Kse * diff2) / ell[dd] ** 3
    g[D] = 0.5 * np.sum(W * Kse) * 2.0 / sf
    g[D + 1] = 0.5 * np.trace(W) * 2.0 * sn
    value = 0.5 * float(y @ alpha) + float(np.sum(np.log(np.abs(np.diag(L)))))
    return value, g


# --------------------------------------------------------------------------
# f4  prior mean functions
# --------------------------------------------------------------------------
def mean_param_count(func, Nx):
    """h_m of optimize.py:136-145."""
    if func == 'zero':
        return 0
    if func == 'const':
        return 1
    if func == 'linear':
        return Nx + 1
    if func == 'polynomial':
        return 2 * Nx + 1
    raise NameError('No mean function called: ' + func)        # gp_functions.py:67 / optimize.py:145


def mean_function(hyper_row, X, func='zero'):
    """`get_mean_function` gp_functions.py:25-69 evaluated on the rows of X[n, Nx] (the reference hands it X.T and
    indexes columns): the parameters are the LAST entries of the hyper row --
    const: a = hyp[-1]; linear: a = hyp[-Nx-1:-1], b = hyp[-1]; polynomial: a = hyp[-2Nx-1:-Nx-1], b = hyp[-Nx-1:-1],
    c = hyp[-1], m(x) = a^T x^2 + b^T x + c."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    n, Nx = X.shape
    hyp = np.asarray(hyper_row, dtype=np.float64)
    if func == 'zero':
        return np.zeros(n)
    if func == 'const':
        return np.full(n, hyp[-1])
    if func == 'linear':
        return X @ hyp[-Nx - 1:-1] + hyp[-1]
    if func == 'polynomial':
        return (X ** 2) @ hyp[-2 * Nx - 1:-Nx - 1] + X @ hyp[-Nx - 1:-1] + hyp[-1]
    raise NameError('No mean function called: ' + func)


def nll_mean(hyper, X, y, func='zero'):
    """`calc_NLL` optimize.py:22-97 with prior=None (:157): NLL(Y - m(X), alpha, log det) :95-97, i.e. the pinned
    `calc_NLL_numpy` objective on the residual y - m(X) (alpha itself is formed from the residual, :75)."""
    D = X.shape[1]
    return nll(np.asarray(hyper)[:D + 2], X, y - mean_function(hyper, X, func))


def nll_mean_grad(hyper, X, y, func='zero'):
    """Gradient of `nll_mean` w.r.t. the whole row: the kernel part as `nll_grad` on the residual, the mean part
    -alpha^T dm/dtheta (no reference function: CasADi AD inside IPOPT, optimize.py:168-194)."""
    D = X.shape[1]
    hyper = np.asarray(hyper, dtype=np.float64)
    r = y - mean_function(hyper, X, func)
    v, g = nll_grad(hyper[:D + 2], X, r)
    f = fit_output(X, r, hyper[:D + 2], want_invK=False)
    al = f['alpha']
    if func == 'const':
        gm = np.array([-al.sum()])
    elif func == 'linear':
        gm = np.concatenate([-(X.T @ al), [-al.sum()]])
    elif func == 'polynomial':
        gm = np.concatenate([-((X ** 2).T @ al), -(X.T @ al), [-al.sum()]])
    else:
        gm = np.zeros(0)
    return v, np.concatenate([g, gm])


def nll_prior(hyper, X, y, prior, func='zero'):
    """`calc_NLL` with hyper-priors, optimize.py:77-97 LITERALLY: prior_gauss(theta, mu, s2) = -(theta-mu)^2/(2 s2)
    - 0.5 log(2 pi s2) summed over every ell_i (ell_mean, ell_std^2), over sf2 = hyper[Nx]^2 (sf_mean, sf_std^2) and
    sn2 = hyper[Nx+1]^2 (sn_mean, sn_std^2), and `return NLL(...) + log_prior` (:97: the log-prior is added)."""
    D = X.shape[1]

    def prior_gauss(theta, mu, s2):
        return -(theta - mu) ** 2 / (2 * s2) - 0.5 * np.log(2 * np.pi * s2)
    log_prior = 0.0
    for i in range(D):
        log_prior += prior_gauss(hyper[i], prior['ell_mean'], prior['ell_std'] ** 2)
    log_prior += prior_gauss(hyper[D] ** 2, prior['sf_mean'], prior['sf_std'] ** 2)
    log_prior += prior_gauss(hyper[D + 1] ** 2, prior['sn_mean'], prior['sn_std'] ** 2)
    return nll_mean(hyper, X, y, func) + log_prior


def fit_mean(X, Y, hyper, func='zero', want_invK=True):
    """train_gp's recomputation at the optimum with a mean function, optimize.py:264-285: K, L, invK from the kernel
    part; alpha = K^-1 (y - m(X))."""
    N, Ny = Y.shape
    D = X.shape[1]
    R = np.stack([Y[:, a] - mean_function(hyper[a], X, func) for a in range(Ny)], axis=1)
    return fit(X, R, np.asarray(hyper)[:, :D + 2], want_invK)


def mean_func_jac(hyper_row, Z, func='zero'):
    """d m / d z at the rows of Z (what CasADi's jacobian of build_gp's mean adds, gp_functions.py:131,146-147)."""
    Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
    n, Nx = Z.shape
    hyp = np.asarray(hyper_row, dtype=np.float64)
    if func == 'linear':
        return np.tile(hyp[-Nx - 1:-1], (n, 1))
    if func == 'polynomial':
        return 2 * Z * hyp[-2 * Nx - 1:-Nx - 1] + hyp[-Nx - 1:-1]
    return np.zeros((n, Nx))


def train_bounds_mean(Nx, func, meanF, numpy_path=False):
    """Box bounds of the IPOPT path with a mean function, optimize.py:204-229 (numpy path :434-458 sets the same
    mean-parameter bounds AFTER `bounds` was assembled :443, so SLSQP never sees them)."""
    h_m = mean_param_count(func, Nx)
    lb = -np.inf * np.ones(Nx + 2 + h_m)
    ub = np.inf * np.ones(Nx + 2 + h_m)
    lb[:Nx], ub[:Nx] = (1 - 2, 2e2) if numpy_path else (1e-2, 1e2)
    lb[Nx], ub[Nx] = 1e-8, 1e2
    lb[Nx + 1], ub[Nx + 1] = 10 ** -10, 10 ** -2
    if numpy_path:
        return lb, ub
    if func == 'const':
        lb[-1], ub[-1] = -1e2, 1e2
    elif func != 'zero':
        lb[-1] = meanF / 10 - 1e-8
        ub[-1] = meanF * 10 + 1e-8
        lb[-h_m:-1] = -1e-2
        ub[-h_m:-1] = 1e-2
    return lb, ub


# --------------------------------------------------------------------------
# a8  multistart training
# --------------------------------------------------------------------------
def train_bounds(Nx):
    """numpy-path box bounds, optimize.py:434-443 (note lb_ell = 1-2 = -1)."""
    num_hyp = Nx + 2
    lb = -np.inf * np.ones(num_hyp)
    ub = np.inf * np.ones(num_hyp)
    lb[:Nx] = 1 - 2
    ub[:Nx] = 2e2
    lb[Nx] = 1e-8
    ub[Nx] = 1e2
    lb[Nx + 1] = 10 ** -10
    ub[Nx + 1] = 10 ** -2
    return lb, ub


def train_init(X, y):
    """optimize.py:445-449: ell = std(X), sf = std(y), sn = 1e-5."""
    Nx = X.shape[1]
    h = np.zeros(Nx + 2)
    h[:Nx] = np.std(X, 0)
    h[Nx] = np.std(y)
    h[Nx + 1] = 1e-5
    return h


def train(X, Y, multistart=1, hyper_init=None, optimizer_opts=None):
    """a8: `train_gp_numpy` optimize.py:359-503, zero mean: SLSQP with
    finite-difference gradients, tol=1e-12, every restart from the same
    initial point (F7), arg-min, then a1,a3-a6 at the optimum."""
    from scipy.optimize import minimize
    N, Nx = X.shape
    Ny = Y.shape[1]
    options = {'disp': False, 'maxiter': 10000}
    if optimizer_opts is not None:
        options.update(optimizer_opts)
    hyp_opt = np.zeros((Ny, Nx + 2))
    for a in range(Ny):
        lb, ub = train_bounds(Nx)
        bounds = np.hstack((lb.reshape(-1, 1), ub.reshape(-1, 1)))
        h0 = train_init(X, Y[:, a]) if hyper_init is None else hyper_init[a, :]
        obj = np.zeros(multistart)
        loc = np.zeros((multistart, Nx + 2))
        for i in range(multistart):
            res = minimize(nll, h0, args=(X, Y[:, a]), method='SLSQP',
                           options=options, bounds=bounds, tol=1e-12)
            obj[i] = res.fun
            loc[i, :] = res.x
        hyp_opt[a, :] = loc[np.argmin(obj)]
    f = fit(X, Y, hyp_opt)
    return dict(hyper=hyp_opt, lam_x=0, invK=f['invK'], alpha=f['alpha'],
                chol=f['chol'])


# --------------------------------------------------------------------------
# a9 / a10  mean, variance, mean Jacobian, TA covariance
# --------------------------------------------------------------------------
def mean_var_jac(Z, X, hyper, alpha, chol, want_jac=True, mean_func='zero'):
    """a9: `build_gp` gp_functions.py:72-149, zero mean, batched over rows of
    Z[B,d].  ks_i direct form :114-117; mean = ks^T alpha :119-120,135;
    v = L^-1 ks :122-123,133; var = sf^2 - v^T v :125-126,136 (kss = sf^2, no
    noise).  J = d mean / d z (:146-147 is CasADi AD; analytically
    J[a,dd] = sum_i alpha_i ks_i (X_i,dd - z_dd) / ell_a,dd^2)."""
    from scipy.linalg import solve_triangular
    Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
    B, d = Z.shape
    Ny = hyper.shape[0]
    mean = np.zeros((B, Ny))
    var = np.zeros((B, Ny))
    J = np.zeros((B, Ny, d)) if want_jac else None
    for a in range(Ny):
        ell = hyper[a, :d]
        sf2 = hyper[a, d] ** 2
        ks = cov_se_ard_direct(X, Z, ell, sf2)          # [N, B]
        mean[:, a] = ks.T @ alpha[a] + mean_function(hyper[a], Z, mean_func)     # :119-120,131,135
        v = solve_triangular(chol[a], ks, lower=True)
        var[:, a] = sf2 - np.sum(v * v, axis=0)
        if want_jac:
            w = ks * alpha[a][:, None]                  # [N, B]
            for dd in range(d):
                diff = X[:, dd:dd + 1] - Z[None, :, dd]  # [N, B]
                J[:, a, dd] = np.sum(w * diff, axis=0) / ell[dd] ** 2
            J[:, a, :] += mean_func_jac(hyper[a], Z, mean_func)
    return mean, var, J


def mean_var_sens(Z, X, hyper, alpha, chol):
    """Second-order information of `build_gp`'s functions (the reference gets
    it from CasADi AD of gp_functions.py:114-147 inside IPOPT, there is no reference function):
    Hm[b,a,p,q] = d2 mean_a/dz_p dz_q, dvar[b,a,p] = d var_a/dz_p.  With r_ip = (X_ip - z_p)/l_p^2:
    d ks_i/dz_p = ks_i r_ip, d2 ks_i/dz_p dz_q = ks_i (r_ip r_iq - delta_pq/l_p^2), and
    var = sf^2 - ks^T K^-1 ks gives dvar_p = -2 (K^-1 ks)^T d ks/dz_p (K^-1 from chol, two
    triangular solves).  Pinned by finite differences of `mean_var_jac` in tests/test_oracle.py."""
    from scipy.linalg import solve_triangular
    Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
    B, d = Z.shape
    Ny = hyper.shape[0]
    Hm = np.zeros((B, Ny, d, d))
    dvar = np.zeros((B, Ny, d))
    for a in range(Ny):
        ell2 = hyper[a, :d] ** 2
        sf2 = hyper[a, d] ** 2
        ks = cov_se_ard_direct(X, Z, hyper[a, :d], sf2)                     # [N, B]
        u = solve_triangular(chol[a], solve_triangular(chol[a], ks, lower=True), lower=True, trans='T')
        for b in range(B):
            r = (X - Z[b]) / ell2                                            # [N, d]
            ka = ks[:, b] * alpha[a]
            Hm[b, a] = (r * ka[:, None]).T @ r - np.diag(ka.sum() / ell2)
            dvar[b, a] = -2.0 * (r * (u[:, b] * ks[:, b])[:, None]).sum(axis=0)
    return Hm, dvar


def ta_cov_sens(var, J, Hm, dvar, Sigma):
    """Derivatives of the 'TA' covariance cov = diag(var) + J Sigma J^T (gp_functions.py:167-171)
    with respect to the input mean z and the input covariance Sigma, for ONE input:
    dcov_dz[a,c,p] = delta_ac dvar[a,p] + sum_de (Hm[a,d,p] S_de J[c,e] + J[a,d] S_de Hm[c,e,p]),
    dcov_dS[a,c,d,e] = J[a,d] J[c,e].  ('ME': Sigma = 0 leaves dcov_dz = diag(dvar), dcov_dS = 0.)"""
    Ny, d = J.shape
    dz = np.einsum('adp,de,ce->acp', Hm, Sigma, J) + np.einsum('ad,de,cep->acp', J, Sigma, Hm)
    for a in range(Ny):
        dz[a, a] += dvar[a]
    dS = np.einsum('ad,ce->acde', J, J)
    return dz, dS


def ta_cov(var, J, Sigma):
    """a10: `build_TA_cov` gp_functions.py:152-173:
    cov = diag(var) + J Sigma J^T, batched: var[B,Ny], J[B,Ny,d], Sigma[B,d,d]."""
    B, Ny = var.shape
    cov = np.einsum('bad,bde,bce->bac', J, Sigma, J)
    idx = np.arange(Ny)
    cov[:, idx, idx] += var
    return cov


# --------------------------------------------------------------------------
# a11  exact moment matching
# --------------------------------------------------------------------------
def maha(a1, b1, Q1):
    """`maha` gp_functions.py:421-430."""
    aQ = a1 @ Q1
    bQ = b1 @ Q1
    return (np.sum(aQ * a1, axis=1)[:, None] + np.sum(bQ * b1, axis=1)[None, :]
            - 2 * aQ @ b1.T)


def _absdet_qr(A):
    """`determinant` gp_functions.py:378-380: exp(trace(log(R))) of a QR
    factorisation, i.e. prod diag(R); equals |det A| for the SPD-ish
    arguments it is called with."""
    R = np.linalg.qr(A, mode='r')
    return float(np.abs(np.prod(np.diag(R))))


def exact_moment(invK, X, Y, hyper, inputmean, inputcov):
    """a11: `gp_exact_moment` gp_functions.py:344-418 (Deisenroth), zero
    prior mean, one input distribution N(inputmean[d], inputcov[d,d]).
    Restated line by line (works in log-hyper space like :367)."""
    hyper = np.log(np.asarray(hyper, dtype=np.float64))
    Ny = len(invK)
    N, Nx = X.shape
    inputmean = np.asarray(inputmean, dtype=np.float64).reshape(1, Nx)
    inputcov = np.asarray(inputcov, dtype=np.float64)
    mean = np.zeros((Ny, 1))
    beta = np.zeros((N, Ny))
    log_k = np.zeros((N, Ny))
    v = X - np.repeat(inputmean, N, axis=0)
    covariance = np.zeros((Ny, Ny))
    eye = np.eye(Nx)
    for a in range(Ny):
        beta[:, a] = invK[a] @ Y[:, a]
        iLambda = np.diag(np.exp(-2 * hyper[a, :Nx]))
        R = inputcov + np.diag(np.exp(2 * hyper[a, :Nx]))
        iR = iLambda @ (eye - np.linalg.solve(eye + inputcov @ iLambda,
                                              inputcov @ iLambda))
        T = v @ iR
        c = (np.exp(2 * hyper[a, Nx]) / np.sqrt(_absdet_qr(R))
             * np.exp(np.sum(hyper[a, :Nx])))
        q2 = c * np.exp(-np.sum(T * v, axis=1) * 0.5)
        qb = q2 * beta[:, a]
        mean[a] = np.sum(qb)
        t = np.repeat(np.exp(hyper[a, :Nx]).reshape(1, Nx), N, axis=0)
        v1 = v / t
        log_k[:, a] = 2 * hyper[a, Nx] - np.sum(v1 * v1, axis=1) * 0.5
    for a in range(Ny):
        ii = v / np.exp(2 * hyper[a, :Nx]).reshape(1, Nx)
        for b in range(a + 1):
            R = inputcov @ np.diag(np.exp(-2 * hyper[a, :Nx])
                                   + np.exp(-2 * hyper[b, :Nx])) + eye
            t = 1.0 / np.sqrt(_absdet_qr(R))
            ij = v / np.exp(2 * hyper[b, :Nx]).reshape(1, Nx)
            Q = np.exp(log_k[:, a][:, None] + log_k[:, b][None, :]
                       + maha(ii, -ij, np.linalg.solve(R, inputcov * 0.5)))
            A = np.outer(beta[:, a], beta[:, b])
            if b == a:
                A = A - invK[a]
            A = A * Q
            covariance[a, b] = t * np.sum(A)
            covariance[b, a] = covariance[a, b]
        covariance[a, a] = covariance[a, a] + np.exp(2 * hyper[a, Nx])
    covariance = covariance - mean @ mean.T
    return mean.reshape(Ny), covariance


def exact_moment_sens(invK, X, Y, hyper, inputmean, inputcov):
    """First derivatives of `exact_moment` (gp_functions.py:344-418) with respect to the input mean and the input
    covariance -- what CasADi's AD of gp_exact_moment hands to IPOPT when 'EM' is the MPC's propagation method
    (gp_class.py:220-224, mpc_class.py:412-413); there is no reference function.  The d x d entries of the covariance
    are independent variables (as in `ca.jacobian(..., covar_s)`); formulas are evaluated at a symmetric Sigma.
    With v_i = x_i - mu, P_a = (Sigma + Lambda_a)^-1, w_i = beta_ai q_ai:
        d mean_a/d mu    = P_a sum_i w_i v_i
        d mean_a/d Sigma = -1/2 P_a mean_a + 1/2 P_a (sum_i w_i v_i v_i^T) P_a
    and with Lab = Lambda_a^-1 + Lambda_b^-1, R = Sigma Lab + I, G = R^-T, z_ij = Lambda_a^-1 v_i + Lambda_b^-1 v_j,
    W = A o Q (A = beta_a beta_b^T - [a==b] K_a^-1), s = sum W, t = det(R)^-1/2:
        d(t s)/d mu    = t G sum_ij W_ij z_ij
        d(t s)/d Sigma = t (-1/2 G Lab s + 1/2 G (sum_ij W_ij [ii_i ii_i^T + ij_j ij_j^T + 2 ii_i ij_j^T]) G^T)
        cov_ab = t s + [a==b] sf_a^2 - mean_a mean_b  ->  product rule for the last term.
    Returns dmean_dz[Ny,d], dmean_dS[Ny,d,d], dcov_dz[Ny,Ny,d], dcov_dS[Ny,Ny,d,d].
    Pinned by complex-step differentiation of a complex-safe transcription of `exact_moment` (tests/test_oracle.py)."""
    H = np.asarray(hyper, dtype=np.float64)
    Ny = len(invK)
    N, d = X.shape
    mu = np.asarray(inputmean, dtype=np.float64).reshape(d)
    Sg = np.asarray(inputcov, dtype=np.float64)
    v = X - mu
    beta = np.stack([invK[a] @ Y[:, a] for a in range(Ny)], axis=1)
    mean = np.zeros(Ny)
    dm_dz = np.zeros((Ny, d))
    dm_dS = np.zeros((Ny, d, d))
    lk = np.zeros((N, Ny))
    for a in range(Ny):
        lam = H[a, :d] ** 2
        P = np.linalg.inv(Sg + np.diag(lam))
        c = H[a, d] ** 2 * np.prod(H[a, :d]) / np.sqrt(abs(np.linalg.det(Sg + np.diag(lam))))
        w = c * np.exp(-0.5 * np.einsum('ik,kl,il->i', v, P, v)) * beta[:, a]
        mean[a] = w.sum()
        dm_dz[a] = P @ (v.T @ w)
        dm_dS[a] = -0.5 * P * mean[a] + 0.5 * P @ ((v * w[:, None]).T @ v) @ P
        lk[:, a] = 2 * np.log(H[a, d]) - 0.5 * np.sum(v * v / lam, axis=1)
    dc_dz = np.zeros((Ny, Ny, d))
    dc_dS = np.zeros((Ny, Ny, d, d))
    for a in range(Ny):
        ila = 1.0 / H[a, :d] ** 2
        for b in range(a + 1):
            ilb = 1.0 / H[b, :d] ** 2
            lab = ila + ilb
            R = Sg * lab[None, :] + np.eye(d)
            t = 1.0 / np.sqrt(abs(np.linalg.det(R)))
            Sm = np.linalg.solve(R, Sg * 0.5)
            G = np.linalg.inv(R).T
            ii, ij = v * ila, v * ilb
            Q = np.exp(lk[:, a][:, None] + lk[:, b][None, :] + maha(ii, -ij, Sm))
            A = np.outer(beta[:, a], beta[:, b])
            if a == b:
                A = A - invK[a]
            W = A * Q
            s0 = W.sum()
            r, c = W.sum(axis=1), W.sum(axis=0)
            z1 = ila * (v.T @ r) + ilb * (v.T @ c)
            Xc = (v * ila).T @ W @ (v * ilb)                       # sum_ij W_ij ii_i ij_j^T
            # `maha` (:421-430) is written a Q a^T + b Q b^T - 2 a Q b^T: its cross term is NOT symmetrised in Q, so
            # the derivative w.r.t. independent entries of Sigma carries 2 Xc, not Xc + Xc^T
            ZZ = (ii * r[:, None]).T @ ii + (ij * c[:, None]).T @ ij + 2 * Xc
            dz = t * (G @ z1) - mean[b] * dm_dz[a] - mean[a] * dm_dz[b]
            dS = t * (-0.5 * (G * lab[None, :]) * s0 + 0.5 * G @ ZZ @ G.T) - mean[b] * dm_dS[a] - mean[a] * dm_dS[b]
            dc_dz[a, b] = dc_dz[b, a] = dz
            dc_dS[a, b] = dc_dS[b, a] = dS
    return dm_dz, dm_dS, dc_dz, dc_dS


# --------------------------------------------------------------------------
# a12  legacy methods
# --------------------------------------------------------------------------
def old_me(invK, X, Y, hyper, z):
    """a12 'old_ME': `gp` gp_functions.py:176-256, zero mean, alpha=None:
    mean = (ks^T K^-1) y, var = kss - (ks^T K^-1) ks, covar = diag(var)."""
    Ny = len(invK)
    N, Nx = X.shape
    z = np.asarray(z, dtype=np.float64).reshape(1, Nx)
    mean = np.zeros(Ny)
    var = np.zeros(Ny)
    for a in range(Ny):
        ell = hyper[a, :Nx]
        sf2 = hyper[a, Nx] ** 2
        kss = float(cov_se_ard_direct(z, z, ell, sf2)[0, 0])
        ks = cov_se_ard_direct(X, z, ell, sf2)[:, 0]
        ksT_invK = ks @ invK[a]
        mean[a] = ksT_invK @ Y[:, a]
        var[a] = kss - ksT_invK @ ks
    return mean, np.diag(var)


def old_ta(invK, X, Y, hyper, z, inputcovar):
    """a12 'old_TA': `gp_taylor_approx(..., diag=True)` gp_functions.py:259-340
    restated LITERALLY, including its self-documented bug (:325) and its
    CasADi linear indexing (`v[e]` on the N x Nx MX `v` is element e of
    column 0).  Zero mean."""
    Ny = len(invK)
    N, Nx = X.shape
    z = np.asarray(z, dtype=np.float64).reshape(1, Nx)
    mean = np.zeros(Ny)
    var = np.zeros(Nx)
    v = X - np.repeat(z, N, axis=0)
    covar_temp = np.zeros((Ny, Ny))
    covariance = np.zeros((Ny, Ny))
    d_mean = np.zeros((Ny, 1))
    dd_var = np.zeros((Ny, Ny))
    vflat = v.flatten(order='F')            # CasADi column-major linear index
    for a in range(Ny):
        ell = hyper[a, :Nx]
        w = 1 / ell ** 2
        sf2 = hyper[a, Nx] ** 2
        iK = invK[a]
        alpha = iK @ Y[:, a]
        kss = sf2
        ks = cov_se_ard_direct(X, z, ell, sf2)[:, 0]
        invKks = iK @ ks
        mean[a] = ks @ alpha
        var[a] = kss - ks @ invKks
        d_mean[a] = (w[a] * v[:, a] * ks) @ alpha
        for dd in range(Ny):
            for e in range(Ny):
                dd_var1a = (v[:, dd] * ks) @ iK
                dd_var1b = dd_var1a @ (vflat[e] * ks)
                dd_var2 = (vflat[dd] * vflat[e] * ks) @ invKks
                dd_var[dd, e] = -2 * w[dd] * w[e] * (dd_var1b + dd_var2)
                if dd == e:
                    dd_var[dd, e] = dd_var[dd, e] + 2 * w[dd] * (kss - var[dd])
        mean_mat = d_mean @ d_mean.T
        covar_temp[0, 0] = inputcovar[a, a]
        covariance[a, a] = var[a] + np.trace(covar_temp @ (.5 * dd_var + mean_mat))
    return mean, covariance


# --------------------------------------------------------------------------
# a13  GP.predict with the reference's standardisation conventions
# --------------------------------------------------------------------------
class OracleGP:
    """Numeric restatement of the `GP` object state + `predict`
    (gp_class.py:245-263), `set_method` (:193-242), `discrete_linearize`
    (:647-661), `covar` (:353-381), `validate` (:145-190), rollout loop of
    `predict_compare` (:777-804).  Built from stored factors (like
    `load_model` :737-743) or by `fit`."""

    def __init__(self, X, Y, hyper, chol=None, alpha=None, invK=None,
                 normalize=False, meta=None, gp_method='TA'):
        self.X = np.array(X, dtype=np.float64)
        self.Y = np.array(Y, dtype=np.float64)
        self.hyper = np.array(hyper, dtype=np.float64)
        self.N, self.Nx = self.X.shape
        self.Ny = self.Y.shape[1]
        self.Nu = self.Nx - self.Ny
        if chol is None:
            f = fit(self.X, self.Y, self.hyper)
            chol, alpha, invK = f['chol'], f['alpha'], f['invK']
        self.chol = np.array(chol)
        self.alpha = np.array(alpha)
        self.invK = None if invK is None else np.array(invK)
        self.normalize = normalize
        self.meta = None if meta is None else {k: np.array(v) for k, v in meta.items()}
        self.gp_method = gp_method

    def set_method(self, m):
        if m not in ('ME', 'TA', 'EM', 'old_ME', 'old_TA'):
            raise NameError('No GP method called: ' + m)   # gp_class.py:237
        self.gp_method = m

    def _predict_std(self, z, cov):
        d = self.Nx
        m = self.gp_method
        if m in ('ME', 'TA'):
            mean, var, J = mean_var_jac(z.reshape(1, d), self.X, self.hyper,
                                        self.alpha, self.chol, want_jac=(m == 'TA'))
            if m == 'ME':
                return mean[0], np.diag(var[0])              # gp_class.py:212-215
            return mean[0], ta_cov(var, J, cov.reshape(1, d, d))[0]   # :216-219
        if m == 'EM':
            return exact_moment(self.invK, self.X, self.Y, self.hyper, z, cov)  # :220-224
        if m == 'old_ME':
            return old_me(self.invK, self.X, self.Y, self.hyper, z)
        return old_ta(self.invK, self.X, self.Y, self.hyper, z, cov)

    def predict(self, x, u, cov):
        """gp_class.py:245-263: standardise x,u; un-standardise the MEAN only."""
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        u = np.asarray(u, dtype=np.float64).reshape(-1)
        if self.normalize:
            x = (x - self.meta['meanX']) / self.meta['stdX']
            u = (u - self.meta['meanU']) / self.meta['stdU']
        mean, c = self._predict_std(np.concatenate([x, u]), np.asarray(cov, dtype=np.float64))
        if self.normalize:
            mean = mean * self.meta['stdY'] + self.meta['meanY']
        return mean.reshape(self.Ny, 1), c

    def discrete_linearize(self, x0, u0, cov0):
        """gp_class.py:647-661: Jacobian of predict()[0] w.r.t. the
        STANDARDISED x and u (ME/TA/old_*: the analytic GP-mean Jacobian; EM: of the exact-moment mean)."""
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        u0 = np.asarray(u0, dtype=np.float64).reshape(-1)
        if self.normalize:
            x0 = (x0 - self.meta['meanX']) / self.meta['stdX']
            u0 = (u0 - self.meta['meanU']) / self.meta['stdU']
        z = np.concatenate([x0, u0]).reshape(1, self.Nx)
        if self.gp_method == 'EM':        # jac of gp_exact_moment's mean w.r.t. the input mean (gp_class.py:220-224,239-242)
            J = exact_moment_sens(self.invK, self.X, self.Y, self.hyper, z[0], np.asarray(cov0, dtype=np.float64))[0]
            return J[:, :self.Ny].copy(), J[:, self.Ny:].copy()
        _, _, J = mean_var_jac(z, self.X, self.hyper, self.alpha, self.chol)
        return J[0][:, :self.Ny].copy(), J[0][:, self.Ny:].copy()

    def covar(self, X_new):
        """a14: gp_class.py:353-381 (LU solve against L, expanded-form ks)."""
        X_new = np.atleast_2d(np.asarray(X_new, dtype=np.float64))
        n, D = X_new.shape
        out = np.zeros((D, n, n))
        for a in range(self.Ny):
            ell = self.hyper[a, :self.Nx]
            sf2 = self.hyper[a, self.Nx] ** 2
            ks = cov_se_ard(self.X, X_new, ell, sf2)
            v = np.linalg.solve(self.chol[a], ks)
            out[a] = sf2 - v.T @ v
        return out

    def noise_variance(self):
        return self.hyper[:, self.Nx + 1] ** 2            # gp_class.py:675-678

    def validate(self, X_test, Y_test):
        """a16: gp_class.py:145-190 (SMSE divides by std, :166)."""
        X_test = np.array(X_test, dtype=np.float64)
        Y_test = np.array(Y_test, dtype=np.float64)
        if self.normalize:
            Y_test = (Y_test - self.meta['meanY']) / self.meta['stdY']
            X_test = (X_test - self.meta['meanZ']) / self.meta['stdZ']
        mean, var, _ = mean_var_jac(X_test, self.X, self.hyper, self.alpha,
                                    self.chol, want_jac=False)
        var = var + self.noise_variance()
        N = Y_test.shape[0]
        loss = np.sum((Y_test - mean) ** 2, axis=0) / N
        NLP = np.sum(0.5 * np.log(2 * np.pi * var) + (Y_test - mean) ** 2 / (2 * var), axis=0)
        return loss / np.std(Y_test, 0), NLP / N

    def rollout(self, x0, U, methods=('EM', 'TA', 'ME'), feedback=False, x_ref=None, Q=None, R=None, K=None):
        """a17: the numeric loop of `predict_compare` gp_class.py:777-804: covar = eye(d)*1e-6 with the state block
        diag(sn^2), feed (mean_t, cov_t) back; variances un-standardised by stdY^2.
        feedback=True (:772-803): K from `lqr` (mpc_class.py:972-973) of `discrete_linearize` at (x0, u[0]) unless
        given, u_t = K (mean_t - x_ref) evaluated on VECTORS (the reference's `mean_t - x_ref` mixes an (Ny x 1) array
        with an (Ny,) one, :790 -- its own "#TODO: Fix feedback"), control blocks of covar from K and cov_t, and
        -- literally as the reference -- `covar` is never reset between methods."""
        import scipy.linalg
        Nx, Ny = self.Nx, self.Ny
        U = np.atleast_2d(np.asarray(U, dtype=np.float64))
        Nt = U.shape[0]
        initVar = self.hyper[:, Nx + 1] ** 2
        mean = np.zeros((len(methods), Nt + 1, Ny))
        var = np.zeros((len(methods), Nt + 1, Ny))
        covar = np.eye(Nx) * 1e-6
        if Q is None:
            Q = np.eye(Ny)
        if R is None:
            R = np.eye(Nx - Ny)
        if x_ref is None and feedback:
            x_ref = np.zeros(Ny)
        keep = self.gp_method
        self.controls = np.zeros((len(methods), Nt, Nx - Ny))
        for i, m in enumerate(methods):
            self.set_method(m)
            mean_t = np.asarray(x0, dtype=np.float64).reshape(Ny)
            covar[:Ny, :Ny] = np.diag(initVar)
            mean[i, 0, :] = mean_t
            if feedback:
                if K is None:
                    A, B = self.discrete_linearize(mean_t, U[0], covar)
                    P = np.array(scipy.linalg.solve_discrete_are(A, B, Q, R))
                    Km = -np.array(scipy.linalg.solve(R + B.T @ P @ B, B.T @ P @ A))
                else:
                    Km = np.asarray(K, dtype=np.float64)
            for t in range(1, Nt + 1):
                u_t = Km @ (mean_t - x_ref) if feedback else U[t - 1]
                self.controls[i, t - 1] = u_t
                mean_t, covar_x = self.predict(mean_t, u_t, covar)
                mean_t = mean_t.reshape(Ny)
                mean[i, t, :] = mean_t
                var[i, t, :] = np.diag(covar_x)
                if self.normalize:
                    var[i, t, :] = var[i, t, :] * self.meta['stdY'] ** 2
                if feedback:
                    covar_u = Km @ covar_x @ Km.T
                    cov_xu = covar_x @ Km.T
                    covar[Ny:, Ny:] = covar_u
                    covar[Ny:, :Ny] = cov_xu.T
                    covar[:Ny, Ny:] = cov_xu
                covar[:Ny, :Ny] = covar_x
        self.set_method(keep)
        return mean, var


# --------------------------------------------------------------------------
# synthetic workloads: the generator lives in the product package
# (gp_mpc_amd/synthetic.py: data only, no GP arithmetic) so that bench.py needs oracle/ only
# for its cpu_baseline leg; re-exported here for the tests that take it from the oracle
# --------------------------------------------------------------------------
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
from gp_mpc_amd.synthetic import synthetic_problem  # noqa: E402,F401
