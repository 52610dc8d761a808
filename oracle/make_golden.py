#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own numpy code.

Runs ONLY in the build container (needs /root/reference).  Nothing here, and
no reference source, travels to the GPU box: the outputs are plain data
(inputs + expected outputs) committed under tests/golden/.

Recipe: casadi and pyDOE are not installable, so
empty stub modules are injected before `import gp_mpc`; the pure-numpy parts
of the hot path then run unmodified:
  optimize.calc_cov_matrix   (a1)      optimize.calc_NLL_numpy (a3,a4,a5,a7)
  GP.covSEard (a1 two-input) GP.covar  (a14, via object.__new__)
  optimize.train_gp_numpy    (a8,a6; zero mean, get_mean_function patched to
                              its 'zero' definition gp_functions.py:44-47)
plus the two saved models examples/models/gp_{tank,car}_example.json, the only
known-answer artefacts the reference ships (written by gp_class.py:693-726).

Usage:  python oracle/make_golden.py [legacy | ta | em | refmodel]      (one group only; the other fixtures stay untouched)
"""
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def import_reference():
    ca = types.ModuleType('casadi')
    ca.tools = types.ModuleType('casadi.tools')
    sys.modules.update({'casadi': ca, 'casadi.tools': ca.tools,
                        'pyDOE': types.ModuleType('pyDOE')})
    import matplotlib
    matplotlib.use('Agg')
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import gp_mpc  # noqa: F401
    from gp_mpc import optimize, GP
    ca.MX = lambda x: x
    optimize.get_mean_function = \
        lambda h, Xt, func='zero': (lambda A: np.zeros(A.shape[1]))
    return optimize, GP


def ref_gp(GP, X, hyper, chol):
    d = X.shape[1]
    g = object.__new__(GP)
    g._GP__Ny, g._GP__X, g._GP__chol = hyper.shape[0], X, chol
    g._GP__hyper_length_scales = hyper[:, :d]
    g._GP__hyper_signal_variance = hyper[:, d] ** 2
    return g


def pack_lower(A):
    """[Ny,N,N] -> [Ny, N(N+1)/2] row-major lower triangle."""
    il = np.tril_indices(A.shape[-1])
    return np.stack([a[il] for a in A])


def from_model(optimize, GP, name, n_test, seed):
    d = json.load(open(f'{REF}/examples/models/gp_{name}_example.json'))
    X = np.array(d['X'])
    Y = np.array(d['Y'])
    H = np.array(d['hyper']['hyper'])
    chol = np.array(d['hyper']['chol'])
    alpha = np.array(d['hyper']['alpha'])
    invK = np.array(d['hyper']['invK'])
    N, D = X.shape
    Ny = Y.shape[1]
    assert np.all(np.triu(chol[0], 1) == 0.0)
    rng = np.random.default_rng(seed)
    Z = X[rng.integers(0, N, n_test)] + 0.25 * rng.standard_normal((n_test, D)) * X.std(0)
    g = ref_gp(GP, X, H, chol)
    covar = g.covar(Z.copy())[:Ny]                       # reference a14
    K = np.stack([optimize.calc_cov_matrix(X, H[a, :D], H[a, D] ** 2)   # reference a1
                  for a in range(Ny)])
    ks = np.stack([g.covSEard(X.copy(), Z.copy(), H[a, :D], H[a, D] ** 2)  # reference a1 (2-input)
                   for a in range(Ny)])
    nll = np.array([float(optimize.calc_NLL_numpy(H[a], X, Y[:, a]))     # reference a7
                    for a in range(Ny)])
    out = dict(X=X, Y=Y, hyper=H, alpha=alpha,
               chol_packed=pack_lower(chol), invK_packed=pack_lower(invK),
               length_scale=np.array(d['hyper']['length_scale']),
               signal_var=np.array(d['hyper']['signal_var']),
               noise_var=np.array(d['hyper']['noise_var']),
               hyper_mean=np.array(d['hyper']['mean']),
               normalize=np.array(bool(d['normalize'])),
               Z=Z, ref_covar_diag=np.stack([np.diag(c) for c in covar]),
               ref_covar=covar, ref_K_packed=pack_lower(K), ref_ks=ks, ref_nll=nll)
    if d.get('normalize'):
        for k, v in d['meta'].items():
            out['meta_' + k] = np.array(v)
        for k in ('xlb', 'xub', 'ulb', 'uub'):
            out[k] = np.array(d[k])
    np.savez_compressed(os.path.join(OUT, f'{name}_model.npz'), **out)
    print(name, 'N', N, 'D', D, 'Ny', Ny, 'nll', nll)


def synthetic(optimize, GP):
    """Small seeded problem run through the reference's numpy training path."""
    rng = np.random.default_rng(20180101)
    N, d, Ny = 40, 2, 2
    X = rng.uniform(-2, 2, (N, d))
    Y = np.stack([np.sin(X[:, 0]) * np.cos(0.5 * X[:, 1]),
                  0.3 * X[:, 0] ** 2 - X[:, 1]], axis=1) + 1e-3 * rng.standard_normal((N, Ny))
    opt = optimize.train_gp_numpy(X, Y, multistart=1, optimizer_opts={'disp': False})  # reference a8
    H = opt['hyper']
    nll = np.array([float(optimize.calc_NLL_numpy(H[a], X, Y[:, a])) for a in range(Ny)])
    # NLL on a grid of hyper rows (a7) incl. one that needs the jitter branch
    probes = np.array([[1.0, 1.0, 1.0, 1e-2], [0.5, 2.0, 0.7, 1e-3],
                       [3.0, 3.0, 2.0, 1e-5], [50.0, 50.0, 1.0, 1e-10]])
    import io
    import contextlib
    probe_nll = np.zeros((len(probes), Ny))
    probe_jit = np.zeros((len(probes), Ny), dtype=np.int32)
    for i, h in enumerate(probes):
        for a in range(Ny):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                probe_nll[i, a] = float(optimize.calc_NLL_numpy(h, X, Y[:, a]))
            probe_jit[i, a] = int('jitter' in buf.getvalue())
    Z = rng.uniform(-2, 2, (16, d))
    g = ref_gp(GP, X, H, opt['chol'])
    covar = g.covar(Z.copy())[:Ny]
    np.savez_compressed(os.path.join(OUT, 'train_small.npz'), X=X, Y=Y, hyper=H,
                        chol=opt['chol'], alpha=opt['alpha'], invK=opt['invK'],
                        nll=nll, probes=probes, probe_nll=probe_nll, probe_jitter=probe_jit,
                        Z=Z, ref_covar=covar)
    print('train_small hyper', H, 'nll', nll, 'jitter flags', probe_jit.tolist())


def synthetic2(optimize, GP):
    """A second run of the reference's numpy training path: three inputs with different scales, one output whose noise
    (3e-2) sits above the reference's upper bound on sn (1e-2, optimize.py:441-442), so the optimum has an active bound."""
    rng = np.random.default_rng(20190607)
    N, d = 60, 3
    X = rng.uniform(-1, 1, (N, d)) * np.array([1.0, 4.0, 0.3])
    Y = (np.tanh(X[:, 0]) + 0.2 * np.sin(X[:, 1]) + X[:, 2] ** 2 + 3e-2 * rng.standard_normal(N))[:, None]
    opt = optimize.train_gp_numpy(X, Y, multistart=1, optimizer_opts={'disp': False})  # reference a8
    H = opt['hyper']
    nll = np.array([float(optimize.calc_NLL_numpy(H[0], X, Y[:, 0]))])
    Z = rng.uniform(-1, 1, (12, d)) * np.array([1.0, 4.0, 0.3])
    g = ref_gp(GP, X, H, opt['chol'])
    covar = g.covar(Z.copy())[:1]
    K = optimize.calc_cov_matrix(X, H[0, :d], H[0, d] ** 2)[None]
    np.savez_compressed(os.path.join(OUT, 'train_small2.npz'), X=X, Y=Y, hyper=H, chol=opt['chol'], alpha=opt['alpha'],
                        invK=opt['invK'], nll=nll, Z=Z, ref_covar=covar, ref_K=K)
    print('train_small2 hyper', H, 'nll', nll)


def synthetic3(optimize, GP):
    """Third training run: standardised data, three outputs with noise 1e-3, 3e-2 and 0.3.  The reference's SLSQP (finite-
    difference gradient) ends with sn on its LOWER bound for the first output and does not leave the start at all for the
    other two (their NLL at the start is 8e5 and 3e7): what `train_gp_numpy` returns in such cases is part of a8 too."""
    rng = np.random.default_rng(77)
    N, d = 45, 2
    X = rng.standard_normal((N, d))
    F = np.stack([np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1], np.cos(X[:, 0] * X[:, 1]), X[:, 0] ** 2 - 0.5 * X[:, 1]], axis=1)
    Y = F + np.array([1e-3, 3e-2, 0.3]) * rng.standard_normal((N, 3))
    Y = (Y - Y.mean(0)) / Y.std(0)
    opt = optimize.train_gp_numpy(X, Y, multistart=1, optimizer_opts={'disp': False})  # reference a8
    H = opt['hyper']
    nll = np.array([float(optimize.calc_NLL_numpy(H[a], X, Y[:, a])) for a in range(3)])
    np.savez_compressed(os.path.join(OUT, 'train_small3.npz'), X=X, Y=Y, hyper=H, chol=opt['chol'], alpha=opt['alpha'],
                        invK=opt['invK'], nll=nll)
    print('train_small3 hyper', H, 'nll', nll)


def legacy_pin(optimize, GP, name):
    """Reference-made pin for a12 'old_ME' (`gp`, gp_functions.py:176-256 with alpha=None): per output
        mean = (ks^T K^-1) y   (:229, :246),      var = kss - (ks^T K^-1) ks   (:231-232, :247),   kss = covSE(z, z) = sf^2.
    The CasADi graph itself cannot run here (casadi absent), but its two matrix products can be composed from what the
    reference DOES run and ship: ks from its own numpy GP.covSEard (gp_class.py:314-350) at the test points of
    <name>_model.npz, K^-1 and Y as stored in its saved model (written by gp_class.py:693-726).  The products are formed in
    the graph's order ((ks^T K^-1) first).  Written to a file of its own so that the other fixtures stay byte-identical."""
    d = json.load(open(f'{REF}/examples/models/gp_{name}_example.json'))
    X, Y = np.array(d['X']), np.array(d['Y'])
    H = np.array(d['hyper']['hyper'])
    invK = np.array(d['hyper']['invK'])
    chol = np.array(d['hyper']['chol'])
    D, Ny = X.shape[1], Y.shape[1]
    Z = np.load(os.path.join(OUT, f'{name}_model.npz'))['Z']
    g = ref_gp(GP, X, H, chol)
    mean = np.zeros((Ny, len(Z)))
    var = np.zeros((Ny, len(Z)))
    mscale = np.zeros((Ny, len(Z)))
    vscale = np.zeros((Ny, len(Z)))
    for a in range(Ny):
        sf2 = H[a, D] ** 2
        ks = g.covSEard(X.copy(), Z.copy(), H[a, :D], sf2)          # [N, n_test], reference a1 (two-input)
        ksT_invK = ks.T @ invK[a]                                     # gp_functions.py:221-222
        mean[a] = ksT_invK @ Y[:, a]                                  # :229
        var[a] = sf2 - np.sum(ksT_invK * ks.T, axis=1)                # :231-232, kss = sf2 exp(0)
        # what one rounding of the operands moves these sums by (the comparison scale; cond(K) up to 7e10 on the car model)
        mscale[a] = (np.abs(ks).T @ np.abs(invK[a])) @ np.abs(Y[:, a])
        vscale[a] = np.sum((np.abs(ks).T @ np.abs(invK[a])) * np.abs(ks).T, axis=1)
    np.savez_compressed(os.path.join(OUT, f'{name}_old_me.npz'), Z=Z, ref_old_me_mean=mean, ref_old_me_var=var,
                        mean_scale=mscale, var_scale=vscale)
    print(name, 'old_ME pin: mean', mean[:, :2], 'var', var[:, :2])


def _ref_mean_complex(g, X, Zc, H, alpha):
    """mean_a(z) = covSEard(X, z)^T alpha_a through the REFERENCE's own numpy kernel (gp_class.py:314-350; its operation
    order is complex-safe: squares, sums, one dot, one exp), alpha as the reference stored / trained it."""
    D, Ny = X.shape[1], H.shape[0]
    Xc = X.astype(Zc.dtype)
    return np.stack([g.covSEard(Xc.copy(), Zc.copy(), H[a, :D], H[a, D] ** 2).T @ alpha[a] for a in range(Ny)], axis=1)


def _ref_var(g, Zq, Ny, chunk=800):
    """var_a(z) = diag of the reference's GP.covar (gp_class.py:353-381), evaluated in chunks of points (covar forms the
    n x n matrix per output)."""
    out = np.zeros((len(Zq), Ny))
    for s in range(0, len(Zq), chunk):
        c = g.covar(Zq[s:s + chunk].copy())[:Ny]
        out[s:s + chunk] = np.stack([np.diag(ci) for ci in c], axis=1)
    return out


def ta_pin(optimize, GP, name):
    """Reference-run pin for a9's Jacobian and a10 'TA' (`build_gp` / `build_TA_cov`, gp_functions.py:146-147,152-173; CasADi
    graphs, not runnable here), composed from code the reference DOES run:
        mean(z) = GP.covSEard(X, z)^T alpha                    (gp_class.py:314-350, alpha from the saved model)
        J       = Im mean(z + i h e_p) / h,  h = 1e-30         (complex step through that same function: exact to rounding)
        var(z)  = diag GP.covar(z)                             (gp_class.py:353-381)
        cov     = diag(var) + J Sigma J^T                      (the one line of build_TA_cov, :167-171)
    at the test points of <name>_model.npz with seeded SPD input covariances."""
    d = json.load(open(f'{REF}/examples/models/gp_{name}_example.json'))
    X = np.array(d['X'])
    H = np.array(d['hyper']['hyper'])
    chol = np.array(d['hyper']['chol'])
    alpha = np.array(d['hyper']['alpha'])
    D, Ny = X.shape[1], H.shape[0]
    Z = np.load(os.path.join(OUT, f'{name}_model.npz'))['Z']
    n = len(Z)
    rng = np.random.default_rng(31 + len(name))
    A = rng.standard_normal((n, D, D)) * X.std(0)[None, :, None]
    Sigma = 1e-2 * A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(D)
    g = ref_gp(GP, X, H, chol)
    mean = _ref_mean_complex(g, X, Z.astype(complex), H, alpha).real
    J = np.zeros((n, Ny, D))
    h = 1e-30
    for p in range(D):
        Zc = Z.astype(complex)
        Zc[:, p] += 1j * h
        J[:, :, p] = _ref_mean_complex(g, X, Zc, H, alpha).imag / h
    var = _ref_var(g, Z, Ny)
    cov = np.einsum('bad,bde,bce->bac', J, Sigma, J)
    cov[:, np.arange(Ny), np.arange(Ny)] += var
    # comparison scales: the size of the terms of the sums (cond(K) up to 7e10 on the car model makes alpha huge)
    mscale = np.zeros((n, Ny))
    jscale = np.zeros((n, Ny, D))
    for a in range(Ny):
        ks = g.covSEard(X.copy(), Z.copy(), H[a, :D], H[a, D] ** 2)
        mscale[:, a] = np.abs(ks).T @ np.abs(alpha[a])
        for p in range(D):
            jscale[:, a, p] = (np.abs(ks * alpha[a][:, None]) * np.abs(X[:, p:p + 1] - Z[None, :, p])).sum(0) / H[a, p] ** 2
    np.savez_compressed(os.path.join(OUT, f'{name}_ta.npz'), Z=Z, Sigma=Sigma, ref_mean=mean, ref_J=J, ref_var=var,
                        ref_ta_cov=cov, mean_scale=mscale, J_scale=jscale)
    print(name, 'TA pin: mean', mean[0], 'J', J[0, 0], 'cov diag', np.diag(cov[0]))


def em_pin(optimize, GP):
    """Reference-run pins for a11 'EM' (`gp_exact_moment`, gp_functions.py:344-430; a CasADi graph): what it computes in closed
    form are the first two moments of the GP prediction under z ~ N(mu, Sigma),
        E[mean(z)],   Cov[mean(z)] + diag(E[var(z)]),
    here by tensor Gauss-Hermite quadrature (80 x 80 nodes, checked against 64 x 64) of the REFERENCE's own numeric
    predictor -- GP.covSEard(X, z)^T alpha and diag GP.covar(z) -- on two models the reference's `train_gp_numpy` produced
    (d = 2, Ny = 2, N = 40; chol, alpha, invK as the reference returned them):
      train_small.npz  smooth data, sn ~ 1e-3: cond(K) = 2e7 / 7e8.  The closed form works on K^-1 (beta = K^-1 y, sum of
                       K^-1_ij Q_ij against sf^2), so in fp64 it carries cond * eps * sf^2 ~ 1e-6 of rounding there; the
                       quadrature (Cholesky-based predictor) does not.  Pin at that level only.
      em_model2.npz    rough noisy data (made here, trained by the reference from a stated start): cond(K) ~ 4e3 -> the pin
                       holds to the quadrature's own accuracy (<= 1e-11 observed, gated at 1e-9)."""
    rng = np.random.default_rng(606)
    N, D, Ny = 40, 2, 2
    X2 = rng.uniform(-2, 2, (N, D))
    Y2 = np.stack([np.sin(3 * X2[:, 0]) * np.cos(2 * X2[:, 1]), np.cos(2.5 * X2[:, 0] + 1.5 * X2[:, 1])], axis=1) \
        + 0.1 * rng.standard_normal((N, Ny))
    start = np.array([[0.6, 0.6, 0.8, 1e-2], [0.7, 0.9, 0.8, 1e-2]])
    opt = optimize.train_gp_numpy(X2, Y2, multistart=1, optimizer_opts={'disp': False}, hyper_init=start)   # reference a8
    np.savez_compressed(os.path.join(OUT, 'em_model2.npz'), X=X2, Y=Y2, hyper=opt['hyper'], chol=opt['chol'],
                        alpha=opt['alpha'], invK=opt['invK'], hyper_init=start)
    print('em_model2 hyper', opt['hyper'], 'cond', [np.linalg.cond(L @ L.T) for L in opt['chol']])

    cases = {'train_small': [(np.array([0.3, -0.4]), np.array([[0.09, 0.03], [0.03, 0.16]])),
                             (np.array([-1.1, 0.6]), np.array([[0.5, -0.2], [-0.2, 0.3]])),
                             (np.array([1.5, 1.2]), np.array([[0.02, 0.0], [0.0, 0.4]])),
                             (np.array([0.0, 0.0]), np.array([[1e-4, 5e-5], [5e-5, 1e-4]]))],
             'em_model2': [(np.array([0.3, -0.4]), np.array([[0.02, 0.008], [0.008, 0.03]])),
                           (np.array([-1.1, 0.6]), np.array([[0.04, -0.015], [-0.015, 0.025]])),
                           (np.array([1.2, 1.0]), np.array([[0.005, 0.0], [0.0, 0.04]])),
                           (np.array([0.0, 0.0]), np.array([[1e-4, 5e-5], [5e-5, 1e-4]]))]}
    for model, cs in cases.items():
        t = np.load(os.path.join(OUT, model + '.npz'))
        X, H, chol, alpha = t['X'], t['hyper'], t['chol'], t['alpha']
        g = ref_gp(GP, X, H, chol)

        def quad(mu, Sigma, n):
            tt, w = np.polynomial.hermite_e.hermegauss(n)
            w = w / np.sqrt(2 * np.pi)
            A = np.linalg.cholesky(Sigma)
            T = np.stack([gg.ravel() for gg in np.meshgrid(tt, tt, indexing='ij')], axis=1)
            W = np.outer(w, w).ravel()
            Zq = mu + T @ A.T
            m = _ref_mean_complex(g, X, Zq, H, alpha)
            v = _ref_var(g, Zq, Ny)
            Em, Ev = W @ m, W @ v
            return Em, (m * W[:, None]).T @ m - np.outer(Em, Em) + np.diag(Ev)

        mus, Sigmas, means, covs, conv = [], [], [], [], []
        for mu, Sigma in cs:
            m80, c80 = quad(mu, Sigma, 80)
            m64, c64 = quad(mu, Sigma, 64)
            mus.append(mu); Sigmas.append(Sigma); means.append(m80); covs.append(c80)
            conv.append([np.abs(m80 - m64).max(), np.abs(c80 - c64).max()])
            print(model, 'EM pin: mu', mu, 'mean', m80, 'cov', c80.ravel(), '80 vs 64 nodes', conv[-1])
        np.savez_compressed(os.path.join(OUT, model + '_em.npz'), mu=np.array(mus), Sigma=np.array(Sigmas),
                            ref_em_mean=np.array(means), ref_em_cov=np.array(covs), quad_convergence=np.array(conv))


def ref_written_model(optimize, GP):
    """A model file WRITTEN BY THE REFERENCE: its `GP.optimize` (gp_class.py:82-142: normalisation + `train_gp_numpy`) and
    its `GP.save_model` (:693-734) run here on a small seeded data set (N = 30, Ny = 2, Nu = 1, normalize=True), the instance
    made with object.__new__ because `GP.__init__` goes on to build CasADi graphs.  The JSON is data (tests/golden/
    ref_written_model.json); next to it the reference's GP.covar / covSEard^T alpha at test points in the standardised
    coordinates the model works in."""
    rng = np.random.default_rng(4242)
    N, Ny, Nu = 30, 2, 1
    X = rng.uniform(-1, 1, (N, Ny + Nu)) * np.array([2.0, 0.5, 3.0]) + np.array([0.5, -1.0, 0.0])
    Y = np.stack([np.sin(X[:, 0]) + 0.3 * X[:, 2], X[:, 1] * np.cos(0.5 * X[:, 0]) - 0.1 * X[:, 2] ** 2], axis=1)
    Y = Y + 1e-3 * rng.standard_normal(Y.shape)
    g = object.__new__(GP)
    g._GP__X, g._GP__Y = X.copy(), Y.copy()
    g._GP__Ny, g._GP__Nx, g._GP__N, g._GP__Nu = Ny, Ny + Nu, N, Nu
    g._GP__gp_method = 'TA'
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        g.optimize(X=X, Y=Y, opts={'disp': False}, mean_func='zero', xlb=[-2.0, -2.0], xub=[3.0, 0.0], ulb=[-3.0], uub=[3.0],
                   multistart=1, normalize=True, optimize_nummeric=True)
    g.save_model(os.path.join(OUT, 'ref_written_model'))
    Zs = rng.standard_normal((10, Ny + Nu))                                 # standardised coordinates
    covar = g.covar(Zs.copy())[:Ny]
    mean = _ref_mean_complex(g, g._GP__X, Zs, g._GP__hyper, g._GP__alpha)
    np.savez_compressed(os.path.join(OUT, 'ref_written_model_outputs.npz'), Zs=Zs, ref_covar=covar, ref_mean_std=mean,
                        X_raw=X, Y_raw=Y)
    print('reference-written model:', os.path.getsize(os.path.join(OUT, 'ref_written_model.json')), 'bytes; hyper',
          g._GP__hyper)


def main():
    os.makedirs(OUT, exist_ok=True)
    optimize, GP = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'legacy':       # only the old_ME pins (the other fixtures untouched)
        legacy_pin(optimize, GP, 'tank')
        legacy_pin(optimize, GP, 'car')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'ta':           # r06: reference-run pins for J / TA
        ta_pin(optimize, GP, 'tank')
        ta_pin(optimize, GP, 'car')
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'em':           # r06: reference-run pin for EM
        em_pin(optimize, GP)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'refmodel':     # r06: a model file written by the reference
        ref_written_model(optimize, GP)
        return
    from_model(optimize, GP, 'tank', 24, 1)
    from_model(optimize, GP, 'car', 24, 2)
    synthetic(optimize, GP)
    synthetic2(optimize, GP)
    synthetic3(optimize, GP)
    legacy_pin(optimize, GP, 'tank')
    legacy_pin(optimize, GP, 'car')
    ta_pin(optimize, GP, 'tank')
    ta_pin(optimize, GP, 'car')
    em_pin(optimize, GP)
    ref_written_model(optimize, GP)


if __name__ == '__main__':
    main()
