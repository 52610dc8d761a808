"""CasADi `Callback` around the HIP GP predictor.

In the reference `GP.predict` returns a symbolic `ca.Function` call that is inlined into the NLP
graph (gp_class.py:207-242,259; mpc_class.py:412-413).  A GPU predictor cannot be inlined, so an
MPC layer uses it through this callback instead: same signature `(x[Ny], u[Nu], covar[Nx x Nx])
-> (mean[Ny], cov[Ny x Ny])` as `__predict` (gp_class.py:212-224), with `get_jacobian` served by
`jacobian_blocks` below: exact derivatives from one device call for 'ME', 'TA' (`gpmpc_predict_sens`)
and 'EM' (`gpmpc_predict_em_sens`), central differences of the device predictor -- mean AND
covariance, so that values and derivatives stay consistent -- for the legacy 'old_ME' / 'old_TA'.
Use it with IPOPT options `expand=False` (a Callback cannot be flattened to SX; note
mpc_class.py:169 reads solver_opts['expand']) and `hessian_approximation='limited-memory'`.

casadi is not installable in the build image: the Callback classes are import-guarded
and cannot be exercised here.  Everything numeric they do lives in `jacobian_blocks`, which needs no
casadi and IS tested (tests/parity_cases.py::check_callback_blocks).
"""
try:
    import casadi as ca
except Exception:          # pragma: no cover - casadi absent in this image
    ca = None

import numpy as np


def jacobian_blocks(gp, x, u, S, fd_eps=1e-6):
    """The six dense Jacobian blocks a casadi Callback's `get_jacobian` function returns for
    `(x, u, covar) -> (mean, cov)`, in CasADi's layout: matrices are vectorised COLUMN-major, so
        [d mean/dx (Ny x Ny), d mean/du (Ny x Nu), d mean/d vec(covar) (Ny x Nx^2),
         d vec(cov)/dx (Ny^2 x Ny), d vec(cov)/du (Ny^2 x Nu), d vec(cov)/d vec(covar) (Ny^2 x Nx^2)],
    row index of vec(cov) = a + Ny c, column index of vec(covar) = p + Nx q.  Derivatives are with respect to the
    RAW x, u (chain rule through GP.predict's standardisation, gp_class.py:253-261)."""
    N, Ny, Nu = gp.get_size()
    Nx = Ny + Nu
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    u = np.asarray(u, dtype=np.float64).reshape(-1)
    S = np.asarray(S, dtype=np.float64).reshape(Nx, Nx)
    method = gp._GP__gp_method
    if method in ('ME', 'TA', 'EM'):                         # exact, one device call
        _, _, D = gp.predict_derivatives(x, u, S, values=False)
        vec_rows = lambda T, n: T.reshape(Ny * Ny, n, order='F')                    # [a, c, k] -> row a + Ny c
        dmS = D['dmean_dcov'].reshape(Ny, Nx * Nx, order='F')                        # [a, p, q] -> col p + Nx q
        dcS = D['dcov_dcov'].reshape(Ny * Ny, Nx, Nx, order='F').reshape(Ny * Ny, Nx * Nx, order='F')
        return [D['dmean_dx'], D['dmean_du'], dmS, vec_rows(D['dcov_dx'], Ny), vec_rows(D['dcov_du'], Nu), dcS]
    z = np.concatenate([x, u])

    def both(zv, Sv):
        m, c = gp.predict(zv[:Ny], zv[Ny:], Sv)
        return np.concatenate([np.array(m).reshape(-1), np.array(c).reshape(-1, order='F')])
    Jz = np.zeros((Ny + Ny * Ny, Nx))
    for k in range(Nx):
        e = np.zeros(Nx)
        e[k] = fd_eps * max(1.0, abs(z[k]))
        Jz[:, k] = (both(z + e, S) - both(z - e, S)) / (2 * e[k])
    JS = np.zeros((Ny + Ny * Ny, Nx * Nx))
    for k in range(Nx * Nx):
        E = np.zeros(Nx * Nx)
        E[k] = fd_eps
        E = E.reshape(Nx, Nx, order='F')
        JS[:, k] = (both(z, S + E) - both(z, S - E)) / (2 * fd_eps)
    return [Jz[:Ny, :Ny], Jz[:Ny, Ny:], JS[:Ny], Jz[Ny:, :Ny], Jz[Ny:, Ny:], JS[Ny:]]


def make_predict_callback(gp, name='gp_hip', fd_eps=1e-6):
    """Return a casadi.Callback instance evaluating `gp.predict` on the GPU."""
    if ca is None:
        raise ImportError('casadi is not installed; the HIP GP can still be used directly via GP.predict')

    N, Ny, Nu = gp.get_size()
    Nx = Ny + Nu

    class _Jac(ca.Callback):
        def __init__(self, opts):
            ca.Callback.__init__(self)
            self.construct(name + '_jac', opts)

        def get_n_in(self): return 5           # x, u, covar, out_mean, out_cov

        def get_n_out(self): return 6          # d{mean,cov}/d{x,u,covar}

        def get_sparsity_in(self, i):
            return [ca.Sparsity.dense(Ny, 1), ca.Sparsity.dense(Nu, 1), ca.Sparsity.dense(Nx, Nx),
                    ca.Sparsity.dense(Ny, 1), ca.Sparsity.dense(Ny, Ny)][i]

        def get_sparsity_out(self, i):
            rows = [Ny, Ny, Ny, Ny * Ny, Ny * Ny, Ny * Ny][i]
            cols = [Ny, Nu, Nx * Nx][i % 3]
            return ca.Sparsity.dense(rows, cols)

        def eval(self, arg):
            return jacobian_blocks(gp, np.array(arg[0]).reshape(-1), np.array(arg[1]).reshape(-1),
                                   np.array(arg[2]).reshape(Nx, Nx), fd_eps)

    class _Predict(ca.Callback):
        def __init__(self, opts):
            ca.Callback.__init__(self)
            self._jac = None
            self.construct(name, opts)

        def get_n_in(self): return 3

        def get_n_out(self): return 2

        def get_sparsity_in(self, i):
            return [ca.Sparsity.dense(Ny, 1), ca.Sparsity.dense(Nu, 1), ca.Sparsity.dense(Nx, Nx)][i]

        def get_sparsity_out(self, i):
            return [ca.Sparsity.dense(Ny, 1), ca.Sparsity.dense(Ny, Ny)][i]

        def eval(self, arg):
            mean, cov = gp.predict(np.array(arg[0]).reshape(-1), np.array(arg[1]).reshape(-1),
                                   np.array(arg[2]).reshape(Nx, Nx))
            return [mean, cov]

        def has_jacobian(self): return True

        def get_jacobian(self, jname, inames, onames, opts):
            self._jac = _Jac(opts)
            return self._jac

    return _Predict({'enable_fd': False})
