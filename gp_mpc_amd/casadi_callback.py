"""CasADi `Callback` around the HIP GP predictor.

In the reference `GP.predict` returns a symbolic `ca.Function` call that is inlined into the NLP
graph (gp_class.py:207-242,259; mpc_class.py:412-413).  A GPU predictor cannot be inlined, so an
MPC layer uses it through this callback instead: same signature `(x[Ny], u[Nu], covar[Nx x Nx])
-> (mean[Ny], cov[Ny x Ny])` as `__predict` (gp_class.py:212-224), with `get_jacobian` served for the
'ME' and 'TA' methods by exact derivatives from one device call (`GP.predict_derivatives` ->
`gpmpc_predict_sens`: mean Jacobian, mean Hessian and variance gradient kernels), and for the other
methods by the analytic mean Jacobian plus central differences of the device predictor for the
covariance block.  Use it with IPOPT options `expand=False` (a Callback cannot be flattened to SX;
note mpc_class.py:169 reads solver_opts['expand']) and `hessian_approximation='limited-memory'`.

casadi is not installable in the build image, so this module is import-guarded and
cannot be exercised by the test-suite here; it only composes entry points that are tested
(`GP.predict`, `GP.predict_derivatives`, `GP.discrete_linearize`).
"""
try:
    import casadi as ca
except Exception:          # pragma: no cover - casadi absent in this image
    ca = None

import numpy as np


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
            x = np.array(arg[0]).reshape(-1)
            u = np.array(arg[1]).reshape(-1)
            S = np.array(arg[2]).reshape(Nx, Nx)
            if gp._GP__gp_method in ('ME', 'TA'):                # exact, one device call
                _, _, D = gp.predict_derivatives(x, u, S)
                col = lambda T, n: T.reshape(Ny * Ny, n, order='F') if T.ndim == 3 else T   # vec(cov) is column-major
                dcS = D['dcov_dcov'].reshape(Ny * Ny, Nx, Nx, order='F').reshape(Ny * Ny, Nx * Nx, order='F')
                return [D['dmean_dx'], D['dmean_du'], np.zeros((Ny, Nx * Nx)),
                        col(D['dcov_dx'], Ny), col(D['dcov_du'], Nu), dcS]
            A, B = gp.discrete_linearize(x, u, S)              # analytic, on the device
            if gp._GP__normalize:                                # d mean_raw / d x_raw
                A = A * gp._GP__stdY[:, None] / gp._GP__stdX[None, :]
                B = B * gp._GP__stdY[:, None] / gp._GP__stdU[None, :]
            z = np.concatenate([x, u])

            def cov_of(zv, Sv):
                return np.array(gp.predict(zv[:Ny], zv[Ny:], Sv)[1]).reshape(-1, order='F')
            Jz = np.zeros((Ny * Ny, Nx))
            for k in range(Nx):
                e = np.zeros(Nx)
                e[k] = fd_eps * max(1.0, abs(z[k]))
                Jz[:, k] = (cov_of(z + e, S) - cov_of(z - e, S)) / (2 * e[k])
            JS = np.zeros((Ny * Ny, Nx * Nx))
            for k in range(Nx * Nx):
                E = np.zeros(Nx * Nx)
                E[k] = fd_eps
                E = E.reshape(Nx, Nx, order='F')
                JS[:, k] = (cov_of(z, S + E) - cov_of(z, S - E)) / (2 * fd_eps)
            return [A, B, np.zeros((Ny, Nx * Nx)), Jz[:, :Ny], Jz[:, Ny:], JS]

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
