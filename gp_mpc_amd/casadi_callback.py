"""CasADi `Callback`s around the HIP GP predictor.

In the reference `GP.predict` returns a symbolic `ca.Function` call that is inlined into the NLP
graph (gp_class.py:207-242,259; mpc_class.py:412-413).  A GPU predictor cannot be inlined, so an
MPC layer uses it through a callback instead.  Two are provided:

  make_predict_callback(gp)            one node per call, the signature of `__predict` (gp_class.py:212-224):
                                       (x[Ny], u[Nu], covar[Nx x Nx]) -> (mean[Ny], cov[Ny x Ny]);
  make_batched_predict_callback(gp, Nt)   all Nt shooting nodes of mpc_class.py:361-423 in ONE call:
                                       (X[Ny x Nt], U[Nu x Nt], C[Nx x Nx Nt]) -> (M[Ny x Nt], V[Ny x Ny Nt]),
                                       node t in column t / column block t; its Jacobian is block diagonal over the
                                       nodes and declared as such (sparsity), so IPOPT sees Nt small dense blocks.
                                       One NLP evaluation then costs ONE device call (0.44 ms for 30 nodes at
                                       N = 8192, Ny = 6 instead of 30 x 43-64 us plus 30 Python round trips).

`get_jacobian` follows the convention of the CasADi that is installed (`casadi.__version__`):
  * 3.4 / 3.5 -- the version the reference was "tested with" (README.md:18-19): the Jacobian function takes the
    nominal inputs and outputs and returns ONE matrix, the Jacobian of all outputs stacked (each vectorised column
    major) with respect to all inputs stacked: [Ny + Ny^2, Ny + Nu + Nx^2] for the single-node callback
    (`jacobian_dense`);
  * >= 3.6: one output per (output, input) pair, `jac_<o>_<i>` in output-major order (`jacobian_blocks`).
Derivatives are exact (one device call) for 'ME', 'TA' (`gpmpc_predict_sens`) and 'EM' (`gpmpc_predict_em_sens`);
central differences of the device predictor -- mean AND covariance, so that values and derivatives stay consistent --
for the legacy 'old_ME' / 'old_TA'.  Use the callbacks with IPOPT options `expand=False` (a Callback cannot be
flattened to SX; note mpc_class.py:169 reads solver_opts['expand']) and `hessian_approximation='limited-memory'`.

casadi is not installable in the build image: the Callback classes are import-guarded and cannot be
executed here.  Everything numeric they do lives in the casadi-free functions `jacobian_blocks`, `jacobian_dense`,
`batched_values`, `batched_jacobian_blocks`, `batched_jacobian_triplets` and `batched_jacobian_dense`, which ARE tested
(tests/parity_cases.py::check_callback_blocks, check_callback_batched); the classes only move their results into
casadi's containers.
"""
try:
    import casadi as ca
except Exception:          # pragma: no cover - casadi absent in this image
    ca = None

import numpy as np


def casadi_version(version=None):
    """(major, minor) of the installed CasADi (or of a version string), None without casadi."""
    if version is None:
        if ca is None:
            return None
        version = ca.__version__
    parts = []
    for tok in str(version).split('.')[:2]:
        digits = ''.join(ch for ch in tok if ch.isdigit())
        parts.append(int(digits) if digits else 0)
    while len(parts) < 2:
        parts.append(0)
    return tuple(parts)


def jacobian_layout(version=None):
    """'dense' (CasADi <= 3.5: one stacked Jacobian) or 'blocks' (>= 3.6: one output per (output, input) pair)."""
    v = casadi_version(version)
    return 'blocks' if v is None or v >= (3, 6) else 'dense'


def _fd_blocks(gp, x, u, S, fd_eps):
    """Central differences of GP.predict in CasADi's vec layout (legacy methods)."""
    N, Ny, Nu = gp.get_size()
    Nx = Ny + Nu
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


def _exact_derivatives(gp, Nx):
    """'ME' / 'TA' / 'EM': one device call (every input dimension the library takes, d <= 16); legacy methods: differenced."""
    return gp._GP__gp_method in ('ME', 'TA', 'EM')


def _blocks_from_D(D, Ny, Nu, Nx):
    """One node's derivative dictionary (GP.predict_derivatives) -> the six blocks in CasADi's column-major vec layout."""
    vec_rows = lambda T, n: T.reshape(Ny * Ny, n, order='F')                    # [a, c, k] -> row a + Ny c
    dmS = D['dmean_dcov'].reshape(Ny, Nx * Nx, order='F')                        # [a, p, q] -> col p + Nx q
    dcS = D['dcov_dcov'].reshape(Ny * Ny, Nx, Nx, order='F').reshape(Ny * Ny, Nx * Nx, order='F')
    return [D['dmean_dx'], D['dmean_du'], dmS, vec_rows(D['dcov_dx'], Ny), vec_rows(D['dcov_du'], Nu), dcS]


def jacobian_blocks(gp, x, u, S, fd_eps=1e-6):
    """The six dense Jacobian blocks of `(x, u, covar) -> (mean, cov)` (CasADi >= 3.6: the outputs of the `get_jacobian`
    function, output-major), matrices vectorised COLUMN-major:
        [d mean/dx (Ny x Ny), d mean/du (Ny x Nu), d mean/d vec(covar) (Ny x Nx^2),
         d vec(cov)/dx (Ny^2 x Ny), d vec(cov)/du (Ny^2 x Nu), d vec(cov)/d vec(covar) (Ny^2 x Nx^2)],
    row index of vec(cov) = a + Ny c, column index of vec(covar) = p + Nx q.  Derivatives are with respect to the
    RAW x, u (chain rule through GP.predict's standardisation, gp_class.py:253-261)."""
    N, Ny, Nu = gp.get_size()
    Nx = Ny + Nu
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    u = np.asarray(u, dtype=np.float64).reshape(-1)
    S = np.asarray(S, dtype=np.float64).reshape(Nx, Nx)
    if _exact_derivatives(gp, Nx):                            # exact, one device call
        _, _, D = gp.predict_derivatives(x, u, S, values=False)
        return _blocks_from_D(D, Ny, Nu, Nx)
    return _fd_blocks(gp, x, u, S, fd_eps)


def jacobian_dense(gp, x, u, S, fd_eps=1e-6):
    """The ONE Jacobian CasADi 3.4 / 3.5 expect from `get_jacobian` (README.md:18-19 names 3.4): all outputs stacked,
    [mean; vec(cov)], against all inputs stacked, [x; u; vec(covar)] -- [Ny + Ny^2, Ny + Nu + Nx^2], the blocks of
    `jacobian_blocks` side by side."""
    b = jacobian_blocks(gp, x, u, S, fd_eps)
    return np.block([[b[0], b[1], b[2]], [b[3], b[4], b[5]]])


# ------------------------------------------------------------------------------------------------------------------
# all shooting nodes in one call
# ------------------------------------------------------------------------------------------------------------------
def _split_nodes(gp, X, U, C):
    N, Ny, Nu = gp.get_size()
    Nx = Ny + Nu
    X = np.asarray(X, dtype=np.float64).reshape(Ny, -1)
    Nt = X.shape[1]
    U = np.asarray(U, dtype=np.float64).reshape(Nu, Nt)
    C = np.asarray(C, dtype=np.float64).reshape(Nx, Nx * Nt)
    Cn = np.stack([C[:, Nx * t:Nx * (t + 1)] for t in range(Nt)])               # [Nt, Nx, Nx]
    return Ny, Nu, Nx, Nt, X.T.copy(), U.T.copy(), Cn


def batched_values(gp, X, U, C):
    """(X[Ny x Nt], U[Nu x Nt], C[Nx x Nx Nt]) -> (M[Ny x Nt], V[Ny x Ny Nt]): `GP.predict` for every node, one device call."""
    Ny, Nu, Nx, Nt, Xn, Un, Cn = _split_nodes(gp, X, U, C)
    mean, cov = gp.predict_batch(np.concatenate([Xn, Un], axis=1), Cn, standardized=False)
    return mean.T.copy(), np.concatenate([cov[t] for t in range(Nt)], axis=1)


def batched_jacobian_blocks(gp, X, U, C, fd_eps=1e-6):
    """Per node t the six blocks of `jacobian_blocks`: a list of Nt lists.  'ME' / 'TA' / 'EM': ONE device call for all
    nodes (`GP.predict_derivatives_batch`); legacy methods: central differences node by node."""
    Ny, Nu, Nx, Nt, Xn, Un, Cn = _split_nodes(gp, X, U, C)
    if _exact_derivatives(gp, Nx):
        _, _, D = gp.predict_derivatives_batch(Xn, Un, Cn, values=False)
        return [_blocks_from_D({k: v[t] for k, v in D.items()}, Ny, Nu, Nx) for t in range(Nt)]
    return [_fd_blocks(gp, Xn[t], Un[t], Cn[t], fd_eps) for t in range(Nt)]


def batched_block_sparsity(Ny, Nu, Nx, Nt):
    """Row / column index arrays (column-major vec layout of the batched signature) of the nonzeros of each of the six
    (output, input) Jacobian pairs: output M[Ny x Nt] or V[Ny x Ny Nt], input X[Ny x Nt], U[Nu x Nt] or C[Nx x Nx Nt].
    vec index of M[a, t] = a + Ny t; of V[a, Ny t + c] = a + Ny (Ny t + c); of X[p, t] = p + Ny t; of U[p, t] = p + Nu t;
    of C[p, Nx t + q] = p + Nx (Nx t + q).  Node t only depends on node t: block diagonal.  Entries are listed node by
    node, each node's dense block in column-major order (so block.reshape(-1, order='F') are its values)."""
    out = []
    nrow = [Ny, Ny * Ny]
    ncol = [Ny, Nu, Nx * Nx]
    for o in range(2):
        for i in range(3):
            r0 = np.arange(nrow[o])
            c0 = np.arange(ncol[i])
            rr, cc = np.meshgrid(r0, c0, indexing='ij')
            rows = np.concatenate([(rr + nrow[o] * t).reshape(-1, order='F') for t in range(Nt)])
            cols = np.concatenate([(cc + ncol[i] * t).reshape(-1, order='F') for t in range(Nt)])
            out.append((rows, cols, (nrow[o] * Nt, ncol[i] * Nt)))
    return out


def batched_jacobian_triplets(gp, X, U, C, fd_eps=1e-6):
    """The six (output, input) Jacobian pairs of the batched callback as (rows, cols, values, shape): block diagonal over
    the nodes, values in the order of `batched_block_sparsity` -- what the >= 3.6 `get_jacobian` function returns, one
    sparse matrix per pair."""
    Ny, Nu, Nx, Nt, _, _, _ = _split_nodes(gp, X, U, C)
    per_node = batched_jacobian_blocks(gp, X, U, C, fd_eps)
    sp = batched_block_sparsity(Ny, Nu, Nx, Nt)
    out = []
    for k, (rows, cols, shape) in enumerate(sp):
        vals = np.concatenate([per_node[t][k].reshape(-1, order='F') for t in range(Nt)])
        out.append((rows, cols, vals, shape))
    return out


def batched_jacobian_dense(gp, X, U, C, fd_eps=1e-6):
    """The single stacked Jacobian of the batched callback (CasADi 3.4 / 3.5), dense: [Ny Nt + Ny^2 Nt, Ny Nt + Nu Nt +
    Nx^2 Nt] = the six pairs of `batched_jacobian_triplets` side by side (the Callback hands it over sparse)."""
    trip = batched_jacobian_triplets(gp, X, U, C, fd_eps)
    mats = []
    for rows, cols, vals, shape in trip:
        Mx = np.zeros(shape)
        Mx[rows, cols] = vals
        mats.append(Mx)
    return np.block([[mats[0], mats[1], mats[2]], [mats[3], mats[4], mats[5]]])


# ------------------------------------------------------------------------------------------------------------------
# the casadi classes (thin: they only move the arrays above into casadi containers)
# ------------------------------------------------------------------------------------------------------------------
def _need_casadi():
    if ca is None:
        raise ImportError('casadi is not installed; the HIP GP can still be used directly via GP.predict')


def _sparsity_from(rows, cols, shape):
    return ca.Sparsity.triplet(int(shape[0]), int(shape[1]), [int(r) for r in rows], [int(c) for c in cols])


def make_predict_callback(gp, name='gp_hip', fd_eps=1e-6, layout=None):
    """A casadi.Callback evaluating `gp.predict` on the GPU, one node per call.  layout: 'dense' | 'blocks' | None (from
    the installed CasADi's version, `jacobian_layout`)."""
    _need_casadi()
    layout = layout or jacobian_layout()
    N, Ny, Nu = gp.get_size()
    Nx = Ny + Nu
    sp_in = lambda: [ca.Sparsity.dense(Ny, 1), ca.Sparsity.dense(Nu, 1), ca.Sparsity.dense(Nx, Nx)]
    sp_out = lambda: [ca.Sparsity.dense(Ny, 1), ca.Sparsity.dense(Ny, Ny)]

    class _Jac(ca.Callback):
        def __init__(self, jname, opts):
            ca.Callback.__init__(self)
            self.construct(jname, opts)

        def get_n_in(self): return 5           # x, u, covar, out_mean, out_cov

        def get_n_out(self): return 1 if layout == 'dense' else 6

        def get_sparsity_in(self, i): return (sp_in() + sp_out())[i]

        def get_sparsity_out(self, i):
            if layout == 'dense':
                return ca.Sparsity.dense(Ny + Ny * Ny, Ny + Nu + Nx * Nx)
            rows = [Ny, Ny, Ny, Ny * Ny, Ny * Ny, Ny * Ny][i]
            cols = [Ny, Nu, Nx * Nx][i % 3]
            return ca.Sparsity.dense(rows, cols)

        def eval(self, arg):
            a = (np.array(arg[0]).reshape(-1), np.array(arg[1]).reshape(-1), np.array(arg[2]).reshape(Nx, Nx))
            if layout == 'dense':
                return [jacobian_dense(gp, *a, fd_eps)]
            return jacobian_blocks(gp, *a, fd_eps)

    class _Predict(ca.Callback):
        def __init__(self, opts):
            ca.Callback.__init__(self)
            self._jac = None
            self.construct(name, opts)

        def get_n_in(self): return 3

        def get_n_out(self): return 2

        def get_sparsity_in(self, i): return sp_in()[i]

        def get_sparsity_out(self, i): return sp_out()[i]

        def eval(self, arg):
            mean, cov = gp.predict(np.array(arg[0]).reshape(-1), np.array(arg[1]).reshape(-1),
                                   np.array(arg[2]).reshape(Nx, Nx))
            return [mean, cov]

        def has_jacobian(self): return True

        def get_jacobian(self, jname, inames, onames, opts):
            self._jac = _Jac(jname, opts)
            return self._jac

    return _Predict({'enable_fd': False})


def make_batched_predict_callback(gp, Nt, name='gp_hip_nodes', fd_eps=1e-6, layout=None):
    """A casadi.Callback serving ALL Nt shooting nodes in one call: (X[Ny x Nt], U[Nu x Nt], C[Nx x Nx Nt]) ->
    (M[Ny x Nt], V[Ny x Ny Nt]); the Jacobian is block diagonal over the nodes and handed over with that sparsity."""
    _need_casadi()
    layout = layout or jacobian_layout()
    N, Ny, Nu = gp.get_size()
    Nx = Ny + Nu
    sp_in = lambda: [ca.Sparsity.dense(Ny, Nt), ca.Sparsity.dense(Nu, Nt), ca.Sparsity.dense(Nx, Nx * Nt)]
    sp_out = lambda: [ca.Sparsity.dense(Ny, Nt), ca.Sparsity.dense(Ny, Ny * Nt)]
    pairs = batched_block_sparsity(Ny, Nu, Nx, Nt)

    def dense_sparsity():
        """the six pairs side by side: rows offset by the outputs above, columns by the inputs to the left"""
        roff = [0, 0, 0, Ny * Nt, Ny * Nt, Ny * Nt]
        coff = [0, Ny * Nt, Ny * Nt + Nu * Nt] * 2
        rows = np.concatenate([p[0] + roff[k] for k, p in enumerate(pairs)])
        cols = np.concatenate([p[1] + coff[k] for k, p in enumerate(pairs)])
        return rows, cols, (Ny * Nt + Ny * Ny * Nt, Ny * Nt + Nu * Nt + Nx * Nx * Nt)

    class _Jac(ca.Callback):
        def __init__(self, jname, opts):
            ca.Callback.__init__(self)
            self.construct(jname, opts)

        def get_n_in(self): return 5

        def get_n_out(self): return 1 if layout == 'dense' else 6

        def get_sparsity_in(self, i): return (sp_in() + sp_out())[i]

        def get_sparsity_out(self, i):
            if layout == 'dense':
                return _sparsity_from(*dense_sparsity())
            return _sparsity_from(*pairs[i])

        def eval(self, arg):
            trip = batched_jacobian_triplets(gp, np.array(arg[0]), np.array(arg[1]), np.array(arg[2]), fd_eps)
            if layout == 'dense':
                rows, cols, shape = dense_sparsity()
                vals = np.concatenate([t[2] for t in trip])
                return [ca.DM.triplet([int(r) for r in rows], [int(c) for c in cols], ca.DM(vals), int(shape[0]), int(shape[1]))]
            return [ca.DM.triplet([int(r) for r in t[0]], [int(c) for c in t[1]], ca.DM(t[2]), int(t[3][0]), int(t[3][1]))
                    for t in trip]

    class _Predict(ca.Callback):
        def __init__(self, opts):
            ca.Callback.__init__(self)
            self._jac = None
            self.construct(name, opts)

        def get_n_in(self): return 3

        def get_n_out(self): return 2

        def get_sparsity_in(self, i): return sp_in()[i]

        def get_sparsity_out(self, i): return sp_out()[i]

        def eval(self, arg):
            M, V = batched_values(gp, np.array(arg[0]), np.array(arg[1]), np.array(arg[2]))
            return [M, V]

        def has_jacobian(self): return True

        def get_jacobian(self, jname, inames, onames, opts):
            self._jac = _Jac(jname, opts)
            return self._jac

    return _Predict({'enable_fd': False})
