"""A stand-in for the part of CasADi's Python API that a `casadi.Callback` touches -- TEST INFRASTRUCTURE.

casadi is not installable in the build image, so the Callback classes of
gp_mpc_amd/casadi_callback.py could never execute.  This module restates the protocol CasADi drives a
Callback through (casadi/core/callback.hpp, function.hpp; the reference uses it implicitly at
gp_class.py:212-224 / mpc_class.py:412-413) so that every line of those classes runs in the CPU tier and on the GPU:

  * `Callback.construct(name, opts)` asks the subclass for get_n_in / get_n_out, get_sparsity_in(i) / get_sparsity_out(i)
    (cached: the object is immutable afterwards) and has_jacobian();
  * calling the function converts every argument to a `DM` OF THE DECLARED SPARSITY (a shape mismatch raises, as
    `Function::call` does), hands `eval` the list, and converts what comes back to `DM`s that must fit the declared
    output sparsities; a 1-D numpy array becomes a column, as in CasADi's numpy typemap;
  * `.jacobian()` calls `get_jacobian(name, inames, onames, opts)` with CasADi's naming -- inputs
    (i0.., out_o0..), and, depending on `__version__`, ONE output `jac` holding the Jacobian of all outputs stacked
    (vectorised column-major) against all inputs stacked (3.4 / 3.5: the reference's CasADi, README.md:18-19), or one
    output `jac_<o>_<i>` per pair in output-major order (>= 3.6) -- and checks the returned function's signature;
  * `Sparsity.dense / Sparsity.triplet`, `DM(...)`, `DM.triplet(rows, cols, values, nrow, ncol)`, `numpy.array(DM)` (dense,
    2-D), all column-major.

It is not a CasADi re-implementation: no symbolics, no solvers.  Only tests import it; the package imports the real
`casadi` (and works without one)."""
import numpy as np

__version__ = '3.6.3'


def set_version(v):
    global __version__
    __version__ = v


def _version_tuple():
    a, b = __version__.split('.')[:2]
    return int(a), int(''.join(ch for ch in b if ch.isdigit()))


class Sparsity:
    """Compressed-column pattern; entries ordered column-major like CasADi's."""

    def __init__(self, nrow, ncol, rows, cols):
        self.nrow, self.ncol = int(nrow), int(ncol)
        rc = sorted(set(zip((int(c) for c in cols), (int(r) for r in rows))))       # (col, row): column-major order
        if len(rc) != len(list(rows)):
            raise RuntimeError('Sparsity.triplet: duplicate entries')
        for c, r in rc:
            if not (0 <= r < self.nrow and 0 <= c < self.ncol):
                raise RuntimeError('Sparsity.triplet: entry (%d, %d) outside %d x %d' % (r, c, self.nrow, self.ncol))
        self._rows = np.array([r for c, r in rc], dtype=int)
        self._cols = np.array([c for c, r in rc], dtype=int)

    @staticmethod
    def dense(nrow, ncol=1):
        nrow, ncol = int(nrow), int(ncol)
        rr, cc = np.meshgrid(np.arange(nrow), np.arange(ncol), indexing='ij')
        return Sparsity(nrow, ncol, rr.reshape(-1, order='F'), cc.reshape(-1, order='F'))

    @staticmethod
    def triplet(nrow, ncol, rows, cols):
        if not all(isinstance(v, int) for v in list(rows)[:8] + list(cols)[:8]):
            raise TypeError('Sparsity.triplet takes lists of int (SWIG typemap)')
        return Sparsity(nrow, ncol, rows, cols)

    def size1(self): return self.nrow

    def size2(self): return self.ncol

    def nnz(self): return len(self._rows)

    def is_dense(self): return self.nnz() == self.nrow * self.ncol

    def get_triplet(self): return list(self._rows), list(self._cols)

    @property
    def shape(self): return (self.nrow, self.ncol)

    def __eq__(self, other):
        return (isinstance(other, Sparsity) and self.shape == other.shape and np.array_equal(self._rows, other._rows)
                and np.array_equal(self._cols, other._cols))


class DM:
    """Numeric matrix with a sparsity pattern (dense storage underneath; structural zeros tracked by the pattern)."""

    def __init__(self, x=None, sp=None):
        if isinstance(x, DM):
            self._a, self._sp = x._a.copy(), x._sp
            return
        a = np.array(x if x is not None else np.zeros((0, 0)), dtype=np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)                       # CasADi's numpy typemap: a 1-D array is a column vector
        elif a.ndim != 2:
            raise TypeError('DM: arrays of more than two dimensions are not accepted')
        self._a = a
        self._sp = sp if sp is not None else Sparsity.dense(*a.shape)

    @staticmethod
    def triplet(rows, cols, values, nrow, ncol):
        if not isinstance(values, DM):
            raise TypeError('DM.triplet: values must be a DM')
        v = values._a.reshape(-1, order='F')
        if not (len(rows) == len(cols) == len(v)):
            raise RuntimeError('DM.triplet: rows, cols and values differ in length')
        a = np.zeros((int(nrow), int(ncol)))
        for r, c, x in zip(rows, cols, v):            # duplicates add up, as in CasADi
            a[int(r), int(c)] += x
        return DM(a, Sparsity.triplet(int(nrow), int(ncol), [int(r) for r in rows], [int(c) for c in cols]))

    def sparsity(self): return self._sp

    def full(self): return self._a.copy()

    def nnz(self): return self._sp.nnz()

    @property
    def shape(self): return self._a.shape

    def __array__(self, dtype=None, copy=None):
        return self._a.copy() if dtype is None else self._a.astype(dtype)

    def __float__(self):
        if self._a.size != 1:
            raise TypeError('only 1 x 1 DM converts to float')
        return float(self._a[0, 0])


def _to_dm(x, sp, what):
    """Function::call's conversion of an argument / result to the declared sparsity."""
    d = x if isinstance(x, DM) else DM(x)
    if d.shape != sp.shape:
        if d.shape == (1, 1):                          # scalars broadcast
            d = DM(np.full(sp.shape, float(d)))
        elif d.shape == (sp.shape[1], sp.shape[0]) and 1 in sp.shape:
            d = DM(d._a.T)                             # a row where a column is declared: CasADi transposes vectors
        else:
            raise RuntimeError('%s: dimension mismatch, expected %s, got %s' % (what, sp.shape, d.shape))
    # entries outside the declared pattern must be structural zeros of the callee; project onto the pattern
    keep = np.zeros(sp.shape, dtype=bool)
    keep[sp._rows, sp._cols] = True
    if np.any(d._a[~keep] != 0.0):
        raise RuntimeError('%s: nonzero outside the declared sparsity pattern' % what)
    return DM(d._a, sp)


class Function:
    """What `Callback.construct` turns the object into: a callable with named, typed inputs and outputs."""

    def _finalize(self, name, n_in, n_out, sp_in, sp_out, name_in=None, name_out=None):
        self._name, self._n_in, self._n_out, self._sp_in, self._sp_out = name, n_in, n_out, sp_in, sp_out
        self._name_in = name_in or ['i%d' % i for i in range(n_in)]
        self._name_out = name_out or ['o%d' % i for i in range(n_out)]

    def name(self): return self._name

    def n_in(self): return self._n_in

    def n_out(self): return self._n_out

    def sparsity_in(self, i): return self._sp_in[i]

    def sparsity_out(self, i): return self._sp_out[i]

    def size_in(self, i): return self._sp_in[i].shape

    def size_out(self, i): return self._sp_out[i].shape

    def name_in(self, i=None): return list(self._name_in) if i is None else self._name_in[i]

    def name_out(self, i=None): return list(self._name_out) if i is None else self._name_out[i]


class Callback(Function):
    def __init__(self):
        self._constructed = False
        self.n_eval = 0

    # defaults a subclass may leave alone (callback.hpp)
    def init(self): pass

    def get_n_in(self): return 1

    def get_n_out(self): return 1

    def get_sparsity_in(self, i): return Sparsity.dense(1, 1)

    def get_sparsity_out(self, i): return Sparsity.dense(1, 1)

    def has_jacobian(self): return False

    def get_jacobian(self, name, inames, onames, opts):
        raise RuntimeError('get_jacobian not defined')

    def eval(self, arg):
        raise RuntimeError('eval not defined')

    def construct(self, name, opts=None):
        if self._constructed:
            raise RuntimeError('Callback.construct called twice')
        if not isinstance(name, str) or not name.isidentifier():
            raise RuntimeError('Function name "%s" is not a valid identifier' % name)
        self._opts = dict(opts or {})
        self.init()
        n_in, n_out = int(self.get_n_in()), int(self.get_n_out())
        sp_in = [self.get_sparsity_in(i) for i in range(n_in)]
        sp_out = [self.get_sparsity_out(i) for i in range(n_out)]
        for s in sp_in + sp_out:
            if not isinstance(s, Sparsity):
                raise TypeError('get_sparsity_* must return a Sparsity')
        self._finalize(name, n_in, n_out, sp_in, sp_out)
        self._has_jac = bool(self.has_jacobian())
        self._jac_fn = None
        self._constructed = True

    def __call__(self, *args):
        if not self._constructed:
            raise RuntimeError('Callback used before construct()')
        if len(args) != self._n_in:
            raise RuntimeError('%s: %d inputs expected, %d given' % (self._name, self._n_in, len(args)))
        arg = [_to_dm(a, self._sp_in[i], '%s input %d' % (self._name, i)) for i, a in enumerate(args)]
        res = self.eval(arg)
        self.n_eval += 1
        if not isinstance(res, (list, tuple)) or len(res) != self._n_out:
            raise RuntimeError('%s: eval must return a list of %d outputs' % (self._name, self._n_out))
        out = [_to_dm(r, self._sp_out[i], '%s output %d' % (self._name, i)) for i, r in enumerate(res)]
        return out[0] if self._n_out == 1 else tuple(out)

    def jacobian(self):
        """Function::jacobian(): the derivative function the NLP solver asks for, with the version's signature."""
        if not self._has_jac:
            raise RuntimeError('%s: no Jacobian available (has_jacobian() is false and enable_fd is off)' % self._name)
        if self._jac_fn is None:
            inames = list(self._name_in) + ['out_' + o for o in self._name_out]
            if _version_tuple() >= (3, 6):
                onames = ['jac_%s_%s' % (o, i) for o in self._name_out for i in self._name_in]
                want = [(self._sp_out[o].nnz() if False else self._sp_out[o].nrow * self._sp_out[o].ncol,
                         self._sp_in[i].nrow * self._sp_in[i].ncol) for o in range(self._n_out) for i in range(self._n_in)]
            else:
                onames = ['jac']
                want = [(sum(s.nrow * s.ncol for s in self._sp_out), sum(s.nrow * s.ncol for s in self._sp_in))]
            J = self.get_jacobian('jac_' + self._name, inames, onames, dict(self._opts))
            if not isinstance(J, Function) or not getattr(J, '_constructed', False):
                raise RuntimeError('get_jacobian must return a constructed Function')
            if J.n_in() != self._n_in + self._n_out or J.n_out() != len(onames):
                raise RuntimeError('jacobian function of %s: %d inputs / %d outputs, expected %d / %d'
                                   % (self._name, J.n_in(), J.n_out(), self._n_in + self._n_out, len(onames)))
            for k, s in enumerate(self._sp_in + self._sp_out):
                if J.sparsity_in(k).shape != s.shape:
                    raise RuntimeError('jacobian function input %d has shape %s, expected %s' % (k, J.sparsity_in(k).shape, s.shape))
            for k, w in enumerate(want):
                if J.sparsity_out(k).shape != w:
                    raise RuntimeError('jacobian function output %d (%s) has shape %s, expected %s'
                                       % (k, onames[k], J.sparsity_out(k).shape, w))
            J._name_in, J._name_out = inames, onames
            self._jac_fn = J
        return self._jac_fn
