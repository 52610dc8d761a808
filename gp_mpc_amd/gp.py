"""`GP`: host-side mirror of the reference's `gp_mpc.GP` class (gp_class.py:20-861) for the hot path.

Same constructor, method names, argument meaning and error behaviour as the reference, so that a
caller such as `mpc_class.MPC` (mpc_class.py:167,234,412-413,596) can use it unchanged; all GP
arithmetic runs in the HIP kernels behind the C ABI (include/gpmpc.h) -- there is no CPU path.
Differences that are deliberate are marked "DIFF".

Standardisation stays on the host exactly as in the reference: `predict` standardises x and u,
calls the (device) predictor in the GP's own units and un-standardises the MEAN only
(gp_class.py:245-263); `discrete_linearize` returns Jacobians in standardised coordinates
(gp_class.py:647-661).
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import _lib
from .train import train_gp

METHODS = ('ME', 'TA', 'EM', 'old_ME', 'old_TA')


def _listify(v):
    """numpy arrays -> nested lists, recursively through dicts (json.dump input)."""
    if isinstance(v, dict):
        return {k: _listify(x) for k, x in v.items()}
    if isinstance(v, np.ndarray):
        return v.tolist()
    return v


class GP:
    def __init__(self, X, Y, mean_func="zero", gp_method="TA",
                 optimizer_opts=None, hyper=None, normalize=True, multistart=1,
                 xlb=None, xub=None, ulb=None, uub=None, meta=None,
                 optimize_nummeric=True, device=0, lib=None, predict_adds_mean=False, optimizer='native'):
        """Initialize and optimize GP model (gp_class.py:21-75).

        Extra arguments: `device` (GPU ordinal), `lib` (a loaded `GpmpcLib`; default: the in-tree
        libgpmpc_hip.so, raising if it is missing) and `predict_adds_mean`: the reference builds its predictor
        WITHOUT the mean function (`build_gp(...)` is called with its default meanFunc='zero', gp_class.py:68-71),
        so a model trained with mean_func != 'zero' predicts ks^T alpha only; that is the default here too.
        `optimizer`: 'native' (default) or 'scipy', see gp_mpc_amd.train.train_gp.
        True gives build_gp(..., meanFunc=mean_func) (gp_functions.py:131,135): mean(z) = ks^T alpha + m(z)."""
        self._lib = lib if lib is not None else _lib.get_lib()
        self._device = device
        X = np.array(X, dtype=np.float64).copy()
        Y = np.array(Y, dtype=np.float64).copy()
        self.__X = X
        self.__Y = Y
        self.__Ny = Y.shape[1]
        self.__Nx = X.shape[1]
        self.__N = X.shape[0]
        self.__Nu = self.__Nx - self.__Ny
        self.__gp_method = gp_method
        self.__mean_func = mean_func
        self.__normalize = normalize
        self._predict_adds_mean = bool(predict_adds_mean)
        self._h = None
        self._check_mean_func(mean_func)

        if meta is not None:
            self.__meanY = np.array(meta['meanY'])
            self.__stdY = np.array(meta['stdY'])
            self.__meanZ = np.array(meta['meanZ'])
            self.__stdZ = np.array(meta['stdZ'])
            self.__meanX = np.array(meta['meanX'])
            self.__stdX = np.array(meta['stdX'])
            self.__meanU = np.array(meta['meanU'])
            self.__stdU = np.array(meta['stdU'])
        if xlb is not None:
            self.__xlb, self.__xub = np.array(xlb), np.array(xub)
            self.__ulb, self.__uub = np.array(ulb), np.array(uub)

        if hyper is None:
            self.optimize(X=X, Y=Y, opts=optimizer_opts, mean_func=mean_func,
                          xlb=xlb, xub=xub, ulb=ulb, uub=uub,
                          multistart=multistart, normalize=normalize,
                          optimize_nummeric=optimize_nummeric, optimizer=optimizer)
        else:
            # load_model branch (gp_class.py:58-66): stored X is already standardised
            self.__hyper = np.array(hyper['hyper'], dtype=np.float64)
            self.__hyper_length_scales = self.__hyper[:, :self.__Nx]
            self.__hyper_signal_variance = self.__hyper[:, self.__Nx] ** 2
            self.__hyper_noise_variance = self.__hyper[:, self.__Nx + 1] ** 2
            self.__hyper_mean = self.__hyper[:, (self.__Nx + 1):]
            self._new_handle()
            self._h.set_mean_func(mean_func, self._predict_adds_mean)
            self._h.set_factors(self.__hyper, np.array(hyper['chol'], dtype=np.float64),
                                None if hyper.get('alpha') is None else np.array(hyper['alpha'], dtype=np.float64),
                                None if hyper.get('invK') is None else np.array(hyper['invK'], dtype=np.float64))
        self.set_method(gp_method)

    # ------------------------------------------------------------------ internals
    @staticmethod
    def _check_mean_func(mean_func):
        if mean_func not in ('zero', 'const', 'linear', 'polynomial'):
            raise NameError('No mean function called: ' + str(mean_func))      # gp_functions.py:67

    def _new_handle(self):
        if self._h is not None:
            self._h.close()
        self._h = _lib.Handle(self._lib, self.__X, self.__Y, device=self._device)

    def _refit(self, want_invK=False):
        self._new_handle()
        self._h.set_mean_func(self.__mean_func, self._predict_adds_mean)
        self._h.fit(self.__hyper, want_invK=want_invK)

    @property
    def handle(self):
        """The C-ABI handle (for batched / device-pointer use)."""
        return self._h

    # ------------------------------------------------------------------ training
    def optimize(self, X=None, Y=None, opts=None, mean_func='zero',
                 xlb=None, xub=None, ulb=None, uub=None,
                 multistart=1, normalize=True, warm_start=False,
                 optimize_nummeric=True, random_restarts=False, seed=1234, gradient='analytic', optimizer='native'):
        """Optimize hyper-parameters (gp_class.py:78-142).  DIFF: both of the reference's optimiser
        back-ends (scipy SLSQP with finite differences / CasADi+IPOPT) are replaced by one driver
        (`gp_mpc_amd.train.train_gp`) that evaluates the NLL and its analytic gradient on the GPU;
        `optimize_nummeric` selects the reference's bound/initialisation convention of the
        corresponding path (True: optimize.py:434-449, False: optimize.py:207-229)."""
        self._check_mean_func(mean_func)
        self.__mean_func = mean_func
        self.__normalize = normalize

        if normalize and X is not None:
            self.__xlb = np.array(xlb)
            self.__xub = np.array(xub)
            self.__ulb = np.array(ulb)
            self.__uub = np.array(uub)
            self.__meanY = np.mean(Y, 0)
            self.__stdY = np.std(Y, 0)
            self.__meanZ = np.mean(X, 0)
            self.__stdZ = np.std(X, 0)
            self.__meanX = np.mean(X[:, :self.__Ny], 0)
            self.__stdX = np.std(X[:, :self.__Ny], 0)
            self.__meanU = np.mean(X[:, self.__Ny:], 0)
            self.__stdU = np.std(X[:, self.__Ny:], 0)

        if X is not None:
            X = np.array(X, dtype=np.float64).copy()
            self.__X = self.standardize(X, self.__meanZ, self.__stdZ) if normalize else X.copy()
        if Y is not None:
            Y = np.array(Y, dtype=np.float64).copy()
            self.__Y = self.standardize(Y, self.__meanY, self.__stdY) if (normalize and X is not None) else Y.copy()
        self.__N = self.__X.shape[0]

        hyp_init = self.__hyper if warm_start else None
        self._new_handle()
        opt = train_gp(self._h, self.__X, self.__Y, multistart=multistart, hyper_init=hyp_init,
                       optimizer_opts=opts, numpy_path_conventions=optimize_nummeric,
                       random_restarts=random_restarts, seed=seed, gradient=gradient,
                       mean_func=mean_func, predict_adds_mean=self._predict_adds_mean, optimizer=optimizer)
        self.__hyper = opt['hyper']
        self.__lam_x = opt['lam_x']
        self.__hyper_length_scales = self.__hyper[:, :self.__Nx]
        self.__hyper_signal_variance = self.__hyper[:, self.__Nx] ** 2
        self.__hyper_noise_variance = self.__hyper[:, self.__Nx + 1] ** 2
        self.__hyper_mean = self.__hyper[:, (self.__Nx + 1):]      # off-by-one slice kept (gp_class.py:142)
        self.train_info = opt

    # ------------------------------------------------------------------ validation
    def validate(self, X_test, Y_test, verbose=True):
        """gp_class.py:145-190: SMSE (divides by std, :166) and MNLP; all test rows in ONE batched
        device call instead of the reference's per-row loop."""
        Y_test = np.array(Y_test, dtype=np.float64).copy()
        X_test = np.array(X_test, dtype=np.float64).copy()
        if self.__normalize:
            Y_test = self.standardize(Y_test, self.__meanY, self.__stdY)
            X_test = self.standardize(X_test, self.__meanZ, self.__stdZ)
        N, Ny = Y_test.shape
        mean, var = self._h.predict_mean_var(X_test)
        var = var + self.noise_variance()
        loss = np.sum((Y_test - mean) ** 2, axis=0) / N
        NLP = np.sum(0.5 * np.log(2 * np.pi * var) + (Y_test - mean) ** 2 / (2 * var), axis=0)
        SMSE = loss / np.std(Y_test, 0)
        MNLP = NLP / N
        if verbose:
            print('\n________________________________________')
            print('# Validation of GP model ')
            print('----------------------------------------')
            print('* Num training samples: ' + str(self.__N))
            print('* Num test samples: ' + str(N))
            for name, arr in (('Mean squared error', loss), ('Standardized mean squared error', SMSE),
                              ('Mean Negative log Probability', MNLP)):
                print('----------------------------------------')
                print('* %s:' % name)
                for i in range(Ny):
                    print('\t- State %d: %f' % (i + 1, arr[i]))
            print('----------------------------------------\n')
        self.__SMSE = np.max(SMSE)
        return np.array(SMSE).flatten(), np.array(MNLP).flatten()

    # ------------------------------------------------------------------ prediction
    def set_method(self, gp_method='TA'):
        """Select which GP function to use (gp_class.py:193-242).  DIFF: strings are compared with
        `==` (the reference's `is` only works through CPython string interning)."""
        if gp_method not in METHODS:
            raise NameError('No GP method called: ' + str(gp_method))          # gp_class.py:237
        self.__gp_method = gp_method

    def predict(self, x, u, cov):
        """Predict future state (gp_class.py:245-263): x (Ny), u (Nu), cov of z=[x,u] (Nx x Nx)
        -> mean (Ny x 1), cov (Ny x Ny, standardised units)."""
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        u = np.asarray(u, dtype=np.float64).reshape(-1)
        if self.__normalize:
            x = self.standardize(x, self.__meanX, self.__stdX)
            u = self.standardize(u, self.__meanU, self.__stdU)
        z = np.concatenate([x, u]).reshape(1, self.__Nx)
        S = np.asarray(cov, dtype=np.float64).reshape(1, self.__Nx, self.__Nx)
        mean, c = self._h.predict(self.__gp_method, z, S)
        mean = mean[0]
        if self.__normalize:
            mean = self.inverse_mean(mean, self.__meanY, self.__stdY)
        return mean.reshape(self.__Ny, 1), c[0]

    def predict_derivatives(self, x, u, cov, values=True):
        """`predict` plus the exact first derivatives of both outputs with respect to all three inputs,
        for the 'ME', 'TA' and 'EM' methods -- what a casadi Callback standing in for `__predict`
        (gp_class.py:212-224) must provide through `get_jacobian`; the reference gets them from CasADi's
        AD of build_gp / build_TA_cov / gp_exact_moment (gp_functions.py:114-173,344-418).  One device call
        (`gpmpc_predict_sens`: mean, var, J, d2 mean/dz2, d var/dz, closed-form assembly here; 'EM':
        `gpmpc_predict_em_sens`, the Jacobians themselves).

        Returns (mean[Ny,1], cov[Ny,Ny], D) with D a dict of
          'dmean_dx' [Ny,Ny], 'dmean_du' [Ny,Nu], 'dmean_dcov' [Ny,Nx,Nx] (zero for ME / TA),
          'dcov_dx' [Ny,Ny,Ny], 'dcov_du' [Ny,Ny,Nu], 'dcov_dcov' [Ny,Ny,Nx,Nx],
        all with respect to the RAW x, u (the chain rule through the standardisation is applied; cov
        stays in standardised units exactly as `predict` returns it, gp_class.py:262).
        values=False (a Jacobian callback): the returned cov is None for 'EM' -- its pair sums are a quarter
        of the device call and the derivatives do not need them."""
        if self.__gp_method not in ('ME', 'TA', 'EM'):
            raise NotImplementedError("analytic derivatives exist for 'ME', 'TA' and 'EM'; use finite differences of "
                                      "GP.predict for '%s'" % self.__gp_method)
        Ny, Nx, Nu = self.__Ny, self.__Nx, self.__Nu
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        u = np.asarray(u, dtype=np.float64).reshape(-1)
        if self.__normalize:
            x = self.standardize(x, self.__meanX, self.__stdX)
            u = self.standardize(u, self.__meanU, self.__stdU)
        z = np.concatenate([x, u]).reshape(1, Nx)
        S = np.asarray(cov, dtype=np.float64).reshape(Nx, Nx)
        if self.__gp_method == 'EM':
            mean, c, dmean, dmS, dcz, dcS = (None if a is None else a[0]
                                             for a in self._h.predict_em_sens(z, S.reshape(1, Nx, Nx), want_cov=values))
        else:
            mean, var, J, Hm, dvar = (a[0] for a in self._h.predict_sens(z))
            c = np.diag(var)
            dcz = np.zeros((Ny, Ny, Nx))
            dcz[np.arange(Ny), np.arange(Ny)] = dvar                      # d diag(var) / dz
            dcS = np.zeros((Ny, Ny, Nx, Nx))
            dmS = np.zeros((Ny, Nx, Nx))
            if self.__gp_method == 'TA':                                  # cov = diag(var) + J S J^T
                c = c + J @ S @ J.T
                dcz += np.einsum('adp,de,ce->acp', Hm, S, J) + np.einsum('ad,de,cep->acp', J, S, Hm)
                dcS = np.einsum('ad,ce->acde', J, J)
            dmean = J.copy()
        if self.__normalize:
            mean = self.inverse_mean(mean, self.__meanY, self.__stdY)
            sz = np.concatenate([np.atleast_1d(self.__stdX), np.atleast_1d(self.__stdU)])
            dmean = dmean * np.atleast_1d(self.__stdY)[:, None] / sz[None, :]
            dmS = dmS * np.atleast_1d(self.__stdY)[:, None, None]
            dcz = dcz / sz[None, None, :]
        D = {'dmean_dx': dmean[:, :Ny], 'dmean_du': dmean[:, Ny:], 'dmean_dcov': dmS,
             'dcov_dx': dcz[:, :, :Ny], 'dcov_du': dcz[:, :, Ny:], 'dcov_dcov': dcS}
        return mean.reshape(Ny, 1), c, D

    def predict_derivatives_batch(self, X, U, C, values=True):
        """`predict_derivatives` for B shooting nodes in ONE device call (the pattern an NLP evaluation produces: IPOPT
        asks for the Jacobian of all Nt continuity constraints at once, mpc_class.py:361-423).  X[B,Ny], U[B,Nu],
        C[B,Nx,Nx] -> mean[B,Ny], cov[B,Ny,Ny] (None for 'EM' with values=False) and D as in `predict_derivatives` with
        a leading node axis.  'ME' / 'TA' / 'EM' only."""
        if self.__gp_method not in ('ME', 'TA', 'EM'):
            raise NotImplementedError("analytic derivatives exist for 'ME', 'TA' and 'EM'")
        Ny, Nx, Nu = self.__Ny, self.__Nx, self.__Nu
        X = np.asarray(X, dtype=np.float64).reshape(-1, Ny)
        U = np.asarray(U, dtype=np.float64).reshape(-1, Nu)
        B = X.shape[0]
        if self.__normalize:
            X = self.standardize(X, self.__meanX, self.__stdX)
            U = self.standardize(U, self.__meanU, self.__stdU)
        Z = np.concatenate([X, U], axis=1)
        S = np.asarray(C, dtype=np.float64).reshape(B, Nx, Nx)
        if self.__gp_method == 'EM':
            mean, c, dmean, dmS, dcz, dcS = self._h.predict_em_sens(Z, S, want_cov=values)
        else:
            mean, var, J, Hm, dvar = self._h.predict_sens(Z)
            ia = np.arange(Ny)
            c = np.zeros((B, Ny, Ny))
            c[:, ia, ia] = var
            dcz = np.zeros((B, Ny, Ny, Nx))
            dcz[:, ia, ia] = dvar
            dcS = np.zeros((B, Ny, Ny, Nx, Nx))
            dmS = np.zeros((B, Ny, Nx, Nx))
            if self.__gp_method == 'TA':
                c = c + np.einsum('bad,bde,bce->bac', J, S, J)
                dcz += np.einsum('badp,bde,bce->bacp', Hm, S, J) + np.einsum('bad,bde,bcep->bacp', J, S, Hm)
                dcS = np.einsum('bad,bce->bacde', J, J)
            dmean = J.copy()
        if self.__normalize:
            mean = self.inverse_mean(mean, self.__meanY, self.__stdY)
            sz = np.concatenate([np.atleast_1d(self.__stdX), np.atleast_1d(self.__stdU)])
            dmean = dmean * np.atleast_1d(self.__stdY)[None, :, None] / sz[None, None, :]
            dmS = dmS * np.atleast_1d(self.__stdY)[None, :, None, None]
            dcz = dcz / sz[None, None, None, :]
        D = {'dmean_dx': dmean[:, :, :Ny], 'dmean_du': dmean[:, :, Ny:], 'dmean_dcov': dmS,
             'dcov_dx': dcz[:, :, :, :Ny], 'dcov_du': dcz[:, :, :, Ny:], 'dcov_dcov': dcS}
        return mean.reshape(B, Ny), c, D

    def predict_batch(self, Z, Sigma=None, method=None, standardized=True):
        """All B inputs in one device call (the pattern MPC's Nt shooting nodes want):
        Z[B x Nx] (already standardised unless standardized=False), Sigma[B x Nx x Nx]
        -> mean[B x Ny] (un-standardised like `predict`), cov[B x Ny x Ny]."""
        Z = np.asarray(Z, dtype=np.float64).reshape(-1, self.__Nx)
        if self.__normalize and not standardized:
            Z = self.standardize(Z, self.__meanZ, self.__stdZ)
        mean, c = self._h.predict(method or self.__gp_method, Z, Sigma)
        if self.__normalize:
            mean = self.inverse_mean(mean, self.__meanY, self.__stdY)
        return mean, c

    def get_size(self):
        """(N, Ny, Nu) -- gp_class.py:266-274."""
        return self.__N, self.__Ny, self.__Nu

    def get_hyper_parameters(self):
        """gp_class.py:277-290."""
        return dict(length_scale=self.__hyper_length_scales, signal_var=self.__hyper_signal_variance,
                    noise_var=self.__hyper_noise_variance, mean=self.__hyper_mean)

    def print_hyper_parameters(self):
        """Print all hyper-parameters (what gp_class.py:293-312 reports: sizes, per output the d
        length scales, signal variance sf^2 and noise variance sn^2)."""
        rule = '-' * 40
        lines = ['', '_' * 40, '# Hyper-parameters', rule,
                 f'* Num samples: {self.__N}', f'* Ny: {self.__Ny}', f'* Nu: {self.__Nu}',
                 f'* Normalization: {self.__normalize}']
        for a in range(self.__Ny):
            lines += [rule, f'* Lengthscale:  {a}']
            lines += [f'-- l{i}: {ell}' for i, ell in enumerate(self.__hyper_length_scales[a])]
            lines += [f'* Signal variance:  {a}', f'-- sf2: {self.__hyper_signal_variance[a]}',
                      f'* Noise variance:  {a}', f'-- sn2: {self.__hyper_noise_variance[a]}']
        lines.append(rule)
        print('\n'.join(lines))

    def covSEard(self, X, Z, ell, sf2):
        """GP squared exponential kernel k(X, Z) (gp_class.py:314-350), evaluated on the device."""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
        if X.shape[1] != Z.shape[1]:
            raise ValueError('Input dimensions are not the same! D_x=' + str(X.shape[1])
                             + ', D_z=' + str(Z.shape[1]))                      # gp_class.py:342-344
        return self._lib.kernel_matrix(X, Z, np.asarray(ell, dtype=np.float64), float(sf2), device=self._device)

    def covar(self, X_new):
        """Posterior covariance between new inputs (gp_class.py:353-381): (D x n x n) array whose
        first Ny slabs are filled, like the reference."""
        X_new = np.atleast_2d(np.asarray(X_new, dtype=np.float64))
        n, D = X_new.shape
        out = np.zeros((D, n, n))
        out[:self.__Ny] = self._h.covar(X_new)
        return out

    # ------------------------------------------------------------------ data updates
    def update_data(self, X_new, Y_new, N_new=None):
        """The reference documents its incremental update as not working (gp_class.py:384-471:
        argmin for 'max variance' :421, norm instead of norm^2 :443); use `update_data_all`."""
        raise NotImplementedError('update_data is broken in the reference (gp_class.py:384-471); '
                                  'use update_data_all / replace_data_all')

    def _set_data_and_refit(self, X, Y):
        self.__X, self.__Y = X, Y
        self.__N = X.shape[0]
        self._refit()

    def update_data_all(self, X_new, Y_new):
        """Append all new observations and bring chol/alpha/invK up to date with the EXISTING
        hyper-parameters (gp_class.py:474-550).  The reference recomputes everything, O(N^3); here
        the factors are extended on the device (`gpmpc_append`, O(N^2 n)) -- same result to rounding."""
        X_new = np.array(X_new, dtype=np.float64).copy().reshape(-1, self.__Nx)
        Y_new = np.array(Y_new, dtype=np.float64).copy().reshape(-1, self.__Ny)
        if self.__normalize:
            Y_new = self.standardize(Y_new, self.__meanY, self.__stdY)
            X_new = self.standardize(X_new, self.__meanZ, self.__stdZ)
        self._h.append(X_new, Y_new)
        self.__X = np.vstack([self.__X, X_new])
        self.__Y = np.vstack([self.__Y, Y_new])
        self.__N = self.__X.shape[0]

    def replace_data_all(self, X_new, Y_new):
        """Replace the training data, keep the hyper-parameters (gp_class.py:553-626)."""
        X_new = np.array(X_new, dtype=np.float64).copy()
        Y_new = np.array(Y_new, dtype=np.float64).copy()
        if self.__normalize:
            Y_new = self.standardize(Y_new, self.__meanY, self.__stdY)
            X_new = self.standardize(X_new, self.__meanZ, self.__stdZ)
        self._set_data_and_refit(X_new, Y_new)

    # ------------------------------------------------------------------ scaling helpers (gp_class.py:629-644)
    def standardize(self, Y, mean, std):
        return (Y - mean) / std

    def normalize(self, u, lb, ub):
        return (u - lb) / (ub - lb)

    def inverse_mean(self, x, mean, std):
        return (x * std) + mean

    def inverse_variance(self, variance):
        return variance * self.__stdY ** 2

    # ------------------------------------------------------------------ linearisation
    def discrete_linearize(self, x0, u0, cov0):
        """x[k+1] = A x[k] + B u[k] around (x0, u0): Jacobians of the predicted mean w.r.t. the
        standardised x and u (gp_class.py:647-661: `jac_x`, `jac_u` of `__predict`'s mean, :239-242).  For ME / TA /
        old_* that mean is the plain GP mean (analytic Jacobian on the device); for 'EM' it is the exact-moment mean,
        whose derivative with respect to the input mean comes from `gpmpc_predict_em_sens` and depends on cov0."""
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        u0 = np.asarray(u0, dtype=np.float64).reshape(-1)
        if self.__normalize:
            x0 = self.standardize(x0, self.__meanX, self.__stdX)
            u0 = self.standardize(u0, self.__meanU, self.__stdU)
        z = np.concatenate([x0, u0]).reshape(1, self.__Nx)
        if self.__gp_method == 'EM':
            S = np.asarray(cov0, dtype=np.float64).reshape(1, self.__Nx, self.__Nx)
            J = self._h.predict_em_sens(z, S)[2]
        else:
            _, J = self._h.mean_jac(z)
        return J[0][:, :self.__Ny].copy(), J[0][:, self.__Ny:].copy()

    def jacobian(self, x0, u0, cov0):
        """J = d mu / d x at raw (x0, u0), no standardisation (gp_class.py:664-672)."""
        z = np.concatenate([np.asarray(x0, dtype=np.float64).reshape(-1),
                            np.asarray(u0, dtype=np.float64).reshape(-1)]).reshape(1, self.__Nx)
        _, J = self._h.mean_jac(z)
        return J[0][:, :self.__Ny].copy()

    def noise_variance(self):
        return self.__hyper_noise_variance                                      # gp_class.py:675-678

    def sparse(self, M):
        """FITC stub, empty in the reference too (gp_class.py:682-689)."""
        return None

    # ------------------------------------------------------------------ persistence (gp_class.py:693-743)
    def _to_dict(self, as_arrays=False):
        """Model under the key names of the reference's file format (gp_class.py:693-726), factors
        exported from the device; plain lists unless `as_arrays`."""
        f = self._h.get_factors(chol=True, alpha=True, invK=True)
        as_list = (lambda v: np.asarray(v)) if as_arrays else (lambda v: np.asarray(v).tolist())
        out = {'X': as_list(self.__X), 'Y': as_list(self.__Y),
               'hyper': {k: as_list(v) for k, v in (
                   ('hyper', self.__hyper), ('invK', f['invK']), ('alpha', f['alpha']), ('chol', f['chol']),
                   ('length_scale', self.__hyper_length_scales), ('signal_var', self.__hyper_signal_variance),
                   ('noise_var', self.__hyper_noise_variance), ('mean', self.__hyper_mean))},
               'mean_func': self.__mean_func, 'normalize': self.__normalize}
        if self.__normalize:
            for k in ('xlb', 'xub', 'ulb', 'uub'):
                out[k] = as_list(getattr(self, '_GP__' + k))
            out['meta'] = {k: as_list(getattr(self, '_GP__' + k))
                           for k in ('meanY', 'stdY', 'meanZ', 'stdZ', 'meanX', 'stdX', 'meanU', 'stdU')}
        return out

    SIDECAR_MIN_N = 1024     # from this many training points on, save_model moves the matrices out of the JSON

    def save_model(self, filename, sidecar=None):
        """Save model to `filename`.json in the reference's format (gp_class.py:729-734).

        `sidecar` (default: N >= SIDECAR_MIN_N): write the big arrays -- X, Y and hyper's chol, invK,
        alpha -- to `filename`.npz instead and leave `{"__sidecar__": key}` placeholders in the JSON
        (a JSON of Ny x N x N doubles is impractical beyond a few thousand points: 6 x 8192^2
        numbers are ~9 GB of text).  Files without a sidecar are byte-for-byte the reference's layout
        and load with the reference's GP.load_model; files with one need this class."""
        d = self._to_dict(as_arrays=True)
        if sidecar is None:
            sidecar = self.__N >= self.SIDECAR_MIN_N
        if sidecar:
            big = {'X': d['X'], 'Y': d['Y']}
            for k in ('chol', 'invK', 'alpha'):
                big['hyper_' + k] = d['hyper'][k]
                d['hyper'][k] = {'__sidecar__': 'hyper_' + k}
            d['X'] = {'__sidecar__': 'X'}
            d['Y'] = {'__sidecar__': 'Y'}
            d['sidecar_file'] = os.path.basename(filename) + '.npz'
            np.savez(filename + '.npz', **big)
        with open(filename + ".json", "w") as outfile:
            json.dump(_listify(d), outfile)

    @classmethod
    def load_model(cls, filename, **kwargs):
        """Create a new model from `filename`.json (gp_class.py:737-743); files written by the
        reference load unchanged, `__sidecar__` placeholders are resolved from the .npz next to the
        JSON.  kwargs: device=, lib=."""
        with open(filename + ".json") as json_data:
            input_dict = json.load(json_data)
        side = input_dict.pop('sidecar_file', None)
        if side is not None:
            with np.load(os.path.join(os.path.dirname(filename + '.json'), side)) as z:
                def resolve(v):
                    if isinstance(v, dict) and '__sidecar__' in v:
                        return z[v['__sidecar__']]
                    if isinstance(v, dict):
                        return {k: resolve(x) for k, x in v.items()}
                    return v
                input_dict = resolve(input_dict)
        input_dict.update(kwargs)
        return cls(**input_dict)

    # ------------------------------------------------------------------ rollout (numeric part of predict_compare)
    @staticmethod
    def lqr_gain(A, B, Q, R):
        """Infinite-horizon discrete LQR gain u = K x, the three lines of mpc_class.lqr (mpc_class.py:972-973) that
        predict_compare needs: P from the discrete algebraic Riccati equation, K = -(R + B^T P B)^-1 B^T P A."""
        import scipy.linalg
        P = np.array(scipy.linalg.solve_discrete_are(A, B, Q, R))
        return -np.array(scipy.linalg.solve(R + B.T @ P @ B, B.T @ P @ A))

    ROLLOUT_MULTI_MIN_N = 2048      # from this many training points on, rollout runs its methods in lock-step on the device

    def rollout(self, x0, u, methods=None, feedback=False, x_ref=None, Q=None, R=None, K=None, return_controls=False):
        """The numeric loop of `predict_compare` (gp_class.py:746-804) without simulator and
        plots: for every method feed (mean_t, cov_t) back into `predict` for Nt = len(u) steps.
        Returns mean[len(methods), Nt+1, Ny] and var[...] (variances un-standardised by stdY^2 and
        clipped at 0 like :795-796,826-827).

        feedback=True (gp_class.py:772-803): u_t = K (mean_t - x_ref) with the LQR gain of the model linearised at
        (x0, u[0]) (`discrete_linearize` + `lqr_gain`; pass K to use a given gain), and the control blocks of the
        input covariance K C K^T, C K^T.  DIFF: the reference subtracts x_ref (Ny,) from the (Ny x 1) array `predict`
        returns, which numpy broadcasts to an Ny x Ny matrix from the second step on ("#TODO: Fix feedback",
        gp_class.py:806); here the law is evaluated on vectors as written in its docstring."""
        Nx, Ny, Nu = self.__Nx, self.__Ny, self.__Nu
        u = np.atleast_2d(np.asarray(u, dtype=np.float64))
        Nt = u.shape[0]
        initVar = self.__hyper[:, Nx + 1] ** 2
        if methods is None:
            methods = ['EM', 'TA', 'ME']
        mean = np.zeros((len(methods), Nt + 1, Ny))
        var = np.zeros((len(methods), Nt + 1, Ny))
        controls = np.zeros((len(methods), Nt, Nu))
        covar = np.eye(Nx) * 1e-6                                               # gp_class.py:764
        # DIFF: the reference synchronises with the predictor once per step (a Python loop around GP.predict);
        # here the whole horizon runs on the device (`gpmpc_rollout`) and the result comes back once.
        for m in methods:
            if m not in METHODS:
                raise NameError('No GP method called: ' + str(m))
        x0 = np.asarray(x0, dtype=np.float64).reshape(Ny)
        norm = self.__normalize
        one = np.ones(1)
        stdX, meanX = (np.atleast_1d(self.__stdX), np.atleast_1d(self.__meanX)) if norm else (one, 0 * one)
        stdU, meanU = (np.atleast_1d(self.__stdU), np.atleast_1d(self.__meanU)) if norm else (one, 0 * one)
        stdY, meanY = (np.atleast_1d(self.__stdY), np.atleast_1d(self.__meanY)) if norm else (one, 0 * one)
        sa = stdY / stdX if norm else None                                        # x_s(next) = sa * mean_s + sb
        sb = (meanY - meanX) / stdX if norm else None
        Us = (u - meanU) / stdU
        if feedback:
            if x_ref is None:
                x_ref = np.zeros(Ny)                                              # gp_class.py:770-771
            x_ref = np.asarray(x_ref, dtype=np.float64).reshape(Ny)
            Q = np.eye(Ny) if Q is None else np.asarray(Q, dtype=np.float64)      # :765-768
            R = np.eye(Nu) if R is None else np.asarray(R, dtype=np.float64)
        keep = self.__gp_method
        if not feedback and len(methods) > 1 and self.__N >= self.ROLLOUT_MULTI_MIN_N:
            # every method from the same start, in lock-step: ONE pass over the factors per time step serves all of them
            # (gpmpc_rollout_multi; small models keep the per-method calls, whose loops are replayed from captured graphs)
            covar[:Ny, :Ny] = np.diag(initVar)
            z0 = np.concatenate([(x0 - meanX) / stdX, Us[0]])
            mean_s, cov = self._h.rollout_multi(list(methods), z0, Us, covar, sa, sb)
            for i in range(len(methods)):
                controls[i] = u
                mean[i, 0, :] = x0
                mean[i, 1:, :] = self.inverse_mean(mean_s[i], self.__meanY, self.__stdY) if norm else mean_s[i]
                var[i, 1:, :] = np.einsum('tii->ti', cov[i])
                if norm:
                    var[i, 1:, :] = self.inverse_variance(var[i, 1:, :])
            if np.any(var < 0):
                var = var.clip(min=0)
            return (mean, var, controls) if return_controls else (mean, var)
        for i, m in enumerate(methods):
            covar[:Ny, :Ny] = np.diag(initVar)                                    # gp_class.py:780 (other blocks persist)
            if feedback:
                if K is None:
                    self.set_method(m)
                    A, B = self.discrete_linearize(x0, u[0], covar)               # :785-786
                    Km = self.lqr_gain(A, B, Q, R)
                else:
                    Km = np.asarray(K, dtype=np.float64).reshape(Nu, Ny)
                u1 = Km @ (x0 - x_ref)                                            # :789-790
                z0 = np.concatenate([(x0 - meanX) / stdX, (u1 - meanU) / stdU])
                Kz = (Km * stdY[None, :]) / stdU[:, None]
                k0 = (Km @ (meanY * np.ones(Ny) - x_ref) - meanU) / stdU
                mean_s, cov, U_s = self._h.rollout_feedback(m, Nt, z0, covar, Kz, k0, Km, sa, sb)
                controls[i] = U_s * stdU + meanU
                covar[:Ny, Ny:] = cov[-1] @ Km.T                                  # what :798-803 leave behind for the
                covar[Ny:, :Ny] = covar[:Ny, Ny:].T                               # next method (covar is never reset)
                covar[Ny:, Ny:] = Km @ cov[-1] @ Km.T
            else:
                z0 = np.concatenate([(x0 - meanX) / stdX, Us[0]])
                mean_s, cov = self._h.rollout(m, z0, Us, covar, sa, sb)
                controls[i] = u
            mean[i, 0, :] = x0
            mean[i, 1:, :] = self.inverse_mean(mean_s, self.__meanY, self.__stdY) if norm else mean_s
            var[i, 1:, :] = np.einsum('tii->ti', cov)
            if norm:
                var[i, 1:, :] = self.inverse_variance(var[i, 1:, :])
        self.__gp_method = keep
        if np.any(var < 0):
            var = var.clip(min=0)
        return (mean, var, controls) if return_controls else (mean, var)

    def predict_compare(self, x0, u, model=None, num_cols=2, xnames=None, title=None, feedback=False, x_ref=None,
                        Q=None, R=None, methods=None):
        """`predict_compare` (gp_class.py:746-861) without its matplotlib front end: the T-step uncertainty propagation
        of every method in `methods` (default ['EM', 'TA', 'ME'], :759-760) through `GP.rollout` -- the whole horizon on
        the device -- next to the simulated trajectory of `model` when one is given (`model.sim(x0, u)`, :819-820;
        model_class needs casadi + SUNDIALS and is out of scope, so any object with that method -- and
        `sampling_time()` for the time axis, :756 -- does).  The scripts that call it (van_der_pol.py:83-85,
        tank_example.py) keep running; instead of figures they get the arrays the figures were drawn from:

            {'t': [Nt+1], 'methods': [...], 'mean': [M, Nt+1, Ny], 'var': [M, Nt+1, Ny] (clipped at 0, :826-827),
             'y_sim': [Nt+1, Ny] or None}

        num_cols, xnames, title only shaped the plot and are accepted for signature compatibility."""
        u = np.atleast_2d(np.asarray(u, dtype=np.float64))
        Nt = u.shape[0]
        if methods is None:
            methods = ['EM', 'TA', 'ME']
        mean, var = self.rollout(x0, u, methods=methods, feedback=feedback, x_ref=x_ref, Q=Q, R=R)
        y_sim, dt = None, 1.0
        if model is not None:
            if hasattr(model, 'sampling_time'):
                dt = float(model.sampling_time())
            if hasattr(model, 'sim'):
                y_sim = np.vstack([np.asarray(x0, dtype=np.float64).reshape(1, -1),
                                   np.asarray(model.sim(x0, u), dtype=np.float64).reshape(Nt, -1)])   # :819-820
        return {'t': np.linspace(0.0, Nt * dt, Nt + 1), 'methods': list(methods), 'mean': mean, 'var': var.clip(min=0),
                'y_sim': y_sim}

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None
