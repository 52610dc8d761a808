"""ctypes binding of the C ABI in include/gpmpc.h (libgpmpc_hip.so).

The product path has exactly one backend: the hipcc-built gfx950 library that lives in-tree at
gp_mpc_amd/csrc/libgpmpc_hip.so.  `get_lib()` raises if it is missing or cannot be loaded -- there
is no CPU fallback.  (`GpmpcLib(path)` takes an explicit path only so that the CPU test tier can
hand in the HIP-emulator build of the same sources, tests/emu/.)
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.path.join(CSRC, 'libgpmpc_hip.so')
HEADER = os.path.abspath(os.path.join(HERE, '..', 'include', 'gpmpc.h'))

OK, EINVAL, EHIP, ENOTFIT, ENOTPD, ENOMEM = 0, -1, -2, -3, -4, -5
METHODS = {'ME': 0, 'TA': 1, 'EM': 2, 'old_ME': 3, 'old_TA': 4}
MEAN_FUNCS = {'zero': 0, 'const': 1, 'linear': 2, 'polynomial': 3}     # gp_functions.py:25-69
PTR_HOST, PTR_DEVICE = 0, 1
PHASES = ['gram', 'factor', 'solve', 'invK', 'crosscov', 'vargemm', 'finish', 'em', 'nll', 'chain']

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol gpmpc.h declares (tests/test_abi.py checks)
SIGNATURES = {
    'gpmpc_abi_version': (ctypes.c_int, []),
    'gpmpc_runtime_info': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'gpmpc_profile_set_mask': (ctypes.c_int, [_vp, ctypes.c_uint]),
    'gpmpc_last_error': (ctypes.c_char_p, []),
    'gpmpc_device_count': (ctypes.c_int, [_ip]),
    'gpmpc_device_name': (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    'gpmpc_mfma_selftest': (ctypes.c_int, [ctypes.c_int, _ip, _dp]),
    'gpmpc_schedule_stats': (ctypes.c_int, [ctypes.c_int] * 6 + [_dp]),
    'gpmpc_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp,
                                    ctypes.POINTER(_vp)]),
    'gpmpc_destroy': (ctypes.c_int, [_vp]),
    'gpmpc_get_size': (ctypes.c_int, [_vp, _ip, _ip, _ip]),
    'gpmpc_set_mean_func': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    'gpmpc_hyper_width': (ctypes.c_int, [_vp, _ip]),
    'gpmpc_set_hyper_prior': (ctypes.c_int, [_vp, _vp]),
    'gpmpc_set_pointer_mode': (ctypes.c_int, [_vp, ctypes.c_int]),
    'gpmpc_set_stream': (ctypes.c_int, [_vp, _vp]),
    'gpmpc_synchronize': (ctypes.c_int, [_vp]),
    'gpmpc_get_counter': (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]),
    'gpmpc_profile_enable': (ctypes.c_int, [_vp, ctypes.c_int]),
    'gpmpc_profile_read': (ctypes.c_int, [_vp, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_long), ctypes.c_int]),
    'gpmpc_fit': (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    'gpmpc_fit_predict_mean_var': (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp]),
    'gpmpc_get_factors': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    'gpmpc_set_factors': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    'gpmpc_append': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    'gpmpc_predict_mean_var': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    'gpmpc_mean_jac': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    'gpmpc_predict_sens': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gpmpc_predict_em_sens': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gpmpc_predict_jac': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    'gpmpc_rollout': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gpmpc_rollout_feedback': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gpmpc_rollout_multi': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'gpmpc_predict': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp]),
    'gpmpc_covar': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp]),
    'gpmpc_nll': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _dp, _vp, _ip]),
    'gpmpc_train_multistart': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                              ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _ip]),
    'gpmpc_rccl_unique_id': (ctypes.c_int, [ctypes.c_char_p]),
    'gpmpc_rccl_comm_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(_vp)]),
    'gpmpc_rccl_comm_destroy': (ctypes.c_int, [_vp]),
    'gpmpc_rccl_comm_count': (ctypes.c_int, [_vp, _ip]),
    'gpmpc_kernel_matrix': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp,
                                           ctypes.c_double, _vp]),
    'gpmpc_cholesky': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, _vp, _ip]),
    'gpmpc_set_tuning': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'gpmpc_dgemm': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_double, _vp, ctypes.c_int, _vp, ctypes.c_int,
                                   ctypes.c_double, _vp, ctypes.c_int]),
}


class GpmpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'gpmpc error {code}: {msg}')
        self.code = code


class NotPositiveDefinite(GpmpcError, np.linalg.LinAlgError):
    """K not SPD even after the one-shot jitter (the reference lets LinAlgError propagate,
    optimize.py:349-350)."""


def _ptr(a):
    """host ndarray (C-contiguous fp64) or raw device address (int) -> void*"""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return ctypes.c_void_p(int(a))
    assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS'], 'need C-contiguous float64'
    return a.ctypes.data_as(ctypes.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class GpmpcLib:
    """Loaded C-ABI library with typed entry points."""

    def __init__(self, path):
        self.path = path
        self.dll = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args

    def check(self, rc):
        if rc == OK:
            return
        msg = (self.dll.gpmpc_last_error() or b'').decode('utf-8', 'replace')
        if rc == ENOTPD:
            raise NotPositiveDefinite(rc, msg)
        raise GpmpcError(rc, msg)

    # -- library / device
    def device_count(self):
        n = ctypes.c_int(0)
        self.check(self.dll.gpmpc_device_count(ctypes.byref(n)))
        return n.value

    def device_name(self, device=0):
        buf = ctypes.create_string_buffer(256)
        self.check(self.dll.gpmpc_device_name(device, buf, 256))
        return buf.value.decode()

    def mfma_selftest(self, device=0):
        layout = ctypes.c_int(-1)
        tf = ctypes.c_double(0.0)
        self.check(self.dll.gpmpc_mfma_selftest(device, ctypes.byref(layout), ctypes.byref(tf)))
        return layout.value, tf.value

    def schedule_stats(self, mode, tilesM, tilesN, batch, K, slots):
        """Static tile schedule of the persistent products (host code): dict(tiles, max_load, mean_load, home, wrong, longest)."""
        st = (ctypes.c_double * 6)()
        self.check(self.dll.gpmpc_schedule_stats(int(mode), int(tilesM), int(tilesN), int(batch), int(K), int(slots), st))
        return dict(tiles=int(st[0]), max_load=st[1], mean_load=st[2], home=int(st[3]), wrong=int(st[4]), longest=int(st[5]))

    def runtime_info(self):
        """{'hip_runtime', 'hip_path', 'rccl', 'rccl_path'}: the HIP runtime and RCCL build this process runs the library on."""
        buf = ctypes.create_string_buffer(1024)
        self.check(self.dll.gpmpc_runtime_info(buf, 1024))
        return dict(kv.split('=', 1) for kv in buf.value.decode().split(' ') if '=' in kv)

    def set_tuning(self, name, value):
        """Diagnostic knob, e.g. set_tuning('gemm_tile', 64) pins the GEMM tile (0 = automatic)."""
        self.check(self.dll.gpmpc_set_tuning(name.encode(), int(value)))

    # -- RCCL bootstrap of the restart shard
    def rccl_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        self.check(self.dll.gpmpc_rccl_unique_id(buf))
        return buf.raw

    def rccl_comm_create(self, device, world, rank, id128):
        comm = ctypes.c_void_p()
        self.check(self.dll.gpmpc_rccl_comm_create(device, world, rank, ctypes.create_string_buffer(id128, 128), ctypes.byref(comm)))
        return comm

    def rccl_comm_destroy(self, comm):
        self.check(self.dll.gpmpc_rccl_comm_destroy(comm))

    def rccl_comm_count(self, comm):
        """ncclCommCount of a communicator made by rccl_comm_create: the number of ranks RCCL itself sees."""
        n = ctypes.c_int(0)
        self.check(self.dll.gpmpc_rccl_comm_count(comm, ctypes.byref(n)))
        return n.value

    # -- low-level dense ops
    def cholesky(self, A, device=0, want_inverse=False):
        A = _f64(A).copy()
        n = A.shape[0]
        inv = np.zeros_like(A) if want_inverse else None
        info = ctypes.c_int(0)
        self.check(self.dll.gpmpc_cholesky(device, n, _ptr(A), _ptr(inv), ctypes.byref(info)))
        return (A, inv, info.value) if want_inverse else (A, info.value)

    def kernel_matrix(self, X, Z, ell, sf2, device=0):
        X, Z, ell = _f64(X), _f64(Z), _f64(ell)
        out = np.zeros((X.shape[0], Z.shape[0]))
        self.check(self.dll.gpmpc_kernel_matrix(device, X.shape[0], Z.shape[0], X.shape[1], _ptr(X), _ptr(Z), _ptr(ell),
                                                float(sf2), _ptr(out)))
        return out

    def dgemm(self, A, B, C=None, alpha=1.0, beta=0.0, transa=False, transb=False, device=0):
        A, B = _f64(A), _f64(B)
        M = A.shape[1] if transa else A.shape[0]
        K = A.shape[0] if transa else A.shape[1]
        N = B.shape[0] if transb else B.shape[1]
        C = np.zeros((M, N)) if C is None else _f64(C).copy()
        self.check(self.dll.gpmpc_dgemm(device, int(transa), int(transb), M, N, K, alpha, _ptr(A), A.shape[1],
                                        _ptr(B), B.shape[1], beta, _ptr(C), N))
        return C


class Handle:
    """One GP model on one GPU (gpmpc_gp*).  Thin, typed wrapper; numpy in / numpy out in host
    pointer mode, raw device addresses (e.g. torch.Tensor.data_ptr()) in device mode."""

    def __init__(self, lib: GpmpcLib, X, Y, device=0):
        self.lib = lib
        X, Y = _f64(X), _f64(Y)
        self.N, self.d = X.shape
        self.Ny = Y.shape[1]
        assert Y.shape[0] == self.N
        h = ctypes.c_void_p()
        lib.check(lib.dll.gpmpc_create(device, self.N, self.d, self.Ny, _ptr(X), _ptr(Y), ctypes.byref(h)))
        self.h = h
        self.device = int(device)            # the GPU this model lives on (an RCCL communicator for it must be made there)
        self.device_mode = False
        self.nh = self.d + 2                 # entries of a hyper row: [ell.., sf, sn] + mean-function parameters

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dll.gpmpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mean_func(self, mean_func='zero', add_to_prediction=False):
        """Prior mean function 'zero' | 'const' | 'linear' | 'polynomial' (gp_functions.py:25-69); hyper rows grow by
        its parameters.  add_to_prediction: build_gp(meanFunc=...) semantics, see include/gpmpc.h."""
        if mean_func not in MEAN_FUNCS:
            raise NameError('No mean function called: ' + str(mean_func))          # gp_functions.py:67
        self.lib.check(self.lib.dll.gpmpc_set_mean_func(self.h, MEAN_FUNCS[mean_func], int(add_to_prediction)))
        w = ctypes.c_int(0)
        self.lib.check(self.lib.dll.gpmpc_hyper_width(self.h, ctypes.byref(w)))
        self.nh = w.value

    def set_hyper_prior(self, prior=None):
        """prior: dict with the keys of optimize.py:158-165 (ell_mean, ell_std, sf_mean, sf_std, sn_mean, sn_std) or None."""
        p6 = None if prior is None else _f64([prior[k] for k in ('ell_mean', 'ell_std', 'sf_mean', 'sf_std', 'sn_mean', 'sn_std')])
        self.lib.check(self.lib.dll.gpmpc_set_hyper_prior(self.h, _ptr(p6)))

    def set_pointer_mode(self, device: bool):
        self.lib.check(self.lib.dll.gpmpc_set_pointer_mode(self.h, PTR_DEVICE if device else PTR_HOST))
        self.device_mode = device

    def set_stream(self, stream_ptr):
        self.lib.check(self.lib.dll.gpmpc_set_stream(self.h, ctypes.c_void_p(stream_ptr or 0)))

    def synchronize(self):
        self.lib.check(self.lib.dll.gpmpc_synchronize(self.h))

    def counter(self, name):
        """'handoff_timeouts' | 'chained_factorisations' | 'single_queue_factorisations' | 'workspace_blocks_fresh' |
        'workspace_blocks_reused' | 'train_iterations' | 'train_evaluations' | 'predictions_behind_tail' |
        'persistent_variance_products' (include/gpmpc.h)."""
        v = ctypes.c_long(0)
        self.lib.check(self.lib.dll.gpmpc_get_counter(self.h, name.encode(), ctypes.byref(v)))
        return v.value

    def profile_enable(self, on=True, phases=None):
        """HIP-event brackets per phase; `phases` (names from PHASES) restricts them -- every bracket costs stream time."""
        mask = 0
        if on and phases is not None:
            for name in phases:
                mask |= 1 << PHASES.index(name)
        self.lib.check(self.lib.dll.gpmpc_profile_set_mask(self.h, mask))
        self.lib.check(self.lib.dll.gpmpc_profile_enable(self.h, int(bool(on))))

    def profile_read(self, reset=True):
        out = {}
        for i, name in enumerate(PHASES):
            ms, n = ctypes.c_double(0), ctypes.c_long(0)
            self.lib.check(self.lib.dll.gpmpc_profile_read(self.h, i, ctypes.byref(ms), ctypes.byref(n), int(reset)))
            out[name] = (ms.value, n.value)
        return out

    def fit(self, hyper, want_invK=False):
        hyper = _f64(hyper).reshape(self.Ny, self.nh)
        info = np.zeros(self.Ny, dtype=np.int32)
        rc = self.lib.dll.gpmpc_fit(self.h, _ptr(hyper), int(want_invK), info.ctypes.data_as(ctypes.c_void_p))
        self.info = info
        self.lib.check(rc)
        return info

    def get_factors(self, chol=True, alpha=True, invK=False):
        N, Ny = self.N, self.Ny
        hyper = np.zeros((Ny, self.nh))
        L = np.zeros((Ny, N, N)) if chol else None
        al = np.zeros((Ny, N)) if alpha else None
        iK = np.zeros((Ny, N, N)) if invK else None
        self.lib.check(self.lib.dll.gpmpc_get_factors(self.h, _ptr(hyper), _ptr(L), _ptr(al), _ptr(iK)))
        return dict(hyper=hyper, chol=L, alpha=al, invK=iK)

    def set_factors(self, hyper, chol, alpha=None, invK=None):
        hyper = _f64(hyper).reshape(self.Ny, self.nh)
        chol = _f64(chol).reshape(self.Ny, self.N, self.N)
        alpha = None if alpha is None else _f64(alpha).reshape(self.Ny, self.N)
        invK = None if invK is None else _f64(invK).reshape(self.Ny, self.N, self.N)
        self.lib.check(self.lib.dll.gpmpc_set_factors(self.h, _ptr(hyper), _ptr(chol), _ptr(alpha), _ptr(invK)))

    # -- predict family (host mode: numpy in/out)
    def predict_mean_var(self, Z):
        Z = _f64(Z).reshape(-1, self.d)
        B = Z.shape[0]
        mean, var = np.zeros((B, self.Ny)), np.zeros((B, self.Ny))
        self.lib.check(self.lib.dll.gpmpc_predict_mean_var(self.h, B, _ptr(Z), _ptr(mean), _ptr(var)))
        return mean, var

    def mean_jac(self, Z):
        Z = _f64(Z).reshape(-1, self.d)
        B = Z.shape[0]
        mean, J = np.zeros((B, self.Ny)), np.zeros((B, self.Ny, self.d))
        self.lib.check(self.lib.dll.gpmpc_mean_jac(self.h, B, _ptr(Z), _ptr(mean), _ptr(J)))
        return mean, J

    def append(self, Xnew, Ynew):
        """Append training points, keep the hyper-parameters (rank-n update of the factors)."""
        Xnew = _f64(Xnew).reshape(-1, self.d)
        Ynew = _f64(Ynew).reshape(Xnew.shape[0], self.Ny)
        info = np.zeros(self.Ny, dtype=np.int32)
        rc = self.lib.dll.gpmpc_append(self.h, Xnew.shape[0], _ptr(Xnew), _ptr(Ynew), info.ctypes.data_as(ctypes.c_void_p))
        n = ctypes.c_int(0)                 # the library is the authority on the size, whether the call succeeded or not
        self.lib.dll.gpmpc_get_size(self.h, ctypes.byref(n), None, None)
        self.N = n.value
        self.info = info
        self.lib.check(rc)
        return info

    def predict_jac(self, method, Z, Sigma=None):
        """mean[B,Ny], cov[B,Ny,Ny] ('ME'/'TA') and J[B,Ny,d] = d mean / d z from one pass."""
        code = METHODS[method] if isinstance(method, str) else int(method)
        Z = _f64(Z).reshape(-1, self.d)
        B = Z.shape[0]
        if Sigma is not None:
            Sigma = _f64(Sigma).reshape(B, self.d, self.d)
        mean, cov, J = np.zeros((B, self.Ny)), np.zeros((B, self.Ny, self.Ny)), np.zeros((B, self.Ny, self.d))
        self.lib.check(self.lib.dll.gpmpc_predict_jac(self.h, code, B, _ptr(Z), _ptr(Sigma), _ptr(mean), _ptr(cov), _ptr(J)))
        return mean, cov, J

    def rollout(self, method, z0, U, Sigma0, sa=None, sb=None):
        """T-step uncertainty propagation on the device (standardised units): mean[T,Ny], cov[T,Ny,Ny]."""
        code = METHODS[method] if isinstance(method, str) else int(method)
        z0 = _f64(z0).reshape(self.d)
        Nu = self.d - self.Ny
        U = _f64(U).reshape(-1, max(Nu, 1)) if Nu > 0 else np.zeros((int(np.asarray(U).shape[0]), 1))
        T = U.shape[0]
        Sigma0 = _f64(Sigma0).reshape(self.d, self.d)
        sa = None if sa is None else _f64(sa).reshape(self.Ny)
        sb = None if sb is None else _f64(sb).reshape(self.Ny)
        mean, cov = np.zeros((T, self.Ny)), np.zeros((T, self.Ny, self.Ny))
        self.lib.check(self.lib.dll.gpmpc_rollout(self.h, code, T, _ptr(z0), _ptr(U), _ptr(Sigma0), _ptr(sa), _ptr(sb),
                                                  _ptr(mean), _ptr(cov)))
        return mean, cov

    def rollout_multi(self, methods, z0, U, Sigma0, sa=None, sb=None):
        """M roll-outs in lock-step (gpmpc_rollout_multi): methods[M], z0[M,d] (or [d] for all), U[M,T,Nu] (or [T,Nu] for all),
        Sigma0[M,d,d] (or [d,d]); returns mean[M,T,Ny], cov[M,T,Ny,Ny]."""
        codes = np.array([METHODS[m] if isinstance(m, str) else int(m) for m in methods], dtype=np.int32)
        M, Nu = len(codes), self.d - self.Ny
        z0 = np.ascontiguousarray(np.broadcast_to(_f64(z0).reshape(-1, self.d), (M, self.d)))
        U = np.asarray(U, dtype=np.float64)
        T = U.shape[-2] if Nu > 0 else int(U.shape[-1] if U.ndim == 1 else U.shape[-2])
        U = np.ascontiguousarray(np.broadcast_to(U.reshape(-1, T, max(Nu, 1)), (M, T, max(Nu, 1)))) if Nu > 0 else np.zeros((M, T, 1))
        Sigma0 = np.ascontiguousarray(np.broadcast_to(_f64(Sigma0).reshape(-1, self.d, self.d), (M, self.d, self.d)))
        sa = None if sa is None else _f64(sa).reshape(self.Ny)
        sb = None if sb is None else _f64(sb).reshape(self.Ny)
        mean, cov = np.zeros((M, T, self.Ny)), np.zeros((M, T, self.Ny, self.Ny))
        self.lib.check(self.lib.dll.gpmpc_rollout_multi(self.h, M, codes.ctypes.data_as(ctypes.c_void_p), T, _ptr(z0), _ptr(U),
                                                        _ptr(Sigma0), _ptr(sa), _ptr(sb), _ptr(mean), _ptr(cov)))
        return mean, cov

    def rollout_feedback(self, method, T, z0, Sigma0, Kz, k0, Kc, sa=None, sb=None):
        """Roll-out with state feedback u_t = Kz mean_{t-1} + k0 (include/gpmpc.h): mean[T,Ny], cov[T,Ny,Ny], U[T,Nu]."""
        code = METHODS[method] if isinstance(method, str) else int(method)
        Nu = self.d - self.Ny
        z0 = _f64(z0).reshape(self.d)
        Sigma0 = _f64(Sigma0).reshape(self.d, self.d)
        Kz, k0, Kc = _f64(Kz).reshape(Nu, self.Ny), _f64(k0).reshape(Nu), _f64(Kc).reshape(Nu, self.Ny)
        sa = None if sa is None else _f64(sa).reshape(self.Ny)
        sb = None if sb is None else _f64(sb).reshape(self.Ny)
        mean, cov, U = np.zeros((T, self.Ny)), np.zeros((T, self.Ny, self.Ny)), np.zeros((T, Nu))
        self.lib.check(self.lib.dll.gpmpc_rollout_feedback(self.h, code, int(T), _ptr(z0), _ptr(Sigma0), _ptr(sa), _ptr(sb),
                                                           _ptr(Kz), _ptr(k0), _ptr(Kc), _ptr(mean), _ptr(cov), _ptr(U)))
        return mean, cov, U

    def predict_sens(self, Z):
        """mean[B,Ny], var[B,Ny], J[B,Ny,d] = d mean/dz, Hm[B,Ny,d,d] = d2 mean/dz2, dvar[B,Ny,d] = d var/dz."""
        Z = _f64(Z).reshape(-1, self.d)
        B = Z.shape[0]
        mean, var = np.zeros((B, self.Ny)), np.zeros((B, self.Ny))
        J, Hm = np.zeros((B, self.Ny, self.d)), np.zeros((B, self.Ny, self.d, self.d))
        dvar = np.zeros((B, self.Ny, self.d))
        self.lib.check(self.lib.dll.gpmpc_predict_sens(self.h, B, _ptr(Z), _ptr(mean), _ptr(var), _ptr(J), _ptr(Hm),
                                                       _ptr(dvar)))
        return mean, var, J, Hm, dvar

    def predict_em_sens(self, Z, Sigma, want_cov=True):
        """'EM' value and Jacobians: mean[B,Ny], cov[B,Ny,Ny], dmean_dz[B,Ny,d], dmean_dS[B,Ny,d,d],
        dcov_dz[B,Ny,Ny,d], dcov_dS[B,Ny,Ny,d,d].  want_cov=False: cov is None and its pair sums (a quarter of the
        call) are not formed -- what a Jacobian callback wants."""
        Z = _f64(Z).reshape(-1, self.d)
        B, d, Ny = Z.shape[0], self.d, self.Ny
        Sigma = _f64(Sigma).reshape(B, d, d)
        out = [np.zeros(sh) for sh in ((B, Ny), (B, Ny, Ny), (B, Ny, d), (B, Ny, d, d), (B, Ny, Ny, d), (B, Ny, Ny, d, d))]
        if not want_cov:
            out[1] = None
        self.lib.check(self.lib.dll.gpmpc_predict_em_sens(self.h, B, _ptr(Z), _ptr(Sigma), *[_ptr(o) for o in out]))
        return tuple(out)

    def predict(self, method, Z, Sigma=None):
        code = METHODS[method] if isinstance(method, str) else int(method)
        Z = _f64(Z).reshape(-1, self.d)
        B = Z.shape[0]
        if Sigma is not None:
            Sigma = _f64(Sigma).reshape(B, self.d, self.d)
        mean, cov = np.zeros((B, self.Ny)), np.zeros((B, self.Ny, self.Ny))
        self.lib.check(self.lib.dll.gpmpc_predict(self.h, code, B, _ptr(Z), _ptr(Sigma), _ptr(mean), _ptr(cov)))
        return mean, cov

    def covar(self, Xnew):
        Xnew = _f64(Xnew).reshape(-1, self.d)
        n = Xnew.shape[0]
        out = np.zeros((self.Ny, n, n))
        self.lib.check(self.lib.dll.gpmpc_covar(self.h, n, _ptr(Xnew), _ptr(out)))
        return out

    def nll(self, a, hyper_row, want_grad=False):
        hyper_row = _f64(hyper_row).reshape(self.nh)
        val = ctypes.c_double(0.0)
        grad = np.zeros(self.nh) if want_grad else None
        jit = ctypes.c_int(0)
        self.lib.check(self.lib.dll.gpmpc_nll(self.h, int(a), _ptr(hyper_row), ctypes.byref(val), _ptr(grad),
                                              ctypes.byref(jit)))
        self.last_jitter = jit.value
        return (val.value, grad) if want_grad else val.value

    def train_multistart(self, starts, lb, ub, max_iter=0, tol=0.0, rank=0, world=1, comm=None, want_invK=True):
        """gpmpc_train_multistart: starts[Ny, nstart, nh], lb / ub[Ny, nh] -> dict(hyper, obj, theta, info, status,
        iterations, evaluations).  `status` is this rank's device status (0 = fine) when the caller does the exchange
        itself (world > 1, comm None): a rank with status != 0 must still join the exchange and report it."""
        starts = _f64(starts).reshape(self.Ny, -1, self.nh)
        nstart = starts.shape[1]
        lb, ub = _f64(lb).reshape(self.Ny, self.nh), _f64(ub).reshape(self.Ny, self.nh)
        hyper, obj = np.zeros((self.Ny, self.nh)), np.zeros((self.Ny, nstart))
        theta = np.zeros((self.Ny, nstart, self.nh))
        info = np.zeros(self.Ny, dtype=np.int32)
        status = ctypes.c_int(0)
        rc = self.lib.dll.gpmpc_train_multistart(self.h, nstart, _ptr(starts), _ptr(lb), _ptr(ub), int(max_iter), float(tol),
                                                 int(rank), int(world), comm, int(want_invK), _ptr(hyper), _ptr(obj),
                                                 _ptr(theta), info.ctypes.data_as(ctypes.c_void_p), ctypes.byref(status))
        self.info = info
        self.lib.check(rc)
        return dict(hyper=hyper, obj=obj, theta=theta, info=info, status=status.value,
                    status_text=(self.lib.dll.gpmpc_last_error() or b'').decode('utf-8', 'replace') if status.value else '',
                    iterations=self.counter('train_iterations'), evaluations=self.counter('train_evaluations'))

    # -- raw device-pointer entry points (device pointer mode)
    def predict_mean_var_dev(self, B, z_ptr, mean_ptr, var_ptr):
        self.lib.check(self.lib.dll.gpmpc_predict_mean_var(self.h, B, _ptr(z_ptr), _ptr(mean_ptr), _ptr(var_ptr)))

    def fit_predict_mean_var_dev(self, hyper, B, z_ptr, mean_ptr, var_ptr, want_invK=False):
        """gpmpc_fit + gpmpc_predict_mean_var as ONE call with device pointers (the fused route of include/gpmpc.h)."""
        hyper = _f64(hyper).reshape(self.Ny, self.nh)
        info = np.zeros(self.Ny, dtype=np.int32)
        rc = self.lib.dll.gpmpc_fit_predict_mean_var(self.h, _ptr(hyper), int(bool(want_invK)), info.ctypes.data_as(ctypes.c_void_p),
                                                     int(B), ctypes.c_void_p(z_ptr), ctypes.c_void_p(mean_ptr), ctypes.c_void_p(var_ptr))
        self.info = info
        self.lib.check(rc)
        return info

    def fit_predict_mean_var(self, hyper, Z, want_invK=False):
        """The same with host arrays (pointer mode host: the library runs the two calls one after the other)."""
        hyper = _f64(hyper).reshape(self.Ny, self.nh)
        Z = _f64(Z).reshape(-1, self.d)
        B = Z.shape[0]
        info = np.zeros(self.Ny, dtype=np.int32)
        mean, var = np.zeros((B, self.Ny)), np.zeros((B, self.Ny))
        rc = self.lib.dll.gpmpc_fit_predict_mean_var(self.h, _ptr(hyper), int(bool(want_invK)), info.ctypes.data_as(ctypes.c_void_p),
                                                     B, _ptr(Z), _ptr(mean), _ptr(var))
        self.info = info
        self.lib.check(rc)
        return info, mean, var

    def predict_dev(self, method, B, z_ptr, sigma_ptr, mean_ptr, cov_ptr):
        code = METHODS[method] if isinstance(method, str) else int(method)
        self.lib.check(self.lib.dll.gpmpc_predict(self.h, code, B, _ptr(z_ptr), _ptr(sigma_ptr), _ptr(mean_ptr),
                                                  _ptr(cov_ptr)))


# ------------------------------------------------------------------------------------------------
_LIB = None


def build(verbose=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC] + ([] if verbose else ['-s'])
    subprocess.check_call(cmd)
    return LIB_PATH


def get_lib() -> GpmpcLib:
    """The product library.  Fails loudly: no library, no GPU path, no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C {CSRC}` (needs hipcc); gp_mpc_amd has no CPU fallback')
        _LIB = GpmpcLib(LIB_PATH)
    return _LIB
