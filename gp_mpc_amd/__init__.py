"""gp_mpc_amd: MI355X-native (gfx950) Gaussian-process inner loop of GP-MPC.

Hot path only: SE-ARD kernel build, Cholesky + triangular inverse on fp64
MFMA, predictive mean/variance, TA / exact-moment covariance propagation, NLL (+ gradient)
training with a restart shard.  Hand-written HIP kernels behind the C ABI of include/gpmpc.h;
this package is the Python host side that mirrors the reference's `GP` surface.
"""
from ._lib import GpmpcError, GpmpcLib, Handle, NotPositiveDefinite, get_lib  # noqa: F401
from .gp import GP  # noqa: F401
from .train import train_gp  # noqa: F401

__all__ = ['GP', 'Handle', 'GpmpcLib', 'GpmpcError', 'NotPositiveDefinite', 'get_lib', 'train_gp']
