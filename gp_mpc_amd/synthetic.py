"""Seeded synthetic workloads for the GP hot path: the generator that stands in for the
reference's plant-simulator data (`Model.generate_training_data`, /root/reference/gp_mpc/model_class.py:299-369) in
`bench.py`, the tools and the tests.  Data generation only: no GP arithmetic, no dependency on `oracle/` (which imports
this module so that the checker and the product see the same inputs)."""
import numpy as np


def synthetic_problem(N, d, Ny, B, seed=1234, sn=1e-2):
    """X ~ N(0,1) (standardised inputs, as the reference trains on: gp_class.py:105);
    y_a = sin(X w_a) + 0.5 cos(X u_a) + 1e-2 eps, standardised; hyper[a] = [2 (1 + 0.1 a) 1_d, sf = 1, sn];
    test points Z ~ N(0,1) [B x d]; input covariances Sigma = A A^T 1e-3 + 1e-6 I [B x d x d]."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, d))
    Y = np.zeros((N, Ny))
    for a in range(Ny):
        w = rng.standard_normal(d)
        u = rng.standard_normal(d)
        y = np.sin(X @ w) + 0.5 * np.cos(X @ u) + 1e-2 * rng.standard_normal(N)
        Y[:, a] = (y - y.mean()) / y.std()
    hyper = np.zeros((Ny, d + 2))
    for a in range(Ny):
        hyper[a, :d] = 2.0 * (1 + 0.1 * a)
        hyper[a, d] = 1.0
        hyper[a, d + 1] = sn
    Z = rng.standard_normal((B, d))
    A = rng.standard_normal((B, d, d))
    Sigma = np.einsum('bij,bkj->bik', A, A) * 1e-3 + 1e-6 * np.eye(d)
    return dict(X=X, Y=Y, hyper=hyper, Z=Z, Sigma=Sigma)
