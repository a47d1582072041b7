"""Bound-only evaluation: right-hand side carried through the factorisation (chol_rhs=1) vs explicit L_B^-1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
for (N, M, D) in ((3000, 100, 5), (5000, 129, 3), (20000, 512, 8), (20000, 1000, 16), (30000, 3071, 8), (100000, 2048, 32)):
    X, y, Z = bench.synthetic(N, D, M, seed=N + M)
    h = HipGP(0)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.3], np.full((1, D), 0.8 * np.sqrt(D)), 0.02)
    res = {}
    for mode in (0, 1):
        h.set_option("chol_rhs", mode)
        v = h.sparse_lml(); h.sparse_lml()
        t0 = time.perf_counter()
        for _ in range(10):
            h.sparse_lml()
        res[mode] = (v, (time.perf_counter() - t0) * 100)
    g = h.sparse_lml_grad()[0]
    print("N %6d M %4d D %2d: explicit %.10f (%.3f ms)  rhs-row %.10f (%.3f ms)  rel diff %.2e  | bound of the gradient call %.10f"
          % (N, M, D, res[0][0], res[0][1], res[1][0], res[1][1], abs(res[0][0] - res[1][0]) / abs(res[0][0]), g), flush=True)
    h.close()
