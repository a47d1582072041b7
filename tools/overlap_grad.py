"""Bound-only evaluation at Cm under the overlap policies."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=1000)
h = HipGP(0)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for ov in (3, 2, 0, 3, 2):
    h.set_option("overlap", ov)
    for _ in range(3):
        h.sparse_lml_grad()
    t0 = time.perf_counter()
    for _ in range(20):
        h.sparse_lml_grad()
    dt = (time.perf_counter() - t0) / 20
    h.set_option("profile", 1); h.profile_reset()
    for _ in range(5):
        h.sparse_lml_grad()
    ms, n = h.profile_get("SYRK")
    h.set_option("profile", 0)
    print("overlap %d: bound+grad %.3f ms, Phi kernel %.2f ms" % (ov, dt * 1e3, ms / n), flush=True)
