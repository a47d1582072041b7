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
for ov in (3, 2, 1, 0, 3, 2):
    h.set_option("overlap", ov)
    h.sparse_lml_grad(); h.sparse_lml_grad()
    h.set_option("profile", 1); h.profile_reset()
    t0 = time.perf_counter()
    for _ in range(10):
        h.sparse_lml_grad()
    dt = (time.perf_counter() - t0) / 10
    ms, n = h.profile_get("SYRK")
    h.set_option("profile", 0)
    t0 = time.perf_counter()
    for _ in range(10):
        h.sparse_lml()
    dl = (time.perf_counter() - t0) / 10
    print("overlap %d: bound+grad %.2f ms (Phi kernel %.2f ms inside it), bound %.2f ms" % (ov, dt * 1e3, ms / n, dl * 1e3), flush=True)
