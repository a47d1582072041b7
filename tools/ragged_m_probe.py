"""Round-decimal vs power-of-two M and N: per-evaluation times and the Phi kernel's rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
for (N, M, D) in ((100000, 1024, 16), (100000, 1000, 16), (100000, 2048, 32), (100000, 2000, 32), (100003, 2000, 32), (100000, 1500, 20), (100000, 1536, 20)):
    X, y, Z = bench.synthetic(N, D, M, seed=5)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    for _ in range(3):
        h.sparse_lml_grad()
    t0 = time.perf_counter()
    for _ in range(6):
        h.sparse_lml_grad()
    tg = (time.perf_counter() - t0) / 6 * 1e3
    t0 = time.perf_counter()
    for _ in range(6):
        h.sparse_lml()
    tl = (time.perf_counter() - t0) / 6 * 1e3
    h.set_option("profile", 1); h.profile_reset()
    for _ in range(4):
        h.sparse_lml_grad()
    ms, n = h.profile_get("SYRK"); mg, ng = h.profile_get("GRADGEMM")
    h.set_option("profile", 0)
    print("N %6d M %4d D %2d: bound %.2f ms, bound+grad %.2f ms | Phi %.2f ms = %.1f TFLOP/s | gradient contraction scope %.2f ms = %.1f TFLOP/s"
          % (N, M, D, tl, tg, ms / n, N * M * (M + 1.0) / (ms / n * 1e-3) * 1e-12, mg / 4, 2.0 * N * M * M / (mg / 4 * 1e-3) * 1e-12), flush=True)
