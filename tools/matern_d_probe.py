"""Matern vs RBF bound+gradient evaluation at D on both sides of the fused-epilogue limit (32)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M = 100000, 2048
for D in (32, 40, 64):
    X, y, Z = bench.synthetic(N, D, M, seed=5)
    h.set_data(X, y); h.set_inducing(Z)
    for kind in ("RBF", "Matern52"):
        h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
        for _ in range(3):
            h.sparse_lml_grad()
        t0 = time.perf_counter()
        for _ in range(5):
            h.sparse_lml_grad()
        tg = (time.perf_counter() - t0) / 5 * 1e3
        print("D %2d %-8s bound+grad %.2f ms (form %d)" % (D, kind, tg, int(h.get("MODE", 3)[0])), flush=True)
