"""Kernel-matrix build and whole evaluation at feature counts on and off the 16-wide k-tile."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M = 100000, 2048
for D in (16, 15, 10, 5, 32, 30, 20, 48, 40):
    X, y, Z = bench.synthetic(N, D, M, seed=5)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    for _ in range(3):
        h.sparse_lml_grad()
    t0 = time.perf_counter()
    for _ in range(5):
        h.sparse_lml_grad()
    tg = (time.perf_counter() - t0) / 5 * 1e3
    h.set_option("profile", 1); h.profile_reset()
    for _ in range(4):
        h.sparse_lml_grad()
    kb, nk = h.profile_get("KBUILD"); mg, ng = h.profile_get("GRADGEMM")
    h.set_option("profile", 0)
    print("D %2d: bound+grad %.2f ms | K_nm build scope %.2f ms | gradient contraction scope %.2f ms" % (D, tg, kb / 4, mg / 4), flush=True)
