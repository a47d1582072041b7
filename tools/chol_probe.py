import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
h.set_option("overlap", 0)
for M, D in [(2048, 32), (4096, 64), (8192, 64)]:
    N = 16384
    X, y, Z = bench.synthetic(N, D, M, seed=3)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    for fused in (0, 1, 0, 1):
        h.set_option("tri_tile", 64 if fused else 128)
        h.set_option("profile", 1)
        h.sparse_lml(); h.profile_reset()
        for _ in range(3):
            v = h.sparse_lml()
        ms, n = h.profile_get("SMALL")
        h.set_option("profile", 0)
        print("M=%d auto_tri_tile(0 = forced 128 below 1024 tiles)=%d: O(M^3) phase %.3f ms per bound evaluation (lml %.4f)" % (M, fused, ms / 3, v), flush=True)
