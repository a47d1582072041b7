import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=3)
h.set_option("overlap", 0)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for dbg in (128, 64, 128, 64):
    h.set_option("kbuild_tile", dbg)
    h.set_option("profile", 1)
    try:
        h.sparse_lml()
    except Exception:
        pass
    h.profile_reset()
    for _ in range(3):
        try:
            h.sparse_lml()
        except Exception:
            pass
    ms, n = h.profile_get("KBUILD")
    h.set_option("profile", 0)
    print("kbuild_tile %d: K_nm build slot %.3f ms" % (dbg, ms / 3), flush=True)
