"""K_nm build at Cm: full, without the exp (kbuild_dbg 1), without the store (2), without both (3); tile 64 / 128."""
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
for tile in (64, 128):
    for dbg in (0, 1, 2, 3, 0):
        h.set_option("kbuild_tile", tile)
        h.set_option("kbuild_dbg", dbg)
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
        print("tile %3d dbg %d (1 = no exp, 2 = no store): K_nm build scope %.3f ms" % (tile, dbg, ms / 3), flush=True)
h.set_option("kbuild_dbg", 0)
