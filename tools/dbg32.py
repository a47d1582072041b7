import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M, D = 3000, 256, 8
X, y, Z = bench.synthetic(N, D, M, seed=5)
res = {}
for prec in (64, 32):
    h.set_option("precision", prec)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.3], np.full((1, D), 0.9 * np.sqrt(D)), 0.05)
    try:
        print(prec, "lml", h.sparse_lml())
    except Exception as e:
        print(prec, "failed", e)
    print(" mode", h.get("MODE", (1,)))
    res[prec] = (h.get("PHI", (M, M)), h.get("BVEC", (M,)), h.get("KNM", (N, M)))
for i, nm in enumerate(["PHI", "BVEC", "KNM"]):
    a, b = res[32][i], res[64][i]
    print(nm, "max abs diff", np.abs(a - b).max(), "scale", np.abs(b).max())
    if nm == "KNM":
        bad = np.argwhere(np.abs(a - b) > 1e-3)
        print(" bad entries", len(bad), bad[:10])
