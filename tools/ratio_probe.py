import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
for name, N, M, D, kind in [("C2", 100000, 1024, 16, "RBF"), ("Cm", 100000, 2048, 32, "RBF"), ("C3", 125000, 2048, 32, "Matern52"), ("C5-ish", 200000, 4096, 64, "RBF")]:
    X, y, Z = bench.synthetic(N, D, M, seed=1000)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    h.sparse_lml()
    md = h.get("MODE", (3,))
    print(name, "mode", int(md[0]), "pivot ratio^2 %.3g" % md[2], flush=True)
