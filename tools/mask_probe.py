import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=1000)
for side in (0, 8, 16, 24, 32):
    h = HipGP(0)
    h.set_option("side_cus", side)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    v = h.sparse_lml(); h.sparse_lml()
    ts = []
    for i in range(10):
        t0 = time.perf_counter(); h.sparse_lml(); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); h.sparse_lml_grad(); h.sparse_lml_grad(); tg = (time.perf_counter() - t0) * 500
    print("side_cus %2d: lml %.6f  ms %s | grad eval %.1f ms" % (side, v, " ".join("%.2f" % t for t in ts), tg), flush=True)
    h.close()
