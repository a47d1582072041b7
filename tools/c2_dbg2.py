import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M, D = 100000, 1024, 16
X, y, Z = bench.synthetic(N, D, M, seed=7)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
h.set_option("trace", 1)
for i in range(8):
    t0 = time.perf_counter(); v = h.sparse_lml(); t1 = time.perf_counter()
    print("== call", i, "lml %.3f ms" % ((t1 - t0) * 1e3), flush=True)
