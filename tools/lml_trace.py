"""Timeline of a bound-only evaluation at Cm (trace marks of the library + wall clock)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=1000)
h = HipGP(0)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for _ in range(5):
    h.sparse_lml()
t0 = time.perf_counter()
for _ in range(20):
    h.sparse_lml()
print("bound-only: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
h.set_option("trace", 1)
for _ in range(3):
    h.sparse_lml()
    print("----", flush=True)
h.set_option("trace", 0)
