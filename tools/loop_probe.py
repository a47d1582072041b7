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
prof = int(sys.argv[1]) if len(sys.argv) > 1 else 0
h.set_option("profile", prof)
if len(sys.argv) > 2:
    h.set_option("overlap", int(sys.argv[2]))
h.sparse_lml_grad()
ts = []
for i in range(12):
    t0 = time.perf_counter(); h.sparse_lml_grad(); ts.append((time.perf_counter() - t0) * 1e3)
print("profile", prof, "lml_grad ms:", " ".join("%.1f" % t for t in ts))
ts = []
h.sparse_lml()
for i in range(8):
    t0 = time.perf_counter(); h.sparse_lml(); ts.append((time.perf_counter() - t0) * 1e3)
print("lml ms:", " ".join("%.1f" % t for t in ts))
