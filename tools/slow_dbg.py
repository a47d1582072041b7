import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=5)
Z = X[np.random.default_rng(0).permutation(N)[:M]].copy()
h.set_data(X, y); h.set_inducing(Z)
for var, ls, noise in [(1.0, 5.66, 0.01), (1.0, 1.0, 1.0), (0.1, 1.0, 1.3), (1e-3, 1.0, 1.0), (1.0, 5.66, 0.01)]:
    h.set_kernel(["RBF"], [var], np.full((1, D), ls), noise)
    h.sparse_lml_grad()
    h.set_option("trace", 1)
    t0 = time.perf_counter(); out = h.sparse_lml_grad(); dt = time.perf_counter() - t0
    h.set_option("trace", 0)
    print("var %g ls %g noise %g: %.1f ms, lml %.3f mode %d" % (var, ls, noise, dt * 1e3, out[0], h.get("MODE", (1,))[0]), flush=True)
