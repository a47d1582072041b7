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
ref = None
for pb in (0, 1, 0, 1):
    h.set_option("prebuild", pb)
    out = h.sparse_lml_grad(); h.sparse_lml_grad()
    if ref is None: ref = out
    same = all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(ref, out))
    t0 = time.perf_counter()
    for _ in range(10):
        h.sparse_lml_grad()
    dt = (time.perf_counter() - t0) / 10
    h.set_option("profile", 1); h.profile_reset()
    for _ in range(5):
        h.sparse_lml_grad()
    ms, n = h.profile_get("SYRK")
    h.set_option("profile", 0)
    print("prebuild %d: bound+grad %.2f ms, Phi kernel %.2f ms inside it, results identical to the first run: %s" % (pb, dt * 1e3, ms / n, same), flush=True)
