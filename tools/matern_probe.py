"""RBF vs Matern at the C3 shard shape: per-evaluation time and the device time of the kernel build and of the
gradient contraction (HIP events of the library's profile scopes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 125000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=1000)
h = HipGP(0)
h.set_data(X, y); h.set_inducing(Z)
for kind in ("RBF", "Matern52", "Matern32"):
    h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    for _ in range(4):
        h.sparse_lml_grad()
    t0 = time.perf_counter()
    for _ in range(8):
        h.sparse_lml_grad()
    dt = (time.perf_counter() - t0) / 8 * 1e3
    h.set_option("profile", 1); h.profile_reset()
    for _ in range(4):
        h.sparse_lml_grad()
    out = []
    for key in ("KBUILD", "SYRK", "GRADGEMM", "SMALL"):
        ms, n = h.profile_get(key)
        out.append("%s %.2f ms" % (key, ms / 4))
    h.set_option("profile", 0)
    h.sparse_posterior()
    t0 = time.perf_counter()
    for _ in range(4):
        h.predict_train(True, True)
    tp = (time.perf_counter() - t0) / 4 * 1e3
    print("%-9s bound+grad %.2f ms | per evaluation: %s | predict+gradient on N rows %.2f ms" % (kind, dt, ", ".join(out), tp), flush=True)
