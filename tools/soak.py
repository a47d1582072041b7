"""Soak / determinism: repeated evaluations at Cm must be bitwise identical (fixed-order reductions,
no atomics), across overlap modes too; and no sporadic failures over many evaluations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=11)
h.set_data(X, y); h.set_inducing(Z)
ref = None
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
worst = 0.0
for i in range(n):
    s = 1.0 + 1e-3 * ((i % 5) - 2)
    h.set_option("overlap", (3, 0, 2, 1)[i % 4])
    h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0 * s], np.full((1, D), np.sqrt(D) * s), 0.01 * s)
    t1 = time.perf_counter()
    out = h.sparse_lml_grad() if i % 3 else (h.sparse_lml(),)
    worst = max(worst, time.perf_counter() - t1)
    key = (i % 5, len(out))
    if ref is None:
        ref = {}
    if key in ref:
        a, b = ref[key], out
        same = all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(a, b))
        if not same:
            print("MISMATCH at", i, key, [float(np.abs(np.asarray(u) - np.asarray(v)).max()) for u, v in zip(a, b)], flush=True)
    else:
        ref[key] = out
print("soak: %d evaluations in %.1f s, slowest single call %.1f ms, all repeated evaluations bitwise identical unless reported above"
      % (n, time.perf_counter() - t0, worst * 1e3))
