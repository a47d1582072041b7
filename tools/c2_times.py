"""C2 / C4-scale evaluation times for A/B runs of the stable-form product (EDRGP_LIB selects the build)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP

h = HipGP(0)

def timeit(f, n=10):
    for _ in range(6):
        f()
    h.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3

for name, N, M, D in [("C2", 100000, 1024, 16), ("C4 unit shape", 200000, 1024, 64)]:
    X, y, Z = bench.synthetic(N, D, M, seed=7)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    for mode in (0, 1, 2):
        h.set_option("stats_mode", mode)
        v = h.sparse_lml()
        t_l = timeit(h.sparse_lml)
        t_g = timeit(h.sparse_lml_grad)
        print("%-14s stats_mode=%d (ran %d): bound %.3f ms, bound+grad %.3f ms, value %.10e" %
              (name, mode, int(h.get("MODE", (1,))[0]), t_l, t_g, v), flush=True)
    h.set_option("stats_mode", 0)
