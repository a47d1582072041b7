"""Host-side phases of a bound+gradient evaluation in the bench loop (option "trace" prints them to stderr) and the
Python-side time between two evaluations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=bench.RANK0_SEED)
h = HipGP(0)
h.set_data(X, y)
for i in range(4):
    th = bench.theta_path(i, D, "RBF")
    h.set_inducing(Z); h.set_kernel(["RBF"], [th["variance"]], np.full((1, D), th["lengthscale"]), th["noise"]); h.sparse_lml_grad()
h.set_option("trace", 1)
for i in range(2):
    th = bench.theta_path(i, D, "RBF")
    t0 = time.perf_counter()
    h.set_inducing(Z)
    t1 = time.perf_counter()
    h.set_kernel(["RBF"], [th["variance"]], np.full((1, D), th["lengthscale"]), th["noise"])
    t2 = time.perf_counter()
    h.sparse_lml_grad()
    t3 = time.perf_counter()
    print("python: set_inducing %.3f ms, set_kernel %.3f ms, sparse_lml_grad %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), file=sys.stderr)
