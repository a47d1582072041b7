"""fp32 gradient row pass (precision = 32) at a config-5-like size under option groups: bound+gradient times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 1000000, 4096, 64
X, y, Z = bench.synthetic(N, D, M, seed=1)
for group in (sys.argv[1:] or [""]):
    h = HipGP(0)
    h.set_option("precision", 32)
    for kv in filter(None, group.split(",")):
        k, v = kv.split("=")
        h.set_option(k, float(v))
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    h.sparse_lml_grad()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); h.sparse_lml_grad(); ts.append((time.perf_counter() - t0) * 1e3)
    print("[%s] bound+grad %s ms" % (group, " ".join("%.1f" % t for t in ts)), flush=True)
    h.close()
