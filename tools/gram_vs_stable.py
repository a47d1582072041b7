"""Gram form (the fast path) vs stable form (GPy's operation order) at the BASELINE sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())
for name, N, M, D, kind in [("C2", 100000, 1024, 16, "RBF"), ("Cm", 100000, 2048, 32, "RBF"), ("C3 shard", 125000, 2048, 32, "Matern52")]:
    X, y, Z = bench.synthetic(N, D, M, seed=7)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    out = {}
    for mode in (1, 2):
        h.set_option("stats_mode", mode)
        out[mode] = h.sparse_lml_grad()
        pr = h.get("MODE", 3)[2]
        h.sparse_posterior()
        out[mode] = tuple(out[mode]) + tuple(h.predict_train())
    h.set_option("stats_mode", 0)
    g, s = out[1], out[2]
    print("%-9s proxy %.0f: bound %.6f | gram vs stable: bound %.1e (rel), d/dvariance %.1e, d/dlengthscale %.1e, d/dnoise %.1e, dZ %.1e; posterior mean on the N rows %.1e, its Jacobian %.1e"
          % (name, pr, s[0], abs(g[0] - s[0]) / abs(s[0]), rel(g[1], s[1]), rel(g[2], s[2]), abs(g[3] - s[3]) / abs(s[3]), rel(g[4], s[4]), rel(g[5], s[5]), rel(g[6], s[6])), flush=True)
