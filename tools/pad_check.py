import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp_oracle as go
from edrgp_amd.backend import HipGP
hip = HipGP(0)
for (N, M, D, kinds, mode) in [(1200, 300, 4, ["RBF"], 0), (1500, 449, 6, ["Matern52"], 0), (1300, 257, 3, ["RBF", "Matern32"], 0), (1400, 390, 5, ["RBF"], 2)]:
    X, y, Z = go.synthetic_problem(N, D, M, seed=N + M)
    opts = [dict(variance=0.9 + 0.2 * i, lengthscale=(1.5 + i) * np.sqrt(D)) for i in range(len(kinds))]
    kern = go.make_kernel(kinds, D, opts)
    noise = 0.05
    hip.set_data(X, y); hip.set_inducing(Z)
    hip.set_kernel(kinds, [o["variance"] for o in opts], np.array([np.full(D, o["lengthscale"]) for o in opts]), noise)
    hip.set_option("stats_mode", mode)
    inf = go.vardtc_inference(kern, X, Z, noise, y[:, None])
    kg, dn, dZ = go.sparse_gradients(kern, X, Z, inf)
    cond = np.linalg.cond(kern.K(Z) + 1e-8 * np.eye(M))
    for pad in (0, 1):
        hip.set_option("pad_inducing", pad)
        out = hip.sparse_lml_grad()
        print(N, M, D, kinds, mode, "pad", pad, "cond %.1e" % cond, "lml rel %.1e" % (abs(out[0] - inf["log_marginal"][0, 0]) / abs(inf["log_marginal"][0, 0])),
              "dZ err %.2e (max |dZ| %.2e)" % (np.abs(out[4] - dZ).max(), np.abs(dZ).max()), "mode", hip.get("MODE", 3))
    hip.set_option("stats_mode", 0)
