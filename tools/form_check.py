"""One large shape where the Gram and the stable form differ most in the shape fuzz (N = 70 645, M = 4 216, D = 32,
Matern-5/2): which of the two is right?  Both forms and the CPU oracle on the same inputs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
from oracle import gp_oracle as go

N, M, D, kind, ls, var, noise = 70645, 4216, 32, "Matern52", 8.943, 1.426, 0.0318
rng = np.random.default_rng(123)
X = rng.standard_normal((N, D))
y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Z = X[rng.permutation(N)[:M]].copy()
h = HipGP(0)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel([kind], [var], np.full((1, D), ls), noise)
out = {}
for mode in (1, 2, 0):
    h.set_option("stats_mode", mode)
    out[mode] = h.sparse_lml()
    print("stats_mode %d (ran %d, proxy %.1f): bound %.9f" % (mode, int(h.get("MODE", (4,))[0]), h.get("MODE", (4,))[2], out[mode]), flush=True)
h.set_option("stats_mode", 0)
t0 = time.perf_counter()
kern = go.make_kernel([kind], D, [dict(variance=var, lengthscale=ls)])
ref = go.vardtc_inference(kern, X, Z, noise, y[:, None])["log_marginal"][0, 0]
print("oracle %.9f (%.0f s)" % (ref, time.perf_counter() - t0))
sc = max(abs(ref), float(N))
print("Gram form vs oracle %.2e, stable form vs oracle %.2e, Gram vs stable %.2e (relative to max(|bound|, N))"
      % (abs(out[1] - ref) / sc, abs(out[2] - ref) / sc, abs(out[1] - out[2]) / sc))
