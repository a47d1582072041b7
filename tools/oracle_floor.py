"""How far apart are two fp64 evaluations of the SAME bound?  The oracle on inputs whose last bit is perturbed at random
(X (1 + 2^-52 s), s in {0, 1}), at a BASELINE shape: the relative change of the bound and of the posterior mean is the
floor below which a difference between two implementations says nothing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gp_oracle as go
from threadpoolctl import threadpool_limits
from edrgp_amd.utils import effective_cpus
name = sys.argv[1] if len(sys.argv) > 1 else "C5c"
CFG = {"C5c": (131072, 4096, 64, "RBF"), "Cm": (100000, 2048, 32, "RBF"), "C2": (100000, 1024, 16, "RBF")}
N, M, D, kind = CFG[name]
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D))
y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Z = X[rng.permutation(N)[:M]].copy()
kern = go.make_kernel([kind], D, [dict(variance=1.0, lengthscale=float(np.sqrt(D)))])
with threadpool_limits(limits=effective_cpus()):
    out = []
    for trial in range(3):
        r = np.random.default_rng(100 + trial)
        Xp = X * (1.0 + 2.0 ** -52 * r.integers(0, 2, X.shape)) if trial else X
        Zp = Z * (1.0 + 2.0 ** -52 * r.integers(0, 2, Z.shape)) if trial else Z
        t0 = time.perf_counter()
        inf = go.vardtc_inference(kern, Xp, Zp, 0.01, y[:, None])
        mu = (inf["psi1"][:4096] @ inf["woodbury_vector"])[:, 0]
        out.append((float(inf["log_marginal"][0, 0]), mu))
        print("trial %d: bound %.9f (%.1f s)" % (trial, out[-1][0], time.perf_counter() - t0), flush=True)
for t in (1, 2):
    print("%s last-bit perturbation %d: bound rel %.2e, mean rel %.2e" % (name, t, abs(out[t][0] - out[0][0]) / abs(out[0][0]),
          np.abs(out[t][1] - out[0][1]).max() / np.abs(out[0][1]).max()))
