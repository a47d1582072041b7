"""One case of tools/fuzz.py's generator under the A/B switches (which path is responsible for an error?)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import gp_oracle as go
from edrgp_amd.backend import HipGP
seed, want = int(sys.argv[1]), int(sys.argv[2])
kinds_all = ["RBF", "Matern32", "Matern52"]
rng = np.random.default_rng(seed)
for case in range(want + 1):
    D = int(rng.integers(1, 10)) if case % 5 else int(rng.integers(33, 45))
    M = int(rng.integers(1, 200))
    N = int(rng.integers(max(M, 2), 900))
    if case % 7 == 3:
        M = int(rng.integers(257, 520))
        N = int(rng.integers(M, 1600))
    nparts = 1 if rng.random() < 0.7 else 2
    kinds = [kinds_all[int(rng.integers(0, 3))] for _ in range(nparts)]
    opts = [dict(variance=float(rng.uniform(0.3, 2.0)), lengthscale=float(rng.uniform(0.8, 3.0) * np.sqrt(D))) for _ in range(nparts)]
    noise = float(10 ** rng.uniform(-2, 0))
    chunk = int(rng.choice([64, 100, 257, 131072]))
    mode = int(rng.choice([0, 0, 1, 2]))
X, y, Z = go.synthetic_problem(N, D, M, seed=1000 + want)
kern = go.make_kernel(kinds, D, opts)
inf = go.vardtc_inference(kern, X, Z, noise, y[:, None])
ref = inf["log_marginal"][0, 0]
print(dict(N=N, M=M, D=D, kinds=kinds, noise=noise, chunk=chunk, mode=mode), "cond %.2e" % np.linalg.cond(kern.K(Z) + 1e-8 * np.eye(M)), "ref", ref)
h = HipGP(0)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(kinds, [o["variance"] for o in opts], np.array([np.full(D, o["lengthscale"]) for o in opts]), noise)
for sm in (0, 1, 2):
    for pad in (0, 1):
        for rhs in (0, 1):
            for ch in (chunk, 131072):
                h.set_option("stats_mode", sm); h.set_option("pad_inducing", pad); h.set_option("chol_rhs", rhs); h.set_option("chunk_rows", ch)
                a = h.sparse_lml(); b = h.sparse_lml_grad()[0]
                md = h.get("MODE", 3)
                print("stats_mode %d pad %d chol_rhs %d chunk %6d: bound-only rel err %.2e, gradient call %.2e | ran mode %d, pivot ratio^2 %.1f"
                      % (sm, pad, rhs, ch, abs(a - ref) / abs(ref), abs(b - ref) / abs(ref), md[0], md[2]), flush=True)
