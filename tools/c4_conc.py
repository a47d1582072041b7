"""BASELINE config 4 at its real size (N = 200 000, M = 1024, D = 64): ms per (bound + posterior + gradient Gram)
unit over the 16 length-scales of one candidate, by number of concurrent native contexts."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
from edrgp_amd.sweep import projection_sweep

N, M, D = 200000, 1024, 64
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)); y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Zr = rng.permutation(N)[:M]
for dc in (64, 32):
    Ws = [np.linalg.qr(rng.standard_normal((D, dc)))[0] for _ in range(3)]
    for conc in (1, 2, 3):
        hips = [HipGP(0) for _ in range(conc)]
        projection_sweep(X, y, Zr, Ws[:1], n_scales=2 * conc, want_gram=True, concurrency=conc, hips=hips if conc > 1 else None, hip=hips[0])
        t0 = time.perf_counter()
        out = projection_sweep(X, y, Zr, Ws[1:], n_scales=16, want_gram=True, concurrency=conc, hips=hips if conc > 1 else None, hip=hips[0])
        dt = time.perf_counter() - t0
        print("C4 d_c=%d, %d context(s): %.2f ms per unit (%d units)" % (dc, conc, dt * 1e3 / len(out), len(out)), flush=True)
        for h in hips:
            h.close()
