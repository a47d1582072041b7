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
Ws = [np.linalg.qr(rng.standard_normal((D, D - c)))[0] for c in (0, 16, 32, 48)]
ref = None
pool = [HipGP(0) for _ in range(4)]
for conc in (1, 2, 3, 4, 1):
    projection_sweep(X, y, Zr, Ws[:1], n_scales=4, want_gram=True, concurrency=max(conc, 2), hips=pool[:conc] if conc > 1 else None, hip=pool[0] if conc == 1 else None)
    t0 = time.perf_counter()
    out = projection_sweep(X, y, Zr, Ws, n_scales=16, want_gram=True, concurrency=max(conc, 2) if conc > 1 else 1, hips=pool[:conc] if conc > 1 else None, hip=pool[0] if conc == 1 else None)
    dt = time.perf_counter() - t0
    b = np.array([o["bound"] for o in out])
    if ref is None:
        ref = b
    print("concurrency %d: %d evaluations in %.2f s -> %.2f ms each (base upload included); bounds identical: %s"
          % (conc, len(out), dt, dt * 1e3 / len(out), np.array_equal(b, ref)), flush=True)
