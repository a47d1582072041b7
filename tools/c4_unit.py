"""One C4 candidate (N=200k, M=1024, d_c given, 16 length-scales) through edrgp_sweep_scales: wall time per unit.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (the first sweep is the warm-up)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
from edrgp_amd.sweep import projection_sweep

dc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "RBF"
h = HipGP(0)
N, M, D = 200000, 1024, 64
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)); y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Zr = rng.permutation(N)[:M]
W = np.linalg.qr(rng.standard_normal((D, dc)))[0]
kw = dict(n_scales=16, want_gram=True, hip=h)
if kind != "RBF":
    kw["kind"] = kind
projection_sweep(X, y, Zr, [W], **kw)
for r in range(reps):
    t0 = time.perf_counter()
    out = projection_sweep(X, y, Zr, [W], **kw)
    dt = time.perf_counter() - t0
    print("d_c=%d %s: %.2f ms per unit (16 units %.1f ms), modes %s" % (dc, kind, dt * 1e3 / 16, dt * 1e3, [o.get("form") for o in out]), flush=True)
