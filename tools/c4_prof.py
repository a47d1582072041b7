"""C4 unit of work under rocprofv3: one candidate projection (d_c given), 8 length-scales."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
from edrgp_amd.sweep import projection_sweep
h = HipGP(0)
N, M, D = 200000, 1024, 64
dc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)); y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Zr = rng.permutation(N)[:M]
W = np.linalg.qr(rng.standard_normal((D, dc)))[0]
projection_sweep(X, y, Zr, [W], n_scales=2, want_gram=True, hip=h)
t0 = time.perf_counter()
projection_sweep(X, y, Zr, [W], n_scales=8, want_gram=True, hip=h)
print("d_c %d: %.2f ms per evaluation" % (dc, (time.perf_counter() - t0) / 8 * 1e3))
