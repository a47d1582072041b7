"""Per-rank evaluation time against the local row count at the metric's M and D (what a strong-scaling rank of
bench.py --scaling strong computes, without the exchange): bound-only, bound+gradient, predict on the local rows.
The fit t(N) = a + b N gives the rows-independent (replicated O(M^3) + latency) part a and the per-row cost b that the
time model of DESIGN section 7 uses."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
M, D = 2048, 32
h = HipGP(0)
Xf, yf, Z = bench.synthetic(100000, D, M, seed=bench.RANK0_SEED)
res = []
for G in (1, 2, 4, 8):
    n = 100000 // G
    h.set_data(Xf[:n], yf[:n]); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    out = []
    for f in (h.sparse_lml, h.sparse_lml_grad):
        f(); f(); h.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                h.set_inducing(Z)
                f()
            h.sync()
            best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        out.append(best)
    h.sparse_posterior()
    t0 = time.perf_counter()
    for _ in range(5):
        h.predict_train(True, True)
    tp = (time.perf_counter() - t0) / 5 * 1e3
    res.append((n, out[0], out[1], tp))
    print("rows %6d (1/%d of 100k): bound %.3f ms, bound+grad %.3f ms, predict+predict_gradient %.2f ms" % (n, G, out[0], out[1], tp), flush=True)
ns = np.array([r[0] for r in res], dtype=float)
for name, col in (("bound", 1), ("bound+grad", 2)):
    t = np.array([r[col] for r in res])
    b, a = np.polyfit(ns, t, 1)
    print("%s: t(N) = %.3f ms + %.3f us/row x N   (fit over the four sizes; residuals %s)" % (
        name, a, b * 1e3, " ".join("%+.2f" % v for v in (t - (a + b * ns)))))
h.close()
