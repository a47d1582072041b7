"""Soak over the evaluation paths added late in the round: stable form (paired zero-skipping product, W^T y from
the epilogue), phantom-padded inducing sets, the (M+1)-row bound-only factorisation, posterior reuse, chunked rows.
Repeated evaluations of one configuration must be bitwise identical; posterior-from-gradient must equal the full
posterior evaluation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
configs = [(60000, 1000, 16, "RBF", 0), (40000, 1500, 12, "Matern52", 0), (50000, 1024, 8, "RBF", 2), (30000, 700, 5, "Matern32", 0),
           (45000, 2048, 32, "RBF", 1), (70000, 1100, 24, "RBF", 0)]
t0 = time.perf_counter()
bad = 0
for ci, (N, M, D, kind, mode) in enumerate(configs):
    X, y, Z = bench.synthetic(N, D, M, seed=20 + ci)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_option("stats_mode", mode)
    ref = {}
    for i in range(n):
        s = 1.0 + 1e-3 * ((i % 3) - 1)
        h.set_option("chunk_rows", (131072, 20000)[(i // 3) % 2])
        h.set_kernel([kind], [1.0 * s], np.full((1, D), np.sqrt(D) * s), 0.02 * s)
        which = i % 4
        if which == 0:
            out = (h.sparse_lml(),)
        elif which == 1:
            out = h.sparse_lml_grad()
        elif which == 2:
            g = h.sparse_lml_grad()
            lp = h.sparse_posterior()                     # from the gradient evaluation's factors
            out = (g[0], lp, h.get("WV", M), h.predict(X[:256])[0])
        else:
            h.set_option("posterior_reuse", 0)
            lp = h.sparse_posterior()                     # full evaluation
            h.set_option("posterior_reuse", 1)
            out = (lp, lp, h.get("WV", M), h.predict(X[:256])[0])
        key = (i % 3, (i // 3) % 2, which if which < 2 else 2)
        if which >= 2:
            out = out[1:]                                  # (bound, wv, mean): shortcut and full evaluation share a key
        if key in ref:
            same = all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(ref[key], out))
            if not same:
                bad += 1
                print("MISMATCH config", ci, "iteration", i, key, [float(np.abs(np.asarray(u) - np.asarray(v)).max()) for u, v in zip(ref[key], out)], flush=True)
        else:
            ref[key] = out
    print("config %d (N %d, M %d, D %d, %s, stats_mode %d): ran form %d, %d evaluations ok" % (ci, N, M, D, kind, mode, int(h.get("MODE", 3)[0]), n), flush=True)
h.set_option("stats_mode", 0); h.set_option("chunk_rows", 131072)
print("soak_modes: %d mismatches in %.1f s" % (bad, time.perf_counter() - t0))
