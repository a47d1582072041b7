"""Soak of the evaluation path: `reps` bound+gradient evaluations at Cm (and C2, stable form) on re-sent inputs, every
result compared bit for bit with the first; predictions on the training rows likewise.  A data race between the two
streams, a stale workspace or an uninitialised read shows up as a run that differs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
h = HipGP(0)
for name, N, M, D, kind in (("Cm", 100000, 2048, 32, "RBF"), ("C2", 100000, 1024, 16, "RBF"), ("C3s", 125000, 2048, 32, "Matern52")):
    X, y, Z = bench.synthetic(N, D, M, seed=3)
    h.set_data(X, y)
    first = None
    bad = 0
    t0 = time.perf_counter()
    for r in range(reps):
        th = bench.theta_path(r % 7, D, kind)
        h.set_inducing(Z)
        h.set_kernel([kind], [th["variance"]], np.full((1, D), th["lengthscale"]), th["noise"])
        g = h.sparse_lml_grad()
        b = h.sparse_lml()
        key = r % 7
        cur = (g[0], g[1].copy(), g[2].copy(), g[3], g[4].copy(), b)
        if first is None:
            first = {}
        if key not in first:
            first[key] = cur
        else:
            f = first[key]
            same = (cur[0] == f[0] and np.array_equal(cur[1], f[1]) and np.array_equal(cur[2], f[2]) and cur[3] == f[3]
                    and np.array_equal(cur[4], f[4]) and cur[5] == f[5])
            if not same or not np.isfinite(cur[0]):
                bad += 1
                print("%s run %d (theta %d): differs from the first run with these inputs: bound %r vs %r" % (name, r, key, cur[0], f[0]), flush=True)
        if r % 50 == 49:
            h.sparse_posterior()
            mu, jac = h.predict_train(True, True)
            if "pred" not in first:
                first["pred"] = (key, mu, jac)
            elif first["pred"][0] == key and not (np.array_equal(mu, first["pred"][1]) and np.array_equal(jac, first["pred"][2])):
                bad += 1
                print("%s run %d: predictions differ" % (name, r), flush=True)
    print("%s: %d evaluations pairs in %.1f s, %d differ" % (name, reps, time.perf_counter() - t0, bad), flush=True)
