"""Repeats the fused factorisation + inverse at a few sizes and checks every run against the first (bit for bit) and
against LAPACK: a data race in the step kernels shows up as a run that differs."""
import os, sys
import numpy as np
import scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
rng = np.random.default_rng(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for n in (3072, 2048, 3100, 1000):
    Q = rng.standard_normal((n, n)); A = Q @ Q.T / n + np.eye(n)
    ref = sla.solve_triangular(np.linalg.cholesky(A), np.eye(n), lower=True)
    L0 = Li0 = None
    bad = 0
    for r in range(reps):
        L, Li, info = h.dbg_potrf_inv(A)
        Lp, _ = h.dbg_potrf(A)
        e_ref = np.abs(Li - ref).max() / np.abs(ref).max()
        ok = info == 0 and np.array_equal(L, Lp) and np.all(np.isfinite(Li)) and e_ref < 1e-10
        if L0 is None:
            L0, Li0 = L, Li
        same = np.array_equal(L, L0) and np.array_equal(Li, Li0)
        if not (ok and same):
            bad += 1
            d = np.argwhere(Li != Li0)
            print("n=%d run %d: info %d, L==potrf %s, finite %s, err vs LAPACK %.2e, same as run 0: %s; first differing inverse entries %s"
                  % (n, r, info, np.array_equal(L, Lp), np.all(np.isfinite(Li)), e_ref, same, d[:4].tolist()), flush=True)
    print("n=%d: %d of %d runs differ" % (n, bad, reps), flush=True)
