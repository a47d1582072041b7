"""Gram form vs stable form of the row statistics against conditioning estimates of Kmm (which rule should pick the form?):
the pivot-ratio proxy (max / min diag Lm)^2, the O(M^2) estimate  kappa_F = tr(Kmm) |Lm^-1|_F^2  (>= cond_2, within a
factor M), the true 2-norm condition number (host eigvalsh), and the differences of the two forms in the bound and the
posterior mean on 4096 rows."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
CFG = [("C2", 100000, 1024, 16, "RBF", 1.0), ("Cm", 100000, 2048, 32, "RBF", 1.0), ("C3s", 125000, 2048, 32, "Matern52", 1.0),
       ("C5c", 131072, 4096, 64, "RBF", 1.0), ("Cm x1.5", 100000, 2048, 32, "RBF", 1.5), ("Cm x0.7", 100000, 2048, 32, "RBF", 0.7),
       ("C2 x0.7", 100000, 1024, 16, "RBF", 0.7), ("C5c x0.8", 131072, 4096, 64, "RBF", 0.8), ("M3072", 100000, 3072, 32, "RBF", 1.0)]
h = HipGP(0)
for name, N, M, D, kind, lsf in CFG:
    X, y, Z = bench.synthetic(N, D, M, seed=0)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel([kind], [1.0], np.full((1, D), lsf * np.sqrt(D)), 0.01)
    res = {}
    for mode in (1, 2):
        h.set_option("stats_mode", mode)
        lml = h.sparse_posterior()
        mu, _ = h.predict(X[:4096])
        res[mode] = (lml, mu)
    h.set_option("stats_mode", 0)
    info = h.get("MODE", (4,))
    Li = h.get("LMI", (M, M))
    Kmm = h.get("KMM", (M, M))
    kF = np.trace(Kmm) * np.sum(Li * Li)
    w = np.linalg.eigvalsh(Kmm)
    print("%-9s M=%d D=%d %-8s proxy %8.1f  kappa_F %.3e  cond_2 %.3e | gram vs stable: bound %.2e  mean %.2e"
          % (name, M, D, kind, info[2], kF, w[-1] / w[0], abs(res[1][0] - res[2][0]) / abs(res[2][0]),
             np.abs(res[1][1] - res[2][1]).max() / np.abs(res[2][1]).max()), flush=True)
h.close()
