"""Bound / bound+gradient times at Cm and C2 under option settings given as key=value[,key=value...] groups:
    python tools/opt_times.py kmm_early=0 kmm_early=1 kmm_early=1,chol_cap_grid=0"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
CFG = {"Cm": (100000, 2048, 32, "RBF"), "C2": (100000, 1024, 16, "RBF"), "C3s": (125000, 2048, 32, "Matern52"),
       "C3w": (1000000, 2048, 32, "Matern52"), "C4u": (200000, 1024, 64, "RBF")}
names = os.environ.get("CONFIGS", "Cm,C2").split(",")
for name in names:
    N, M, D, kind = CFG[name]
    X, y, Z = bench.synthetic(N, D, M, seed=bench.RANK0_SEED)
    for group in (sys.argv[1:] or [""]):
        h = HipGP(0)
        for kv in filter(None, group.split(",")):
            k, v = kv.split("=")
            h.set_option(k, float(v))
        h.set_data(X, y); h.set_inducing(Z)
        h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
        out = []
        for f in (h.sparse_lml, h.sparse_lml_grad):
            f(); f(); h.sync()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(10 if N <= 200000 else 3):
                    h.set_inducing(Z)
                    r = f()
                h.sync()
                best = min(best, (time.perf_counter() - t0) / (10 if N <= 200000 else 3) * 1e3)
            out.append(best)
        lml = r[0] if isinstance(r, tuple) else r
        print("%s [%s]: bound %.3f ms, bound+grad %.3f ms  (lml %.10g)" % (name, group, out[0], out[1], float(np.ravel(lml)[0])), flush=True)
        h.close()
