"""precision = 32 at the config-5 chunk shape: are repeated evaluations bit for bit the same, within and across calls?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M, D = 131072, 4096, 64
X, y, Z = bench.synthetic(N, D, M, seed=11)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
res = {}
for prec in (64, 32, 64, 32):
    h.set_option("precision", prec)
    out = []
    for r in range(3):
        h.set_inducing(Z)
        g = h.sparse_lml_grad()
        out.append((g[0], g[1].copy(), g[2].copy(), g[3], g[4].copy()))
    same = all(out[0][0] == o[0] and np.array_equal(out[0][4], o[4]) and np.array_equal(out[0][2], o[2]) for o in out[1:])
    print("precision %d: three evaluations bit for bit equal: %s; mode %s" % (prec, same, h.get("MODE", (4,))[:1]))
    if prec in res:
        a, b = res[prec], out[0]
        print("   vs the first block at this precision: bound equal %s, dZ equal %s (max rel diff %.2e), dlen rel diff %.2e"
              % (a[0] == b[0], np.array_equal(a[4], b[4]), np.abs(a[4] - b[4]).max() / np.abs(a[4]).max(),
                 np.abs(a[2] - b[2]).max() / np.abs(a[2]).max()))
    res[prec] = out[0]
d64, d32 = res[64], res[32]
print("fp32 vs fp64: dZ %.2e, dlen %.2e, dvar %.2e" % (np.abs(d32[4] - d64[4]).max() / np.abs(d64[4]).max(),
      np.abs(d32[2] - d64[2]).max() / np.abs(d64[2]).max(), np.abs(d32[1] - d64[1]).max() / np.abs(d64[1]).max()))
