import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M = 100000, 2048
_, p0 = h.dbg_syrk_variant(4096, 512, 0, 0, 1, want_phi=True)
_, p3 = h.dbg_syrk_variant(4096, 512, 3, 0, 1, want_phi=True)
print("glds vs production Phi (4096 x 512): max rel diff", np.abs(p0 - p3).max() / np.abs(p0).max(), flush=True)
for name, n, m in [("Cm", 100000, 2048), ("C2", 100000, 1024), ("C5 chunk", 131072, 4096)]:
    for v in (0, 3, 0, 3):
        ms = min(h.dbg_syrk_variant(n, m, v, 0, 5)[0] for _ in range(2))
        print("%-9s variant %d: %.3f ms -> %.1f TFLOP/s" % (name, v, ms, n * m * (m + 1.0) / ms * 1e-9), flush=True)
