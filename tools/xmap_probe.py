import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
h.set_option("syrk_xmap", 0)
_, p0 = h.dbg_syrk_variant(8192, 640, 0, 0, 1, want_phi=True)
h.set_option("syrk_xmap", 1)
_, p1 = h.dbg_syrk_variant(8192, 640, 0, 0, 1, want_phi=True)
print("xmap vs tile-major Phi: max abs diff", np.abs(p0 - p1).max(), flush=True)
for name, n, m in [("Cm", 100000, 2048), ("C2", 100000, 1024), ("C5 chunk", 131072, 4096)]:
    for xm in (0, 1, 0, 1):
        h.set_option("syrk_xmap", xm)
        ms = min(h.dbg_syrk_variant(n, m, 0, 0, 5)[0] for _ in range(2))
        print("%-9s xmap %d: %.3f ms -> %.1f TFLOP/s" % (name, xm, ms, n * m * (m + 1.0) / ms * 1e-9), flush=True)
