import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
h.set_option("syrk_diag_opt", 0)
_, p0 = h.dbg_syrk_variant(8192, 640, 0, 0, 1, want_phi=True)
h.set_option("syrk_diag_opt", 1)
_, p1 = h.dbg_syrk_variant(8192, 640, 0, 0, 1, want_phi=True)
print("diag_opt vs full diagonal tiles: max abs diff", np.abs(p0 - p1).max(), "symmetric", np.array_equal(p1, p1.T), flush=True)
for name, n, m in [("Cm", 100000, 2048), ("C2", 100000, 1024), ("C5 chunk", 131072, 4096)]:
    for opt in (0, 1, 0, 1):
        h.set_option("syrk_diag_opt", opt)
        ms = min(h.dbg_syrk_variant(n, m, 0, 0, 5)[0] for _ in range(2))
        print("%-9s diag_opt %d: %.3f ms -> %.1f TFLOP/s" % (name, opt, ms, n * m * (m + 1.0) / ms * 1e-9), flush=True)

h.set_option("syrk_diag_opt", 1)
for name, n, m, cands in [("Cm", 100000, 2048, (0, 12, 16)), ("C2", 100000, 1024, (0, 16, 32, 48, 64)), ("C3", 125008, 2048, (0, 12, 16, 20)), ("C5 chunk", 131072, 4096, (0, 8))]:
    for sp in cands:
        ms = min(h.dbg_syrk_variant(n, m, 0, sp, 5)[0] for _ in range(2))
        print("%-9s diag_opt 1 splits %2d: %.3f ms -> %.1f TFLOP/s" % (name, sp, ms, n * m * (m + 1.0) / ms * 1e-9), flush=True)
