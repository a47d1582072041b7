"""Phi syrk time vs split count at the config shapes (auto = 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
for name, N, M, cands in [("Cm", 100000, 2048, (0, 15, 30)), ("C2", 100000, 1024, (0, 14, 28, 56)),
                          ("C5 chunk", 131072, 4096, (0, 3, 15, 16, 19, 29, 31)), ("M512", 20000, 512, (0, 51, 102, 204))]:
    for s in cands:
        best = min(h.dbg_syrk_variant(N, M, 0, s, 5)[0] for _ in range(2))
        print("%-9s N %7d M %5d splits %3d: %.3f ms -> %.1f TFLOP/s" % (name, N, M, s, best, N * M * (M + 1.0) / best * 1e-9), flush=True)
