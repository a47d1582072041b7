"""Phi syrk time vs split count at the config shapes (auto = 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
for name, N, M, cands in [("Cm", 100000, 2048, (0, 4, 8, 12, 16, 20, 24, 36, 48)), ("C2", 100000, 1024, (0, 14, 28, 42, 56)),
                          ("C5 chunk", 131072, 4096, (0, 4, 8, 12, 16)), ("C3 shard", 125008, 2048, (0, 8, 12, 16, 24))]:
    for s in cands:
        best = min(h.dbg_syrk_variant(N, M, 0, s, 5)[0] for _ in range(2))
        print("%-9s N %7d M %5d splits %3d: %.3f ms -> %.1f TFLOP/s" % (name, N, M, s, best, N * M * (M + 1.0) / best * 1e-9), flush=True)
