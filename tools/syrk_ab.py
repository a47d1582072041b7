"""Isolated Phi launch (edrgp_dbg_syrk_time) under option groups, same box, interleaved repeats."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
groups = sys.argv[1:] or ["syrk_persist=0", "syrk_persist=1"]
for N, M in ((100000, 2048), (131072, 2048), (100000, 1024)):
    for rep in range(2):
        for g in groups:
            h = HipGP(0)
            for kv in filter(None, g.split(",")):
                k, v = kv.split("=")
                h.set_option(k, float(v))
            ms = h.dbg_syrk_time(N, M, 0, 20)
            ms = ms[0] if isinstance(ms, tuple) else ms
            print("N=%d M=%d [%s]: %.3f ms  (%.1f TFLOP/s)" % (N, M, g, ms, N * M * (M + 1.0) / ms / 1e9), flush=True)
            h.close()
