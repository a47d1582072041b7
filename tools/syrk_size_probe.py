import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
M = 2048
for N in (4096, 8192, 12288, 16384, 32768, 65536, 100000, 200000, 400000):
    best = 1e9
    for rnd in range(2):
        ms, _ = h.dbg_syrk_variant(N, M, 0, 0, 5)
        best = min(best, ms)
    print("N %7d (K_nm %6.0f MB): %.3f ms -> %.1f TFLOP/s algorithmic" % (N, N * M * 8 / 1e6, best, N * M * (M + 1.0) / best * 1e-9))
