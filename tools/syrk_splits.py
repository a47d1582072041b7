import sys, os
sys.path.insert(0, os.getcwd())
from edrgp_amd.backend import HipGP
h = HipGP(0)
for N, M in [(200000, 1024), (100000, 1024), (100000, 2048)]:
    for s in [0, 16, 32]:
        ms, _ = h.dbg_syrk_time(N, M, s, 5)
        print(N, M, "splits", s, "%.3f ms  %.1f TFLOP/s" % (ms, N * M * (M + 1.0) / ms * 1e-9), flush=True)
