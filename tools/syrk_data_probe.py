import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M = 100000, 2048
flops = N * M * (M + 1.0)
for rnd in range(3):
    for mode in (0, 1, 2, 3, 4):
        h.set_option("gemm_dephase", mode)
        ms, _ = h.dbg_syrk_variant(N, M, 0, 0, 3)
        print("dephase %d: %.3f ms -> %.1f TFLOP/s algorithmic" % (mode, ms, flops / ms * 1e-9))
print("calib", h.calib())
