"""Fixed cost per 128 x 128 tile of the contraction-shaped product C (N x M) = A (N x K) B (K x M), A m-major as K_nm is:
the launch time for K = 1024 / 2048 / 4096 gives a + b K per tile.  Kernel times from the trace:
    rocprofv3 --kernel-trace --stats -- python3 tools/gemm_k_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
N, M = 100000, 2048
rng = np.random.default_rng(0)
h = HipGP(0)
for K in (1024, 2048, 4096):
    A = rng.random((N, K))
    B = rng.random((K, M))
    for rep in range(3):
        C = h.dbg_gemm(A, B)
    print(K, float(C[0, 0]))
h.close()
