"""T^T X alone: C (M x W) = A^T B, A = N x M, B = N x W (the tail of the gradient row pass) for W = 64 and 32 columns;
checks against numpy, kernel times from the trace (tools/ttx_run.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
N, M = 100000, 2048
rng = np.random.default_rng(0)
A = rng.random((N, M)) - 0.5
h = HipGP(0)
h.set_option("dbg_gemm_reps", 3)
for W in (64, 32, 17):
    B = rng.random((N, W)) - 0.5
    C = h.dbg_gemm(A, B, transA=True)
    ref = A.T @ B
    print("W=%d max rel err %.2e" % (W, np.abs(C - ref).max() / np.abs(ref).max()), flush=True)
h.close()
