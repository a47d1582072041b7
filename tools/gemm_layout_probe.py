"""Does the gradient contraction's operand layout cost it?  C (N x M) = A B with A = K_nm (N x M) once as stored
(m-major row tiles) and once from a transposed copy (k-major, the layout the Phi kernel reads).  Kernel times come
from the kernel trace:  rocprofv3 --kernel-trace --stats -- python3 tools/gemm_layout_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 2048)
rng = np.random.default_rng(0)
A = rng.random((N, M))
B = rng.random((M, M))
h = HipGP(0)
for rep in range(2):
    C1 = h.dbg_gemm(A, B)                                   # NN: A m-major
At = np.ascontiguousarray(A.T)
for rep in range(2):
    C2 = h.dbg_gemm(At, B, transA=True)                     # TN: A k-major
print("max |C1 - C2| / max|C1| = %.2e" % (np.abs(C1 - C2).max() / np.abs(C1).max()))
h.close()
