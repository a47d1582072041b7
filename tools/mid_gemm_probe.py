"""Times of the evaluation's M x M x M product types (zero-skipping / lower tiles only) at M = 2048 or argv[1]:
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/mid_gemm_probe.py 2048 [tile]
then read the gemm_f64_kernel durations in launch order (3 launches per type)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LE_ROW, GE_ROW, LE_COL, GE_COL = 1, 2, 4, 8
TYPES = [("plain NN", 0, 0, 0, 0), ("plain TN", 1, 0, 0, 0), ("plain NT", 0, 1, 0, 0),
         ("Li Phi        NN k<=row", 0, 0, LE_ROW, 0), ("A = Tm Li^T   NT k<=col tri", 0, 1, LE_COL, 1),
         ("R = LBi Li    NN k<=row,k>=col tri", 0, 0, LE_ROW | GE_COL, 1), ("P = R^T R     TN k>=row,k>=col tri", 1, 0, GE_ROW | GE_COL, 1),
         ("C1 = Li^T A   TN k>=row", 1, 0, GE_ROW, 0), ("Tm = C1 Li    NN k>=col", 0, 0, GE_COL, 0)]
rng = np.random.default_rng(0)
A = np.tril(rng.random((M, M))); B = np.tril(rng.random((M, M)))
h = HipGP(0)
h.set_option("dbg_gemm_reps", 3)
if tile:
    h.set_option("dbg_gemm_tile", tile)
for name, ta, tb, km, tri in TYPES:
    h.set_option("dbg_gemm_kmode", km); h.set_option("dbg_gemm_tri", tri)
    h.dbg_gemm(A, B, transA=bool(ta), transB=bool(tb))
    print(name)
h.close()
