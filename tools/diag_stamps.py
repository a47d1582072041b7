import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
from edrgp_amd._lib import dptr, check
h = HipGP(0)
rng = np.random.default_rng(0)
Q = rng.standard_normal((64, 64)); A = Q @ Q.T + 64 * np.eye(64)
for do_chol in (1, 0):
    src = A if do_chol else np.linalg.cholesky(A)
    for rep in range(3):
        st = np.zeros(6)
        check(h._lib.edrgp_dbg_diag_stamps(h._ctx, dptr(np.ascontiguousarray(src)), do_chol, dptr(st)))
        print("do_chol", do_chol, "stamps(cycles): load %d factor %d writeback %d inverse %d store %d" % (st[1], st[2]-st[1], st[3]-st[2], st[4]-st[3], st[5]-st[4]))
