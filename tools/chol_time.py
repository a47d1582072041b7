"""Time of the factorisation + inverse of an n x n matrix on the device (edrgp_dbg_potrf_inv includes the copies:
the kernel time is taken from a kernel trace, or from differences between option settings)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
h = HipGP(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
rng = np.random.default_rng(0)
Q = rng.standard_normal((n, n)); A = Q @ Q.T / n + np.eye(n)
for _ in range(3):
    L, Li, info = h.dbg_potrf_inv(A)
print("n=%d info=%d |Li L - I| = %.2e" % (n, info, np.abs(Li @ L - np.eye(n)).max()))
h.close()
