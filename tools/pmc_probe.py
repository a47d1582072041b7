"""Workload for rocprofv3 --pmc passes: a few bound evaluations at the headline size."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from edrgp_amd.backend import HipGP  # noqa: E402

N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=1000)
h = HipGP(0)
if len(sys.argv) > 2:
    h.set_option("syrk_xmap", int(sys.argv[2]))
h.set_data(X, y)
h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print(h.sparse_lml())
h.close()
