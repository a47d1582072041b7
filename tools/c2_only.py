import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 100000, 1024, 16
X, y, Z = bench.synthetic(N, D, M, seed=7)
h = HipGP(0)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for _ in range(5):
    h.sparse_lml()
for _ in range(3):
    h.sparse_lml_grad()
