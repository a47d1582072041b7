import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
from edrgp_amd.sweep import projection_sweep
h = HipGP(0)
N, M, D = 200000, 1024, 64
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)); y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Zr = rng.permutation(N)[:M]
W = np.linalg.qr(rng.standard_normal((D, 32)))[0]
projection_sweep(X, y, Zr, [W], n_scales=2, want_gram=True, hip=h)
projection_sweep(X, y, Zr, [W], n_scales=4, want_gram=True, hip=h)
