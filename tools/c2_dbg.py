import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
N, M, D = 100000, 1024, 16
X, y, Z = bench.synthetic(N, D, M, seed=7)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for ov in (3, 0):
    h.set_option("overlap", ov)
    for i in range(4):
        t0 = time.perf_counter(); v = h.sparse_lml(); t1 = time.perf_counter()
        print("overlap", ov, "lml %.3f ms mode %d" % ((t1 - t0) * 1e3, h.get("MODE", (1,))[0]), flush=True)
    for i in range(3):
        t0 = time.perf_counter(); v = h.sparse_lml_grad(); t1 = time.perf_counter()
        print("overlap", ov, "grad %.3f ms mode %d" % ((t1 - t0) * 1e3, h.get("MODE", (1,))[0]), flush=True)
# gram check
h.sparse_posterior()
mean, J = h.predict_train(True, True)
m2, G, gs = h.predict_train_gram(True)
print("gram rel err", np.abs(G - J.T @ J).max() / np.abs(J.T @ J).max(), "gsum", np.abs(gs - J.sum(0)).max() / np.abs(J.sum(0)).max(), "mean", np.abs(mean - m2).max())
