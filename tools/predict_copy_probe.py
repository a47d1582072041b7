"""predict + predict_gradient on 1M resident training rows: fresh output arrays (what the API returns) against reused ones."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
from edrgp_amd._lib import check, dptr
h = HipGP(0)
N, M, D = 1000000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=7)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["Matern52"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
h.sparse_posterior()
h.predict_train(True, True)
mean, grad = np.empty(N), np.empty((N, D))
for rep in range(3):
    h.sync(); t0 = time.perf_counter()
    check(h._lib.edrgp_predict_train(h._ctx, dptr(mean), dptr(grad))); t1 = time.perf_counter()
    m2, g2 = h.predict_train(True, True); t2 = time.perf_counter()
    _, gram, _ = h.predict_train_gram(True); t3 = time.perf_counter()
    check(h._lib.edrgp_predict_train(h._ctx, None, dptr(grad))); t4 = time.perf_counter()
    print("reused arrays %.1f ms | fresh arrays %.1f ms | Gram on device (+ mean copied) %.1f ms | Jacobian only, reused %.1f ms"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
    del m2, g2
