"""C3 whole on one GPU (1M rows, M = 2048, D = 32, Matern52): eight row chunks (default) against one resident chunk."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP

h = HipGP(0)
N, M, D, kind = 1000000, 2048, 32, "Matern52"
X, y, Z = bench.synthetic(N, D, M, seed=7)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)

def timeit(f, n=3):
    for _ in range(2):
        f()
    h.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3

res = {}
for chunk in (131072, 0):
    h.set_option("chunk_rows", chunk)
    v = h.sparse_lml()
    g = h.sparse_lml_grad()
    t_l = timeit(h.sparse_lml)
    t_g = timeit(h.sparse_lml_grad)
    h.sparse_posterior()
    mu, jac = h.predict_train(True, True)
    t_p = timeit(lambda: h.predict_train(True, True), 2)
    t_pg = timeit(lambda: h.predict_train_gram(True), 2)
    res[chunk] = (v, g, mu, jac)
    print("chunk_rows=%d: bound %.2f ms, bound+grad %.2f ms, predict+predict_gradient %.1f ms (Gram on device %.1f ms); bound %.10e"
          % (chunk, t_l, t_g, t_p, t_pg, v), flush=True)
a, b = res[131072], res[0]
rel = lambda x, y_: float(np.abs(np.asarray(x) - np.asarray(y_)).max() / (np.abs(np.asarray(y_)).max() + 1e-300))
print("one chunk vs eight: bound %.1e, dvar %.1e, dlen %.1e, dnoise %.1e, dZ %.1e, mean %.1e, jac %.1e"
      % (abs(a[0] - b[0]) / abs(a[0]), rel(b[1][1], a[1][1]), rel(b[1][2], a[1][2]), abs(b[1][3] - a[1][3]) / abs(a[1][3]),
         rel(b[1][4], a[1][4]), rel(b[2], a[2]), rel(b[3], a[3])))
