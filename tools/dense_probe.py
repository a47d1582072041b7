import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
h = HipGP(0)
for N in [int(a) for a in sys.argv[1:]] or (2000, 8000, 16000, 32000):
    D = 8
    X, y, _ = bench.synthetic(N, D, 8, seed=3)
    h.set_data(X, y)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.1)
    h.dense_lml()
    t0 = time.perf_counter(); v = h.dense_lml(); t1 = time.perf_counter()
    g = h.dense_lml_grad(); t2 = time.perf_counter()
    h.dense_posterior()
    t3 = time.perf_counter(); mu, var = h.predict(X[:4096]); jac = h.predict_grad(X[:4096]); t4 = time.perf_counter()
    print("dense exact GP N=%d: LML %.1f ms (%.1f TFLOP/s on N^3/3 + N^3/3 + N^3/3), LML+grad %.1f ms, predict+grad on 4096 rows %.1f ms"
          % (N, (t1 - t0) * 1e3, N ** 3 / (t1 - t0) * 1e-12, (t2 - t1) * 1e3, (t4 - t3) * 1e3), flush=True)
