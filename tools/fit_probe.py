"""End-to-end estimator timing at Cm through the sklearn-style API (host optimiser included)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.gp_model import SparseGaussianProcessRegressor
N, M, D = 100000, 2048, 32
X, y, _ = bench.synthetic(N, D, M, seed=5)
np.random.seed(0)
from edrgp_amd import models as _m
_count = {"n": 0, "t": 0.0}
_orig = _m.HipEvaluator.lml_grad
def _counted(self, model):
    t = time.perf_counter(); r = _orig(self, model); _count["t"] += time.perf_counter() - t; _count["n"] += 1
    return r
_m.HipEvaluator.lml_grad = _counted
est = SparseGaussianProcessRegressor(num_inducing=M, kernel_options={"lengthscale": float(np.sqrt(D))})
t0 = time.perf_counter()
est.fit(X, y, max_iters=50)
t1 = time.perf_counter()
g = est.predict_gradient(X)
mu = est.predict(X)
t2 = time.perf_counter()
m = est.estimator_
print("fit(max_iters=50): %.2f s; predict + predict_gradient on N rows: %.3f s; objective evaluations: %s; bound %.3f; train R^2 %.4f"
      % (t1 - t0, t2 - t1, "%d (%.2f s inside the native calls)" % (_count["n"], _count["t"]), float(np.ravel(m.log_likelihood())[0]), 1 - np.mean((mu - y) ** 2) / np.var(y)))
print("points/s through the estimator API: %.0f" % (N / (t2 - t0)))
