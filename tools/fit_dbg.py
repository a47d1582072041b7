import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.gp_model import SparseGaussianProcessRegressor
from edrgp_amd import models as _m
N, M, D = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (100000, 2048, 32)
X, y, _ = bench.synthetic(N, D, M, seed=5)
_orig = _m.HipEvaluator.lml_grad
def _counted(self, model):
    t = time.perf_counter()
    try:
        r = _orig(self, model)
        ok = "lml %.3f" % r[0]
    except Exception as e:
        ok = "FAILED %s" % type(e).__name__
        raise
    finally:
        p = model.kern.parts[0]
        print("eval %.1f ms  var %.4g ls %.4g noise %.4g  |Z-Z0| %.3g  %s mode %s" % ((time.perf_counter() - t) * 1e3, p.variance,
              float(np.mean(p.lengthscale)), model.noise_variance, float(np.abs(model.Z - Z0).max()), ok, self.hip.get("MODE", (1,))[0]), flush=True)
    return r
_m.HipEvaluator.lml_grad = _counted
np.random.seed(0)
est = SparseGaussianProcessRegressor(num_inducing=M, kernel_options={"lengthscale": float(np.sqrt(D))})
Z0 = None
orig_get = est._get_model
def gm(X_, y_, k):
    global Z0
    m = orig_get(X_, y_, k)
    Z0 = m.Z.copy()
    return m
est._get_model = gm
est.fit(X, y, max_iters=20)
