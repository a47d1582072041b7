"""Where a bench step (50 bound+gradient evaluations on the theta path + posterior + predict) spends its wall time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP

N, M, D, kind, iters = 100000, 2048, 32, "RBF", 50
h = HipGP(0)
X, y, Z = bench.synthetic(N, D, M, seed=0)
h.set_data(X, y); h.set_inducing(Z)
for _ in range(2):
    bench.fit_predict_step(h, Z, D, kind, iters)
h.sync()
t = {}
def tic(): h.sync(); return time.perf_counter()
modes = []
t0 = tic()
for i in range(iters):
    th = bench.theta_path(i, D, kind)
    a = time.perf_counter(); h.set_inducing(Z); b = time.perf_counter()
    h.set_kernel([kind], [th["variance"]], np.full((1, D), th["lengthscale"]), th["noise"]); c = time.perf_counter()
    h.sparse_lml_grad(); d = time.perf_counter()
    t["set_inducing"] = t.get("set_inducing", 0) + b - a
    t["set_kernel"] = t.get("set_kernel", 0) + c - b
    t["lml_grad"] = t.get("lml_grad", 0) + d - c
    modes.append(int(h.get("MODE", (1,))[0]))
t1 = tic()
h.sparse_posterior(); t2 = tic()
h.predict_train(want_mean=True, want_grad=True); t3 = tic()
print("50 evaluations %.1f ms (set_inducing %.1f, set_kernel %.1f, lml_grad calls %.1f), posterior %.2f ms, predict_train %.2f ms"
      % ((t1 - t0) * 1e3, t["set_inducing"] * 1e3, t["set_kernel"] * 1e3, t["lml_grad"] * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print("forms:", modes)
# the same 50 evaluations at the fixed reference hyper-parameters
h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
t0 = tic()
for i in range(iters):
    h.sparse_lml_grad()
t1 = tic()
print("50 evaluations at fixed theta: %.1f ms" % ((t1 - t0) * 1e3))
for prof in (1, 0, 1, 0):
    h.set_option("profile", prof)
    h.profile_reset()
    t0 = tic()
    for i in range(iters):
        h.sparse_lml_grad()
    t1 = tic()
    print("profile=%d: 50 evaluations %.1f ms" % (prof, (t1 - t0) * 1e3))
h.set_option("profile", 0)
