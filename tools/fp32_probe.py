"""fp32 row pass vs the fp64 device path and the CPU oracle: errors at a small size, timing at a large one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP

h = HipGP(0)
rng = np.random.default_rng(0)
# building blocks
for (m, n, k, ta, tb) in [(200, 150, 333, False, False), (129, 257, 64, False, True), (300, 70, 1000, True, False), (64, 64, 40, True, True)]:
    A = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
    B = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
    C = h.dbg_gemm32(A, B, ta, tb)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    print("gemm32", m, n, k, ta, tb, "max err / sqrt(k)", np.abs(C - ref).max() / np.sqrt(k))
K = rng.random((5000, 384)).astype(np.float32)
P = h.dbg_syrk32(K)
ref = K.astype(np.float64).T @ K.astype(np.float64)
print("syrk32 rel", np.abs(P - ref).max() / np.abs(ref).max())

def run(N, M, D, kind, prec, grad=True):
    X, y, Z = bench.synthetic(N, D, M, seed=5)
    h.set_option("precision", prec)
    h.set_option("f32_guard", float(os.environ.get("F32_GUARD", "1.0")))
    h.set_option("f32_ratio_max", float(os.environ.get("F32_RATIO", "1e4")))
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel([kind], [1.3], np.full((1, D), 0.9 * np.sqrt(D)), 0.05)
    out = h.sparse_lml_grad() if grad else (h.sparse_lml(),)
    post = h.sparse_posterior()
    mean, g = h.predict_train(True, True)
    md = h.get("MODE", (2,))
    print("   precision", prec, "mode", int(md[0]), "rho %.2e" % md[1])
    return out, post, mean, g

for kind in ["RBF", "Matern52", "Matern32"]:
    for (N, M, D) in [(3000, 256, 8), (20000, 512, 32), (20000, 512, 64)]:
        o64, p64, m64, g64 = run(N, M, D, kind, 64)
        o32, p32, m32, g32 = run(N, M, D, kind, 32)
        rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))
        print(kind, N, M, D, "lml rel %.2e | dvar %.2e dlen %.2e dnoise %.2e dZ %.2e | mean %.2e grad %.2e" % (
            abs(o32[0] - o64[0]) / abs(o64[0]), rel(o32[1], o64[1]), rel(o32[2], o64[2]), abs(o32[3] - o64[3]) / abs(o64[3]),
            rel(o32[4], o64[4]), rel(m32, m64), rel(g32, g64)))

if len(sys.argv) > 1:
    N, M, D = [int(v) for v in sys.argv[1:4]]
    X, y, Z = bench.synthetic(N, D, M, seed=1)
    for prec in (32, 64):
        h.set_option("precision", prec)
        h.set_data(X, y); h.set_inducing(Z)
        h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
        h.sparse_lml()
        t0 = time.perf_counter(); v = h.sparse_lml(); t1 = time.perf_counter()
        g = h.sparse_lml_grad(); t2 = time.perf_counter()
        md = h.get("MODE", (2,))
        print("precision", prec, "mode", int(md[0]), "rho %.2e" % md[1], "N M D", N, M, D, "lml %.6f  lml ms %.1f  lml+grad ms %.1f" % (v, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
