"""BASELINE config 5 (N=2M, M=4096, D=64, RBF): fp32 row pass vs the fp64 path on the same inputs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP

N, M, D = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (2000000, 4096, 64)
t0 = time.perf_counter()
X, y, Z = bench.synthetic(N, D, M, seed=1)
print("synthetic %.1f s" % (time.perf_counter() - t0), flush=True)
h = HipGP(0)
res = {}
for prec in (32, 64):
    h.set_option("precision", prec)
    h.set_option("profile", 1)
    if "CHUNK_ROWS" in os.environ:
        h.set_option("chunk_rows", float(os.environ["CHUNK_ROWS"]))
    if "F32_GUARD" in os.environ:
        h.set_option("f32_guard", float(os.environ["F32_GUARD"]))
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    h.sparse_lml()
    h.profile_reset()
    t0 = time.perf_counter(); v = h.sparse_lml(); t1 = time.perf_counter()
    prof_l = {k: h.profile_get(k)[0] for k in ("SYRK", "KBUILD", "GRADGEMM", "SMALL")}
    h.profile_reset()
    g = h.sparse_lml_grad(); t2 = time.perf_counter()
    prof_g = {k: h.profile_get(k)[0] for k in ("SYRK", "KBUILD", "GRADGEMM", "SMALL")}
    md = h.get("MODE", (2,))
    print("precision %d mode %d rho %.2e  N=%d M=%d D=%d  lml %.6f | lml %.1f ms, lml+grad %.1f ms" % (
        prec, int(md[0]), md[1], N, M, D, v, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    print("   device ms (bound only):", {k: round(x, 2) for k, x in prof_l.items()})
    print("   device ms (bound+grad):", {k: round(x, 2) for k, x in prof_g.items()}, flush=True)
    res[prec] = g
a, b = res[32], res[64]
rel = lambda x, y: float(np.abs(np.asarray(x) - np.asarray(y)).max() / (np.abs(np.asarray(y)).max() + 1e-300))
print("fp32 vs fp64: lml rel %.2e | dvar %.2e dlen %.2e dnoise %.2e dZ %.2e" % (
    abs(a[0] - b[0]) / abs(b[0]), rel(a[1], b[1]), rel(a[2], b[2]), abs(a[3] - b[3]) / abs(b[3]), rel(a[4], b[4])))
