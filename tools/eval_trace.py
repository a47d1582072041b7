"""A few sparse evaluations at a BASELINE configuration for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/eval_trace.py Cm grad 3
prints the wall time per evaluation; tools/trace_timeline.py turns the trace into a per-evaluation timeline."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP

CFG = {"Cm": (100000, 2048, 32), "C2": (100000, 1024, 16), "C4": (200000, 1024, 64), "C3s": (125000, 2048, 32)}
KIND = {"C3s": "Matern52"}
name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
what = sys.argv[2] if len(sys.argv) > 2 else "grad"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N, M, D = CFG[name]
X, y, Z = bench.synthetic(N, D, M, seed=bench.RANK0_SEED)
h = HipGP(0)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel([KIND.get(name, "RBF")], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
f = h.sparse_lml_grad if what == "grad" else (h.sparse_posterior if what == "post" else h.sparse_lml)
f(); f()
h.sync()
t0 = time.perf_counter()
for _ in range(reps):
    h.set_inducing(Z)
    f()
h.sync()
print("%s %s: %.3f ms per evaluation (mode %s)" % (name, what, (time.perf_counter() - t0) / reps * 1e3, h.get("MODE", (4,))))
# one more evaluation behind an idle gap: tools/trace_timeline.py takes the kernels after the last gap > 2 ms
time.sleep(0.01)
h.set_inducing(Z)
f()
h.sync()
h.close()
