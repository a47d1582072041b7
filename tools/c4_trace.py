"""BASELINE config 4 units (bound + posterior + N-point mean-gradient Gram matrix) for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c4 -- python3 tools/c4_trace.py 64 4
args: d_c, number of length-scales; prints ms per unit once the base rows are resident."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
from edrgp_amd.sweep import projection_sweep

dc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N, M, D = 200000, 1024, 64
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)); y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Zr = rng.permutation(N)[:M]
h = HipGP(0)
batch = True
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    if k == "batch":
        batch = bool(int(v))
    else:
        h.set_option(k, float(v))
W0 = np.linalg.qr(rng.standard_normal((D, dc)))[0]
W1 = np.linalg.qr(rng.standard_normal((D, dc)))[0]
projection_sweep(X, y, Zr, [W0], n_scales=ns, want_gram=True, hip=h, batch=batch)           # uploads the base rows, warms up
h.sync()
t0 = time.perf_counter()
out = projection_sweep(X, y, Zr, [W1], n_scales=ns, want_gram=True, hip=h, batch=batch)
dt = time.perf_counter() - t0
modes = h.get("MODE", (4,))
forms = [r.get("form") for r in out]
print("C4 d_c=%d batch=%d: %.2f ms per (bound + posterior + gradient Gram) unit over %d length-scales; forms %s"
      % (dc, batch, dt * 1e3 / ns, ns, forms if batch else modes))
h.close()
