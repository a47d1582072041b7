"""Critical path of the fused factorisation + inverse: phases of workgroup 0 of every step launch (probe build:
EDRGP_BUILD_VARIANT=stamps python edr-gp_amd/build.py):
    EDRGP_LIB=edr-gp_amd/libedrgp_hip_stamps.so EDRGP_CHOL_STAMPS_FILE=/tmp/cs.bin python tools/chol_stamps.py [n=2048]
Slots 0..6 on the 100 MHz wall clock: entry, operands staged, L_i product done, L_i through LDS, update done,
diagonal block staged, diagonal-block body done; slots 8..15 in shader cycles inside the body (barriers of the four
panels, inverse columns done)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
h = HipGP(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
rng = np.random.default_rng(0)
Q = rng.standard_normal((n, n)); A = Q @ Q.T / n + np.eye(n)
for _ in range(3):
    L, Li, info = h.dbg_potrf_inv(A)
st = np.fromfile(os.environ["EDRGP_CHOL_STAMPS_FILE"], dtype=np.uint64).reshape(64, 16).astype(np.int64)
nstep = (n + 63) // 64 - 1
w = st[:nstep, :7] / 100.0            # us
names = ["entry->staged", "L_i product", "L_i via LDS", "update", "stage diag", "diag body"]
print("step: gap from previous exit | " + " | ".join(names) + " | entry->exit   (us)")
rows = []
for k in range(nstep):
    gap = w[k, 0] - w[k - 1, 6] if k > 0 else float("nan")
    d = [w[k, i + 1] - w[k, i] for i in range(6)]
    rows.append([gap] + d + [w[k, 6] - w[k, 0]])
    if k < 3 or k % 8 == 0 or k >= nstep - 2:
        print("%3d: %6.2f | " % (k, gap) + " | ".join("%6.2f" % v for v in d) + " | %6.2f" % (w[k, 6] - w[k, 0]))
r = np.array(rows[1:])
print("mean over steps 1..: gap %.2f | " % np.nanmean(r[:, 0]) + " | ".join("%.2f" % v for v in r[:, 1:7].mean(0)) + " | %.2f; step period %.2f us"
      % (r[:, 7].mean(), (w[nstep - 1, 6] - w[0, 6]) / (nstep - 1)))
cyc = st[:nstep, 8:16] - st[:nstep, 8:9]
b = st[:nstep, 8:12]
print("body, shader cycles: panel barriers w1-w0 %.0f, w2-w1 %.0f, w3-w2 %.0f; inverse tail after barrier 3: %.0f"
      % ((b[:, 1] - b[:, 0]).mean(), (b[:, 2] - b[:, 1]).mean(), (b[:, 3] - b[:, 2]).mean(), (st[:nstep, 12:16].max(1) - b[:, 3]).mean()))
h.close()
