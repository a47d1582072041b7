"""Entry / exit of every workgroup of the Phi launch (probe build, EDRGP_TILE_STAMPS_PHI=1):
    EDRGP_LIB=edr-gp_amd/libedrgp_hip_stamps.so EDRGP_TILE_STAMPS_PHI=1 EDRGP_TILE_STAMPS_FILE=/tmp/phi.bin python tools/phi_stamps.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=bench.RANK0_SEED)
h = HipGP(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for _ in range(3):
    h.sparse_lml()
h.sync()
st = np.fromfile(os.environ["EDRGP_TILE_STAMPS_FILE"], dtype=np.uint64).reshape(-1, 6).astype(np.int64)
st = st[st[:, 3] > 0]
t0 = st[:, 0].min()
ent, ex, pair = (st[:, 0] - t0) / 100.0, (st[:, 3] - t0) / 100.0, st[:, 1] > 0
dur = ex - ent
print("workgroups %d (pair %d), span %.1f us" % (len(st), pair.sum(), ex.max()))
print("regular: mean %.1f p5 %.1f p95 %.1f us; pair: mean %.1f p5 %.1f p95 %.1f us" % (
    dur[~pair].mean(), np.percentile(dur[~pair], 5), np.percentile(dur[~pair], 95),
    dur[pair].mean(), np.percentile(dur[pair], 5), np.percentile(dur[pair], 95)))
order = np.argsort(ent)
if len(ent) > 1024:
    print("entries: round starts (us):", np.round(np.sort(ent)[[0, 511, 512, 1023, 1024, len(ent) - 1]], 1))
ends = np.sort(ex)
for q in (50, 75, 90, 95, 99, 100):
    print("  %3d %% of the workgroups have left by %.1f us" % (q, np.percentile(ex, q)))
last = ex > ex.max() - 300
print("running in the last 300 us: %d workgroups, %d of them pair" % ((ex > ex.max() - 300).sum(), (pair & last).sum()))
# resident workgroups over time
ev = sorted([(t, 1) for t in ent] + [(t, -1) for t in ex])
occ, last_t, area = 0, 0.0, 0.0
for t, d in ev:
    area += occ * (t - last_t); last_t = t; occ += d
print("mean resident workgroups over the span: %.1f of 512" % (area / ex.max()))
