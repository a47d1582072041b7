"""Where a contraction tile's time goes (probe build: EDRGP_BUILD_VARIANT=stamps python edr-gp_amd/build.py):
    EDRGP_LIB=edr-gp_amd/libedrgp_hip_stamps.so EDRGP_TILE_STAMPS_FILE=/tmp/st.bin python tools/tile_stamps.py
Per workgroup: entry, loop start, loop end, exit on the 100 MHz wall clock, HW_ID, XCC_ID."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = 100000, 2048, 32
X, y, Z = bench.synthetic(N, D, M, seed=bench.RANK0_SEED)
h = HipGP(0)
h.set_data(X, y); h.set_inducing(Z)
KIND = os.environ.get("KIND", "RBF")
h.set_kernel([KIND], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    h.set_option(k, float(v))
import time
for _ in range(3):
    h.sparse_lml_grad()
h.sync()
t0 = time.perf_counter()
for _ in range(3):
    h.set_inducing(Z)
    h.sparse_lml_grad()
h.sync()
print("bound+grad %.3f ms per evaluation (with the probe's device sync per contraction)" % ((time.perf_counter() - t0) / 3 * 1e3))
st = np.fromfile(os.environ["EDRGP_TILE_STAMPS_FILE"], dtype=np.uint64).reshape(-1, 6).astype(np.int64)
t0 = st[:, 0].min()
ent, lp, le, ex = [(st[:, i] - t0) / 100.0 for i in range(4)]          # us
hw, xcc = st[:, 4], st[:, 5] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
print("tiles %d, launch span %.1f us" % (len(st), ex.max()))
print("prologue (entry -> loop)  mean %.2f  p50 %.2f  p95 %.2f us" % ((lp - ent).mean(), np.median(lp - ent), np.percentile(lp - ent, 95)))
print("k-loop                    mean %.2f  p50 %.2f  p5 %.2f  p95 %.2f us" % ((le - lp).mean(), np.median(le - lp), np.percentile(le - lp, 5), np.percentile(le - lp, 95)))
print("epilogue (loop end->exit) mean %.2f  p50 %.2f  p95 %.2f us" % ((ex - le).mean(), np.median(ex - le), np.percentile(ex - le, 95)))
# per (xcc, cu) slot occupancy: gaps between a workgroup's exit and the next entry on the same CU
key = xcc * 1024 + cu
gaps, conc = [], []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    ev = sorted([(ent[i], 1) for i in idx] + [(ex[i], -1) for i in idx])
    occ, last, t_by = 0, 0.0, {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    for t, d in ev:
        t_by[min(occ, 3)] += t - last
        last = t
        occ += d
    conc.append([t_by[0], t_by[1], t_by[2], t_by[3]])
    for a, b in zip(sorted(ex[idx])[:-2], sorted(ent[idx])[2:]):
        gaps.append(b - a)
conc = np.array(conc)
gaps = np.array(gaps)
print("turnover on a CU (k-th exit -> (k+2)-th entry): mean %.1f  p50 %.1f  p95 %.1f us" % (gaps.mean(), np.median(gaps), np.percentile(gaps, 95)))
print("CUs seen: %d; per CU time with 0 / 1 / 2 / 3+ resident workgroups (mean us): %s" % (len(conc), np.round(conc.mean(0), 1)))
first_round = np.sort(ent)[:512]
print("entry of the first 512 workgroups: %.1f .. %.1f us; last exit %.1f; tiles per CU min/max %d/%d" % (
    first_round.min(), first_round.max(), ex.max(), np.bincount(np.unique(key, return_inverse=True)[1]).min(),
    np.bincount(np.unique(key, return_inverse=True)[1]).max()))
# loop time of workgroups that started in the last 600 us (the tail) against the bulk
tail = ent > ex.max() - 600
print("k-loop in the tail (%d tiles) mean %.1f us vs bulk %.1f us" % (tail.sum(), (le - lp)[tail].mean(), (le - lp)[~tail].mean()))
