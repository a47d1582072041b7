"""Developer timing probe (not the judged bench): LML / LML+grad / predict
timings and the per-family device time at a given (N, M, D)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP  # noqa: E402


def synthetic(N, D, M, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    Z = X[rng.permutation(N)[:M]].copy()
    return X, y, Z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=100000)
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--D", type=int, default=32)
    ap.add_argument("--kind", default="RBF")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--overlap", type=int, default=1)
    a = ap.parse_args()
    X, y, Z = synthetic(a.N, a.D, a.M)
    h = HipGP(0)
    tf, bw = h.calib()
    out = {"calib_mfma_f64_tflops": tf, "calib_hbm_copy_gbps": bw, "N": a.N, "M": a.M, "D": a.D}
    h.set_option("stats_mode", a.mode)
    h.set_option("syrk_splits", a.splits)
    h.set_option("overlap", a.overlap)
    h.set_data(X, y)
    h.set_inducing(Z)
    h.set_kernel([a.kind], [1.0], np.full((1, a.D), np.sqrt(a.D)), 0.01)
    h.sparse_lml()
    h.set_option("profile", 1)
    for name, fn in (("lml", h.sparse_lml), ("lml_grad", h.sparse_lml_grad), ("posterior", h.sparse_posterior),
                     ("predict_train", h.predict_train)):
        fn()
        h.profile_reset()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        h.sync()
        dt = (time.perf_counter() - t0) / a.reps
        prof = {k: h.profile_get(k) for k in ("SYRK", "KBUILD", "GRADGEMM", "SMALL")}
        out[name] = {"ms": dt * 1e3, "prof_ms_per_call": {k: v[0] / a.reps for k, v in prof.items()},
                     "launches": {k: v[1] / a.reps for k, v in prof.items()}}
    h.set_option("trace", 1)
    t0 = time.perf_counter(); h.sparse_lml_grad(); t1 = time.perf_counter()
    sys.stderr.write("python-side lml_grad wall %.3f ms\n" % ((t1 - t0) * 1e3))
    t0 = time.perf_counter(); h.predict_train(); t1 = time.perf_counter()
    sys.stderr.write("python-side predict_train wall %.3f ms\n" % ((t1 - t0) * 1e3))
    h.set_option("trace", 0)
    syrk_ms = out["lml"]["prof_ms_per_call"]["SYRK"]
    flops = a.N * a.M * (a.M + 1.0)
    out["syrk_tflops_algorithmic"] = flops / (syrk_ms * 1e-3) * 1e-12
    out["mode"] = h.get("MODE", (1,))[0]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
