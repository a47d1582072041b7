#!/usr/bin/env python3
"""Headline benchmark: sparse-GP fit+predict throughput on MI355X.

Metric (BASELINE.json): "GP fit+predict throughput (points/sec) and log-ml
evals/sec at N=100k, M=2048; 1/2/4/8 GPU".  Workload = SURVEY.md section 8(d) "Cm":
N = 100,000 rows per GPU, M = 2048 inducing points, D = 32, RBF, fp64,
sigma_f^2 = 1, l = sqrt(D), sigma_n^2 = 0.01, synthetic tanh data.

One *step* = one fit+predict pass of the hot path over the resident batch:
  50 bound+gradient evaluations (the L-BFGS-B hot loop of `fit`,
  edrgp/gp_model/base.py:69, at a fixed iteration count and along a
  deterministic hyper-parameter path; every evaluation re-sends theta and Z and
  fetches the gradients, as the optimiser does)
  + posterior build + predict and predict_gradient on the N training points
  (edrgp/base.py:274).
value = points processed by all ranks per second = n_gpus * N * steps / time.
Inputs are resident in HBM before the timed region starts.  With N > 1 ranks the
rows are sharded and Phi / K_mn y / gradient partials are sum-all-reduced over
RCCL once per evaluation: --scaling weak (default) keeps N rows on every GPU,
--scaling strong splits ONE data set of N rows over the GPUs (value = N * steps /
time then; the metric's literal "N = 100k at 1/2/4/8 GPUs").

The same JSON line carries the log-ml evals/sec figure, the roofline of the
dominant kernel (the fp64 MFMA syrk Phi = K_mn K_nm, timed with HIP events on the
library's stream inside the timed region) and a CPU baseline (the numpy/scipy
oracle, i.e. GPy's operation sequence: one full-size bound evaluation, the rest from row samples).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_SPEC_TFLOPS = 78.6      # AMD public MI355X fp64 matrix peak; the local guide lists none
ITERS_PER_FIT = 50


RANK0_SEED = 1000


def _rows(N, D, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
    return rng, X, (y - y.mean()) / y.std()


def synthetic(N, D, M, seed, z_seed=RANK0_SEED):
    """X ~ N(0, I); y = tanh(X w) + 0.05 eps standardised; Z = X[perm[:M]] (GPy
    SparseGPRegression's rule) taken from RANK 0's rows, so every rank of a sharded run holds the same Z:
    a rank whose seed is not z_seed regenerates rank 0's rows just for that."""
    rng, X, y = _rows(N, D, seed)
    if seed == z_seed:
        Z = X[rng.permutation(N)[:M]].copy()
    else:
        rz, Xz, _ = _rows(N, D, z_seed)
        Z = Xz[rz.permutation(N)[:M]].copy()
    return X, y, Z


def theta_path(i, D, kind):
    """Deterministic hyper-parameter / inducing-input path standing in for optimiser iterates."""
    s = 1.0 + 1e-3 * ((i % 7) - 3)
    return dict(variance=1.0 * s, lengthscale=np.sqrt(D) * s, noise=0.01 * s)


def fit_predict_step(h, Z, D, kind, iters):
    for i in range(iters):
        th = theta_path(i, D, kind)
        h.set_inducing(Z)
        h.set_kernel([kind], [th["variance"]], np.full((1, D), th["lengthscale"]), th["noise"])
        h.sparse_lml_grad()
    h.sparse_posterior()
    h.predict_train(want_mean=True, want_grad=True)


def sharding_check(h, dist, rank, world, local_rank):
    """Row-sharded evaluation (RCCL all-reduce of [Phi | b] and of the gradient partials) against the same
    problem evaluated whole on rank 0 without a communicator: the relative differences go into the JSON line."""
    from edrgp_amd.backend import HipGP
    n, m, d = 4096, 256, 8
    Xv, yv, Zv = synthetic(n, d, m, seed=77 + rank)
    ls = np.full((1, d), np.sqrt(d))
    h.set_data(Xv, yv)
    h.set_inducing(Zv)
    h.set_kernel(["RBF"], [1.0], ls, 0.05)
    got = h.sparse_lml_grad()
    parts = [None] * world
    if dist is not None:
        dist.all_gather_object(parts, (Xv, yv, float(got[0])))
    else:
        parts = [(Xv, yv, float(got[0]))]
    res = None
    if rank == 0:
        try:                                   # (local to rank 0: no collective inside, a failure is only reported)
            h2 = HipGP(local_rank)
            try:
                h2.set_data(np.vstack([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
                h2.set_inducing(Zv)
                h2.set_kernel(["RBF"], [1.0], ls, 0.05)
                ref = h2.sparse_lml_grad()
            finally:
                h2.close()
            rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))
            res = {"rows": n * world, "bound_rel_err": abs(got[0] - ref[0]) / abs(ref[0]),
                   "grad_rel_err": max(rel(got[1], ref[1]), rel(got[2], ref[2]), abs(got[3] - ref[3]) / abs(ref[3]),
                                       rel(got[4], ref[4])),
                   "ranks_agree": len({p[2] for p in parts}) == 1}
        except Exception as e:                 # noqa: BLE001
            res = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def cpu_baseline(N, M, D, kind, iters, sample_rows=(4000, 12000), full_bound=True):
    """The CPU path timed beside the GPU one, on the host cores of the GPU box: the oracle (GPy's LAPACK/BLAS
    call sequence restated in numpy/scipy, `kind` = "port").  Bounded to about a minute of CPU work:
      * ONE bound evaluation at the full size (N of N rows), all cores  -> log-ml evals/s at the stated config;
      * bound+gradient evaluation and the predict + predict_gradient pass at two row samples, all cores; the time
        is linear in the rows plus a rows-independent O(M^3) part, so the two samples give both and the full-size
        time is a + b N (the round-1 figure divided a 4 000-row time by 4 000 rows, which charged the O(M^3) part
        25 times too often);
      * the bound at the same two samples on ONE thread -> per-core figure, same model.
    value = N / (iters x t_eval(N) + t_bound(N) + t_predict(N)), the step bench.py times on the GPU."""
    from oracle import gp_oracle as go
    from edrgp_amd.utils import effective_cpus
    from threadpoolctl import threadpool_limits
    X, y, Z = synthetic(N, D, M, seed=RANK0_SEED)
    kern = go.make_kernel([kind], D, [dict(variance=1.0, lengthscale=np.sqrt(D))])
    Y = y[:, None]
    # BLAS threads = the cores the process may really use (affinity capped by the cgroup quota): an
    # oversubscribed pool (one thread per visible core of the host) only slows the CPU path down
    threads = effective_cpus()
    n1, n2 = sample_rows

    def timed(f):
        t0 = time.perf_counter()
        r = f()
        return time.perf_counter() - t0, r

    def line(t1, t2, m1=None, m2=None):  # t(n) = a + b n through the two samples -> (a, b, t(N))
        m1, m2 = (n1, n2) if m1 is None else (m1, m2)
        bb = max((t2 - t1) / (m2 - m1), 0.0)
        aa = max(t1 - bb * m1, 0.0)
        return aa, bb, aa + bb * N

    def sample_times(n):
        Xs, Ys = X[:n], Y[:n]
        t_b, inf = timed(lambda: go.vardtc_inference(kern, Xs, Z, 0.01, Ys))
        t_g, _ = timed(lambda: go.sparse_gradients(kern, Xs, Z, inf))
        t_p, _ = timed(lambda: (go.raw_predict(kern, Xs, Z, inf["woodbury_vector"], inf["woodbury_inv"]),
                                go.predictive_gradients_mean(kern, Xs, Z, inf["woodbury_vector"])))
        return t_b, t_b + t_g, t_p

    with threadpool_limits(limits=threads):
        sample_times(min(n1, 1000))      # untimed: first-call costs (imports, allocator warm-up) stay out of the samples
        s1, s2 = sample_times(n1), sample_times(n2)
        t_bound_full, lml_full = None, None
        if full_bound:
            t_bound_full, inf = timed(lambda: go.vardtc_inference(kern, X, Z, 0.01, Y))
            lml_full = float(inf["log_marginal"][0, 0])
            del inf
    h1, h2 = n1 // 2, n2 // 2            # the single-thread leg runs on half-size samples
    with threadpool_limits(limits=1):
        t1_1, _ = timed(lambda: go.vardtc_inference(kern, X[:h1], Z, 0.01, Y[:h1]))
        t1_2, _ = timed(lambda: go.vardtc_inference(kern, X[:h2], Z, 0.01, Y[:h2]))
    _, _, t_bound_model = line(s1[0], s2[0])
    a_e, b_e, t_eval = line(s1[1], s2[1])
    _, _, t_pred = line(s1[2], s2[2])
    _, _, t_bound_1thread = line(t1_1, t1_2, h1, h2)
    t_bound = t_bound_full if t_bound_full is not None else t_bound_model
    step_time = iters * t_eval + t_bound + t_pred
    return {
        "value": N / step_time, "unit": "points/s", "cores": int(threads), "kind": "port",
        "sample": "bound: %d of %d rows measured once at full size (%.1f s, %d threads; the two-sample model below predicts "
                  "%.1f s); bound+gradient evaluation and predict+predict_gradient measured at %d and %d rows (%.2f / %.2f s and "
                  "%.2f / %.2f s) and extended as a + b N to N = %d (rows-independent part %.2f s, %.3f ms per row): %.1f s and "
                  "%.1f s; step = %d x eval + bound + predict = %.0f s; M=%d D=%d %s"
                  % (N if full_bound else 0, N, t_bound, threads, t_bound_model, n1, n2, s1[1], s2[1], s1[2], s2[2], N, a_e,
                     b_e * 1e3, t_eval, t_pred, iters, step_time, M, D, kind),
        "lml_evals_per_sec": 1.0 / t_bound,
        "lml_evals_per_sec_1thread": 1.0 / t_bound_1thread,
        "bound_full_size_s": t_bound_full, "bound_full_size_value": lml_full,
        "bound_1thread_s_model": t_bound_1thread,
        "eval_s_model": t_eval, "predict_s_model": t_pred,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=100000, help="rows per GPU")
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--D", type=int, default=32)
    ap.add_argument("--kind", default="RBF")
    ap.add_argument("--iters", type=int, default=ITERS_PER_FIT)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --N rows on EVERY GPU; strong: --N rows in total, row-sharded over the GPUs "
                         "(BASELINE's metric read literally: N = 100k at 1/2/4/8 GPUs; the replicated O(M^3) phases and "
                         "the exchange are the Amdahl term there, DESIGN section 7)")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 = skip the CPU baseline leg (oracle on the host cores)")
    ap.add_argument("--check-sharding", action="store_true", help="run the sharded-vs-whole check (always on for --gpus > 1)")
    ap.add_argument("--host-allreduce", action="store_true",
                    help="rehearsal of the N > 1 path on ONE device: every rank on GPU 0, torch.distributed on gloo, the "
                         "library's exchange through edrgp_comm_init_host (never what the driver measures)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    dist = None
    rehearsal = args.host_allreduce and world > 1
    if rehearsal:
        import torch.distributed as dist
        local_rank = 0
        dist.init_process_group(backend="gloo")
    elif world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from edrgp_amd.backend import HipGP
    h = HipGP(local_rank)                      # raises without a gfx950 device: no CPU fallback
    # The library can serve a posterior request from the factors of the bound+gradient evaluation that has just
    # run for the same state (as GPy's inference does); the benchmark step does NOT use that shortcut: its
    # "posterior build" is a full evaluation, nothing in the timed region is taken from a cache.
    h.set_option("posterior_reuse", 0)
    if rehearsal:
        h.comm_init_host(world, rank, lambda a: dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM))
    elif world > 1:
        box = [HipGP.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        h.comm_init(world, rank, box[0])

    shard_check = None
    if world > 1 or args.check_sharding:
        shard_check = sharding_check(h, dist, rank, world, local_rank)

    N, M, D = args.N, args.M, args.D
    strong = (args.scaling == "strong") and world > 1
    if strong:
        # one data set of N rows for the whole job (every rank generates it and keeps its contiguous row block)
        Xf, yf, Z = synthetic(N, D, M, seed=RANK0_SEED)
        cuts = [(N * r) // world for r in range(world + 1)]
        X, y = Xf[cuts[rank]:cuts[rank + 1]].copy(), yf[cuts[rank]:cuts[rank + 1]].copy()
        del Xf, yf
    else:
        X, y, Z = synthetic(N, D, M, seed=RANK0_SEED + rank)
    N_local = X.shape[0]
    h.set_data(X, y)                           # resident in HBM from here on
    h.set_inducing(Z)
    h.set_kernel([args.kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)

    def barrier_sync():
        h.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        fit_predict_step(h, Z, D, args.kind, args.iters)
    h.set_option("profile", 1)
    h.profile_reset()
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fit_predict_step(h, Z, D, args.kind, args.iters)
    barrier_sync()
    elapsed = time.perf_counter() - t0
    syrk_ms, syrk_n = h.profile_get("SYRK")
    grad_ms, grad_n = h.profile_get("GRADGEMM")
    kb_ms, kb_n = h.profile_get("KBUILD")
    small_ms, _ = h.profile_get("SMALL")
    chol_ms, chol_n = h.profile_get("CHOL")
    h.set_option("profile", 0)
    red_dev = "cpu" if rehearsal else "cuda"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # log-ml evals/sec (second half of the metric): bound-only evaluations, same resident data
    n_lml = 20
    h.set_kernel([args.kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    lml_ref_dev = h.sparse_lml()              # (world == 1: same inputs as the CPU leg's full-size bound)
    if world > 1:
        lml_ref_dev = None
    barrier_sync()
    t1 = time.perf_counter()
    for _ in range(n_lml):
        h.sparse_lml()
    barrier_sync()
    lml_elapsed = time.perf_counter() - t1
    if dist is not None:
        t = torch.tensor([lml_elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lml_elapsed = float(t.item())
    mode = int(h.get("MODE", (1,))[0])
    # the dominant kernel alone on the GPU (no side-stream work beside it): same shape, device-made operand
    iso_ms = None
    if rank == 0 and N_local % 16 == 0 and M % 128 == 0:
        try:
            h.sync()
            iso_ms, _ = h.dbg_syrk_time(N_local, M, 0, 5)
        except Exception:
            iso_ms = None
    chol_iso_ms = None
    if rank == 0:
        try:
            Bm = np.random.default_rng(5).standard_normal((M, M))
            chol_iso_ms = h.dbg_potrf_time(Bm @ Bm.T / M + np.eye(M), 5)
        except Exception:
            chol_iso_ms = None

    if rank == 0:
        value = (N if strong else world * N) * args.steps / elapsed
        flops_per_launch = float(N_local) * M * (M + 1.0)           # syrk, lower triangle only
        avg_ms = syrk_ms / max(syrk_n, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) * 1e-12 if syrk_n else None
        traffic = None
        pmc_name = "r03_pmc_syrk.json" if os.path.exists(os.path.join(ROOT, "profiles", "r03_pmc_syrk.json")) else "pmc_syrk.json"
        pmc_path = os.path.join(ROOT, "profiles", pmc_name)
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        calib_tf = None
        try:
            calib_tf, _ = h.calib()
        except Exception:
            pass
        out = {
            "metric": "GP fit+predict throughput (points/sec) at N=100k,M=2048 per GPU; log-ml evals/sec alongside",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Cm: inducing-point sparse GP (VarDTC), N=%d rows %s, M=%d, D=%d, %s, fp64; "
                                   "step = %d bound+gradient evaluations + posterior + predict/predict_gradient on N rows"
                                   % (N, "in total, row-sharded (strong scaling)" if strong else "per GPU", M, D, args.kind,
                                      args.iters),
                       "N_per_gpu": N_local, "N_total": N if strong else world * N, "M": M, "D": D, "kernel": args.kind,
                       "evals_per_step": args.iters,
                       "sharding": ("REHEARSAL on one device, exchange over gloo: not a measurement" if rehearsal else
                                    "rows; one RCCL sum-all-reduce of the packed lower triangle of Phi | K_mn y and of the "
                                    "gradient partials per evaluation") if world > 1 else "single GPU", "stats_form": {1: "gram", 2: "stable"}.get(mode, "?")},
            "lml_evals_per_sec": n_lml / lml_elapsed,
            "lml_grad_evals_per_sec": args.steps * args.iters / elapsed,
            "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel<128,128,true,true,1> (Phi = K_mn K_nm, fp64 MFMA syrk)",
                         "achieved": achieved, "peak": FP64_MFMA_SPEC_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / FP64_MFMA_SPEC_TFLOPS) if achieved else None, "traffic": traffic,
                         "traffic_source": "profiles/" + pmc_name + ": FETCH_SIZE x 2 (gfx950 wide reads) + WRITE_SIZE per launch "
                                           "of this kernel at this shape, collected by tools/pmc_collect.py in separate --pmc "
                                           "passes -- a stored collection, not counters of this run" if traffic else None,
                         "avg_launch_ms": avg_ms, "launches": syrk_n, "flops_per_launch": flops_per_launch,
                         "peak_source": "AMD public MI355X fp64 matrix spec (not in the local guide)",
                         "peak_calibrated_tflops": calib_tf,
                         "note": "achieved = HIP events around the launches inside the timed region (the kernel has the GPU to "
                                 "itself: the O(M^3) phase 1 overlaps the K_nm build, not this kernel); isolated = the same "
                                 "kernel alone (its slab reduction included)",
                         "isolated_ms": iso_ms,
                         "isolated_tflops": (flops_per_launch / (iso_ms * 1e-3) * 1e-12) if iso_ms else None},
            # HIP events around the kernel families inside the timed region.  syrk and grad_gemm have the GPU to
            # themselves; kernel_build and small_MxM include work queued on the side stream BESIDE other kernels (the
            # Kmm factorisation runs beside the K_nm build, the Kmm gradient term beside the gradient contraction), so
            # their event spans include waiting for workgroup slots and the four do not add up to ms_per_step
            "device_ms_per_step": {"syrk": syrk_ms / args.steps, "grad_gemm": grad_ms / args.steps,
                                   "kernel_build": kb_ms / args.steps, "small_MxM": small_ms / args.steps,
                                   "note": "side-stream spans include queueing; not additive"},
            # the latency-bound kernel family: factorisation + explicit inverse of one M x M matrix out of the same
            # launches (one per 64-column step); phase 3's (I + A) runs alone, phase 1's (Kmm) beside the K_nm build
            "cholesky": {"what": "potrf + inverse of the factor, M x M fp64, %d per evaluation" % 2,
                         "avg_ms": (chol_ms / chol_n) if chol_n else None, "launches": chol_n,
                         "flops": 2.0 * M ** 3 / 3.0,
                         "tflops": (2.0 * M ** 3 / 3.0 / (chol_ms / chol_n * 1e-3) * 1e-12) if chol_n else None,
                         "frac_of_fp64_mfma_peak": (2.0 * M ** 3 / 3.0 / (chol_ms / chol_n * 1e-3) * 1e-12 / FP64_MFMA_SPEC_TFLOPS)
                         if chol_n else None,
                         # the same factorisation + inverse with the GPU to itself (random SPD matrix, 5 runs)
                         "isolated_ms": chol_iso_ms,
                         "isolated_frac_of_fp64_mfma_peak": (2.0 * M ** 3 / 3.0 / (chol_iso_ms * 1e-3) * 1e-12 /
                                                             FP64_MFMA_SPEC_TFLOPS) if chol_iso_ms else None},
        }
        if shard_check is not None:
            out["sharding_check"] = shard_check
        if args.cpu_baseline and world == 1:          # rank 0 at N=1 only, outside every timed region
            out["cpu_baseline"] = cpu_baseline(N, M, D, args.kind, args.iters)
            if lml_ref_dev is not None and out["cpu_baseline"].get("bound_full_size_value") is not None:
                ref = out["cpu_baseline"]["bound_full_size_value"]
                out["cpu_baseline"]["device_bound_rel_err"] = abs(lml_ref_dev - ref) / abs(ref)
        print(json.dumps(out))
    if world > 1:
        h.comm_destroy()
        dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
