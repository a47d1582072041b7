"""Worker of the multi-process sharding tests (tests/test_sharded_native.py, tests/test_multi_gpu.py); launched as
    python -m torch.distributed.run --nproc-per-node W tests/sharded_worker.py --transport {gloo,rccl} --out DIR

Every rank holds a RAGGED row block of (X, y) and the same Z / hyper-parameters and evaluates through the native
library (csrc/comm.hip exchange points: [N, y'y], the packed lower triangle of [Phi | b], the gradient partials):
  --transport rccl : one GPU per rank (LOCAL_RANK), RCCL all-reduce (edrgp_comm_init)
  --transport gloo : every rank on device 0, the all-reduce supplied by the launcher (edrgp_comm_init_host over a
                     gloo group) -- the same native control flow with several processes on ONE device
Rank 0 also evaluates each case whole, on one context without a communicator, and writes everything to DIR for the
parent test to compare.  Cases straddle `chunk_rows` (one rank chunked, another on the single-chunk path) and
include a Kmm whose pivot-ratio proxy sends the automatic mode into the stable form, with the sync-free driver on
and off: the number and content of the collectives must not depend on a rank's local row count."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name, N, M, D, kind, lengthscale factor (x sqrt(D)), noise, chunk_rows, stats_mode, sync_free
# A name that starts with "fresh_" runs on a NEW context (and communicator): no evaluation has told it which form
# the previous Kmm took, so its first call -- a bound-only evaluation -- starts out assuming the Gram form on an
# ill-conditioned Kmm.  With shards straddling chunk_rows that is the state in which a chunked rank used to queue
# its row pass (one all-reduce) beside phase 1 and repeat it in the stable form (a second one) while the
# single-chunk rank all-reduced once.
CASES = [
    ("fresh_illcond_auto",    2301, 96, 3, "RBF", 0.45, 0.01, 1024, 0, 1),
    ("fresh_illcond_careful", 2301, 96, 3, "RBF", 0.45, 0.01, 1024, 0, 0),
    ("gram_auto",      2301, 96, 5, "RBF",      0.5, 0.05, 1024, 0, 1),
    ("gram_careful",   2301, 96, 5, "RBF",      0.5, 0.05, 1024, 0, 0),
    ("stable_forced",  1777, 64, 4, "Matern52", 1.0, 0.05, 1024, 2, 1),
    ("illcond_auto",   2301, 96, 3, "RBF",      0.45, 0.01, 1024, 0, 1),     # pivot-ratio proxy ~3e3: the rule picks the stable form
    ("illcond_careful", 2301, 96, 3, "RBF",     0.45, 0.01, 1024, 0, 0),
    ("padded_M",       3000, 300, 6, "RBF",     0.45, 0.02, 1024, 0, 1),    # phantom inducing points (M -> 384)
]


def problem(N, D, M, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    Z = X[rng.permutation(N)[:M]].copy()
    return X, y, Z


def shard_bounds(N, world):
    """Ragged: rank r gets a share proportional to 1 + (r % 3); with world = 2 about 1/3 : 2/3."""
    w = np.array([1.0 + (r % 3) for r in range(world)])
    cuts = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * N).astype(int)])
    cuts[-1] = N
    return cuts


def evaluate(h, X, y, Z, kind, ls, noise, Xs):
    D = X.shape[1]
    h.set_data(X, y)
    h.set_inducing(Z)
    h.set_kernel([kind], [1.0], np.full((1, D), ls), noise)
    out = {"bound_only": h.sparse_lml()}
    lml, dvar, dlen, dnoise, dZ = h.sparse_lml_grad()
    out.update(lml=lml, dvar=dvar, dlen=dlen, dnoise=dnoise, dZ=dZ)
    info = h.get("MODE", (4,))
    out["mode"], out["retries"] = info[0], info[3]
    out["bound_post"] = h.sparse_posterior()
    out["mean"], out["var"] = h.predict(Xs)
    out["wv"] = h.get("WV", (Z.shape[0],))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transport", choices=["gloo", "rccl"], required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--sweep", type=int, default=1)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")                     # control plane (and, with --transport gloo, the data plane)
    from edrgp_amd.backend import HipGP
    dev = local_rank if args.transport == "rccl" else 0
    calls = []

    def allreduce(a):
        calls.append(a.size)
        dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM)

    def sharded_context():
        ctx = HipGP(dev)
        if args.transport == "rccl":
            box = [HipGP.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.comm_init(world, rank, box[0])
        else:
            ctx.comm_init_host(world, rank, allreduce)
        return ctx

    h = sharded_context()
    whole = HipGP(dev) if rank == 0 else None
    try:
        for ci, (name, N, M, D, kind, lsf, noise, chunk, smode, sfree) in enumerate(CASES):
            if name.startswith("fresh_") and ci > 0:
                h.close()
                h = sharded_context()
            X, y, Z = problem(N, D, M, seed=100 + ci)
            Xs = X[:64].copy()
            cuts = shard_bounds(N, world)
            lo, hi = int(cuts[rank]), int(cuts[rank + 1])
            ls = lsf * np.sqrt(D)
            for ctx in (h, whole):
                if ctx is not None:
                    ctx.set_option("chunk_rows", chunk)
                    ctx.set_option("stats_mode", smode)
                    ctx.set_option("sync_free", sfree)
            n0 = len(calls) if args.transport == "gloo" else 0
            got = evaluate(h, X[lo:hi], y[lo:hi], Z, kind, ls, noise, Xs)
            if args.transport == "gloo":
                got["collectives"] = np.array(calls[n0:])
            got["rows"] = np.array([lo, hi])
            np.savez(os.path.join(args.out, "%s_rank%d.npz" % (name, rank)), **got)
            if whole is not None:
                np.savez(os.path.join(args.out, "%s_whole.npz" % name), **evaluate(whole, X, y, Z, kind, ls, noise, Xs))
            dist.barrier()
        if args.sweep:
            # BASELINE config 4: the (candidate, scale) pairs are partitioned over the ranks, no data-path collective
            from edrgp_amd.sweep import projection_sweep
            h.comm_destroy()
            rng = np.random.default_rng(3)
            N, D, M = 3000, 8, 64
            X, y, _ = problem(N, D, M, seed=9)
            projections = [np.linalg.qr(rng.standard_normal((D, D - c)))[0] for c in range(3)]
            Zrows = rng.permutation(N)[:M]
            mine = projection_sweep(X, y, Zrows, projections, n_scales=4, rank=rank, world_size=world, hip=h, want_gram=True)
            parts = [None] * world
            dist.all_gather_object(parts, [(r["candidate"], r["scale"], r["bound"], r["gram"]) for r in mine])
            if rank == 0:
                full = projection_sweep(X, y, Zrows, projections, n_scales=4, hip=whole, want_gram=True)
                ref = {(r["candidate"], r["scale"]): (r["bound"], r["gram"]) for r in full}
                seen = [k[:2] for p in parts for k in p]
                ok = sorted(seen) == sorted(ref) and all(
                    b == ref[(c, s)][0] and np.array_equal(g, ref[(c, s)][1]) for p in parts for (c, s, b, g) in p)
                np.savez(os.path.join(args.out, "sweep.npz"), ok=np.array(ok), n=np.array(len(seen)), n_ref=np.array(len(ref)))
            dist.barrier()
    finally:
        h.close()
        if whole is not None:
            whole.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
