"""Candidate-direction x length-scale sweep (BASELINE config 4; SURVEY.md section 8d "C4").

The EDR loop refits a GP on projected inputs and reads the posterior-mean gradients
at the training points (edrgp/base.py:459-463: ``_fit_estimator`` + ``_fit_transformer``).
A sweep evaluates that unit of work -- bound + N-point mean gradient -- for many candidate
projections ``W_c`` (D x d_c) and a grid of isotropic length-scales at fixed variance /
noise, instead of running an optimiser per candidate.  Evaluations are independent:
with ``world_size`` ranks each rank takes every ``world_size``-th (candidate, scale) pair
and the scalars are gathered by the caller (no data-path collective).
"""
from __future__ import annotations

import numpy as np

from .backend import HipGP


def length_scale_grid(d, n=16):
    """l_g = sqrt(d) * 2^((g - n/2) / 4), g = 0..n-1  (SURVEY 8d)."""
    return np.sqrt(d) * 2.0 ** ((np.arange(n) - n // 2) / 4.0)


def _sweep_worker(h, close, X, y, Z_rows, projections, n_scales, kind, variance, noise, rank, world_size,
                  want_gradients, want_gram, lane, lanes, batch=True):
    """One native context working through its share of the (candidate, scale) pairs."""
    out = []
    base_sent = False
    # one chunk for all rows when K_nm fits comfortably (3 N x M fp64 work arrays per context): one Phi
    # launch per evaluation and K_nm stays resident for the gradient pass over the training rows
    big = 3 * X.shape[0] * len(Z_rows) * 8 * lanes <= 96e9
    if big:
        h.set_option("chunk_rows", max(131072, X.shape[0]))
    try:
        for c, W in enumerate(projections):
            W = np.asarray(W, dtype=np.float64)
            scales = length_scale_grid(W.shape[1], n_scales)
            mine = [g for g in range(n_scales) if (c * n_scales + g) % world_size == rank]
            mine = mine[lane::lanes]
            if not mine:
                continue
            # rows stay resident: a candidate uploads W only and projects on the device
            h.set_data_projected(W, X if not base_sent else None, y if not base_sent else None)
            base_sent = True
            h.set_inducing(np.ascontiguousarray(X[Z_rows] @ W))
            todo = list(mine)
            if batch and want_gram and not want_gradients and kind in ("RBF", "Matern32", "Matern52") and 64 < len(Z_rows) <= 3072:
                # the scales of this candidate as ONE batch on the device (edrgp_sweep_scales): squared distances
                # formed once, every factorisation of the 2 x len(mine) advancing in the same launches
                bounds, grams, gsums, modes = h.sweep_scales(scales[todo], variance, noise, kind, want_gram=True)
                left = []
                for i, g in enumerate(todo):
                    if modes[i] < 0:                      # needs jitter: the per-unit path below climbs the ladder
                        left.append(g)
                        continue
                    out.append({"candidate": c, "scale": g, "lengthscale": float(scales[g]), "bound": float(bounds[i]),
                                "gram": grams[i], "gsum": gsums[i], "gram_full": W @ grams[i] @ W.T,
                                "form": int(modes[i])})
                todo = left
            for g in todo:
                ls = np.full((1, W.shape[1]), scales[g])
                h.set_kernel([kind], [variance], ls, noise)
                rec = {"candidate": c, "scale": g, "lengthscale": float(scales[g])}
                rec["bound"] = h.sparse_posterior()
                if want_gram:
                    _, gram, gsum = h.predict_train_gram()
                    rec["gram"], rec["gsum"] = gram, gsum
                    rec["gram_full"] = W @ gram @ W.T
                if want_gradients:
                    _, grad = h.predict_train(want_mean=False, want_grad=True)
                    rec["grad"] = grad
                    rec["grad_full"] = grad @ W.T
                out.append(rec)
    finally:
        if close:
            h.close()
        elif big:
            h.set_option("chunk_rows", 0)
    return out


def projection_sweep(X, y, Z_rows, projections, n_scales=16, kind="RBF", variance=1.0, noise=0.01,
                     rank=0, world_size=1, device=None, want_gradients=False, hip=None, want_gram=False,
                     concurrency=1, hips=None, batch=True):
    """Evaluate the Titsias bound (and optionally the posterior-mean gradients at the
    training rows) for every (projection, length-scale) pair owned by this rank.

    X (N, D), y (N,) standardised; Z_rows: indices of the training rows used as inducing
    inputs (projected with each candidate); projections: list of (D, d_c) matrices.
    Returns a list of dicts: candidate, scale index, lengthscale, bound, and (if asked)
    ``grad`` (N, d_c) and its back-projection ``grad @ W.T`` (N, D), or -- ``want_gram``, the
    cheap form: the gradients never leave the device -- ``gram`` = grad^T grad (d_c, d_c),
    ``gsum`` (d_c,) and the back-projected ``gram_full`` = W gram W^T (D, D), whose leading
    eigenvectors are the EDR directions of that candidate.

    ``batch`` (default): with ``want_gram`` (and without ``want_gradients``) the length-scales a rank owns of one
    candidate are evaluated as ONE batch on the device (``edrgp_sweep_scales``: the squared distances formed once,
    all factorisations advancing in the same launches); records then carry ``form`` (1 Gram, 2 stable statistics).
    ``batch=False`` runs them one by one -- same numbers.

    ``concurrency`` > 1: that many native contexts on the same device, each on its own host thread
    and streams, share the pairs.  One evaluation at M = 1024 spends ~40 % of its time in
    latency-bound O(M^3) launches that leave the chip almost idle; independent evaluations
    interleave those with each other's Phi kernels.  (The C ABI is thread-safe across contexts.)
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    dev = 0 if device is None else device
    lanes = max(1, int(concurrency))
    if lanes == 1:
        own = hip is None
        h = HipGP(dev) if own else hip
        out = _sweep_worker(h, own, X, y, Z_rows, projections, n_scales, kind, variance, noise, rank, world_size,
                            want_gradients, want_gram, 0, 1, batch)
    else:
        from concurrent.futures import ThreadPoolExecutor
        if hips is not None:                      # caller-owned contexts (one per lane), kept open
            ctxs, borrowed = list(hips)[:lanes], True
            lanes = len(ctxs)
        else:
            ctxs, borrowed = [hip if (i == 0 and hip is not None) else HipGP(dev if hip is None else hip.device)
                              for i in range(lanes)], False
        with ThreadPoolExecutor(max_workers=lanes) as ex:
            futs = [ex.submit(_sweep_worker, ctxs[i], not borrowed and not (i == 0 and hip is not None), X, y, Z_rows, projections, n_scales,
                              kind, variance, noise, rank, world_size, want_gradients, want_gram, i, lanes, batch)
                    for i in range(lanes)]
            out = [r for f in futs for r in f.result()]
    out.sort(key=lambda r: (r["candidate"], r["scale"]))
    return out
