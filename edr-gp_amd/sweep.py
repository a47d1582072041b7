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


def projection_sweep(X, y, Z_rows, projections, n_scales=16, kind="RBF", variance=1.0, noise=0.01,
                     rank=0, world_size=1, device=None, want_gradients=False, hip=None, want_gram=False):
    """Evaluate the Titsias bound (and optionally the posterior-mean gradients at the
    training rows) for every (projection, length-scale) pair owned by this rank.

    X (N, D), y (N,) standardised; Z_rows: indices of the training rows used as inducing
    inputs (projected with each candidate); projections: list of (D, d_c) matrices.
    Returns a list of dicts: candidate, scale index, lengthscale, bound, and (if asked)
    ``grad`` (N, d_c) and its back-projection ``grad @ W.T`` (N, D), or -- ``want_gram``, the
    cheap form: the gradients never leave the device -- ``gram`` = grad^T grad (d_c, d_c),
    ``gsum`` (d_c,) and the back-projected ``gram_full`` = W gram W^T (D, D), whose leading
    eigenvectors are the EDR directions of that candidate.
    """
    own = hip is None
    h = HipGP(0 if device is None else device) if own else hip
    out = []
    base_sent = False
    # one chunk for all rows when K_nm fits comfortably (3 N x M fp64 work arrays): one Phi launch per
    # evaluation and K_nm stays resident for the gradient pass over the training rows
    Mz = len(Z_rows)
    if own or 3 * X.shape[0] * Mz * 8 <= 48e9:
        try:
            h.set_option("chunk_rows", max(131072, X.shape[0]))
        except Exception:
            pass
    X = np.ascontiguousarray(X, dtype=np.float64)
    try:
        for c, W in enumerate(projections):
            W = np.asarray(W, dtype=np.float64)
            scales = length_scale_grid(W.shape[1], n_scales)
            mine = [g for g in range(n_scales) if (c * n_scales + g) % world_size == rank]
            if not mine:
                continue
            # rows stay resident: a candidate uploads W only and projects on the device
            h.set_data_projected(W, X if not base_sent else None, y if not base_sent else None)
            base_sent = True
            h.set_inducing(np.ascontiguousarray(X[Z_rows] @ W))
            for g in mine:
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
        if own:
            h.close()
        else:
            h.set_option("chunk_rows", 131072)
    return out
