"""Process-wide native context per device, shared by the models.

A context owns the HBM-resident training data and workspaces.  Models created
one after another (EDR clones and refits its estimator every iteration,
edrgp/base.py:133-134) reuse the same context and its buffers; ``claim`` tells a
model whether its data is still the one resident on the device.
"""
from __future__ import annotations

import os

import numpy as np

from .backend import HipGP

_contexts = {}


# Projections announced by the EDR layer: X_proj = base @ W (edrgp/base.py:459-517 refits the GP on X C^T every
# iteration).  Keyed by the projected array's buffer address; consulted when a model sends its data, so that the
# rows need not be uploaded again -- the device projects the resident base rows (edrgp_set_data_projected).
_projections = {}


def _key(a):
    return (a.__array_interface__["data"][0], a.shape, a.strides)


def announce_projection(X_proj, base, W):
    """X_proj (C-contiguous float64) equals base @ W.  Kept until ``forget_projections()``."""
    if isinstance(X_proj, np.ndarray) and X_proj.dtype == np.float64 and X_proj.flags["C_CONTIGUOUS"]:
        _projections[_key(X_proj)] = (X_proj, np.ascontiguousarray(base, dtype=np.float64),
                                      np.ascontiguousarray(W, dtype=np.float64))
        while len(_projections) > 4:          # only the latest products matter (one per EDR iteration)
            _projections.pop(next(iter(_projections)))


def forget_projections():
    _projections.clear()


class SharedContext:
    def __init__(self, device):
        self.hip = HipGP(device)
        self.owner = None          # token of the model whose data is resident
        self._base = None          # (base rows, y) resident for device-side projections
        self.projected_sends = 0   # data sets that reached the device as W only (diagnostic)

    def send_data(self, X, y):
        """Training rows to the device: as a projection of resident base rows when the EDR layer announced
        X = base @ W (the base rows are uploaded once per EDR fit), otherwise as they are."""
        rec = _projections.get(_key(X)) if isinstance(X, np.ndarray) else None
        if rec is not None and rec[0] is X:
            _, base, W = rec
            resident = (self._base is not None and self._base[0] is base and np.array_equal(self._base[1], y))
            if resident:
                self.hip.set_data_projected(W)
            else:
                self.hip.set_data_projected(W, Xbase=base, y=y)
                self._base = (base, np.array(y, copy=True))
            self.projected_sends += 1
            return
        self._base = None
        self.hip.set_data(X, y)

    def claim(self, token):
        """Returns True when `token` already owns the resident state."""
        same = token is not None and self.owner == token
        self.owner = token
        return same


def default_device():
    return int(os.environ.get("EDRGP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def shared_context(device=None):
    if device is None:
        device = default_device()
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = SharedContext(device)
        _contexts[device] = ctx
    return ctx


def release_all():
    for ctx in _contexts.values():
        ctx.hip.close()
    _contexts.clear()
