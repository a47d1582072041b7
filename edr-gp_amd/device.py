"""Process-wide native context per device, shared by the models.

A context owns the HBM-resident training data and workspaces.  Models created
one after another (EDR clones and refits its estimator every iteration,
edrgp/base.py:133-134) reuse the same context and its buffers; ``claim`` tells a
model whether its data is still the one resident on the device.
"""
from __future__ import annotations

import os

from .backend import HipGP

_contexts = {}


class SharedContext:
    def __init__(self, device):
        self.hip = HipGP(device)
        self.owner = None          # token of the model whose data is resident

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
