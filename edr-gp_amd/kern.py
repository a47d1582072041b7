"""Stationary covariance functions with GPy's constructor signature.

``_BaseGP._make_kernel`` in the reference (edrgp/gp_model/base.py:111-147)
resolves kernel names with ``getattr(GPy.kern, name)(input_dim=D, **options)``
and sums the parts; these classes take the same arguments
(``input_dim, variance=1., lengthscale=None, ARD=False``) and hold the same
parameters (``variance``, ``lengthscale``).  The arithmetic is not here: K is
assembled by the HIP tile kernel (csrc/gp_kernels.hip).
"""
from __future__ import annotations

import numpy as np

KINDS = ("RBF", "Matern32", "Matern52")


class Stationary:
    kind = None

    def __init__(self, input_dim, variance=1.0, lengthscale=None, ARD=False, active_dims=None, name=None):
        if active_dims is not None:
            raise NotImplementedError("active_dims is not supported by the MI355X path")
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        variance = np.asarray(variance, dtype=np.float64)
        if variance.size != 1 or not variance > 0:
            raise ValueError("variance must be one positive number")
        self.variance = float(variance)
        if not self.ARD:
            if lengthscale is None:
                lengthscale = np.ones(1)
            else:
                lengthscale = np.asarray(lengthscale, dtype=np.float64)
                assert lengthscale.size == 1, "Only 1 lengthscale needed for non-ARD kernel"
        else:
            if lengthscale is not None:
                lengthscale = np.asarray(lengthscale, dtype=np.float64)
                assert lengthscale.size in [1, self.input_dim], "Bad number of lengthscales"
                if lengthscale.size != self.input_dim:
                    lengthscale = np.ones(self.input_dim) * lengthscale
            else:
                lengthscale = np.ones(self.input_dim)
        self.lengthscale = np.array(lengthscale, dtype=np.float64).reshape(-1)
        if np.any(self.lengthscale <= 0):
            raise ValueError("lengthscale must be positive")
        self.name = name or self.kind.lower()

    # -- parameters ----------------------------------------------------------
    @property
    def parts(self):
        return [self]

    def expanded_lengthscale(self):
        """Per-dimension lengthscales (an isotropic kernel repeats its value)."""
        return np.broadcast_to(self.lengthscale, (self.input_dim,)).copy()

    def copy(self):
        return type(self)(self.input_dim, self.variance, self.lengthscale.copy(), self.ARD, name=self.name)

    def __add__(self, other):
        return Add(self.parts + other.parts)

    __radd__ = lambda self, other: self if other == 0 else other.__add__(self)   # noqa: E731  (np.sum starts at 0)

    def __repr__(self):
        return "%s(input_dim=%d, variance=%r, lengthscale=%r, ARD=%r)" % (
            self.kind, self.input_dim, self.variance, self.lengthscale.tolist(), self.ARD)

    # -- covariances (device) --------------------------------------------------
    def K(self, X, X2=None):
        return _device_K(self, X, X2)

    def Kdiag(self, X):
        return np.full(np.asarray(X).shape[0], sum(p.variance for p in self.parts))


class RBF(Stationary):
    kind = "RBF"


class Matern32(Stationary):
    kind = "Matern32"


class Matern52(Stationary):
    kind = "Matern52"


class Add:
    """Sum of stationary parts (``np.sum([...])`` at gp_model/base.py:146)."""

    def __init__(self, parts):
        parts = list(parts)
        if not parts:
            raise ValueError("empty kernel sum")
        dims = {p.input_dim for p in parts}
        if len(dims) != 1:
            raise ValueError("all parts must share input_dim")
        self._parts = parts
        self.input_dim = parts[0].input_dim
        self.name = "sum"

    @property
    def parts(self):
        return self._parts

    def copy(self):
        return Add([p.copy() for p in self._parts])

    def __add__(self, other):
        return Add(self.parts + other.parts)

    def __radd__(self, other):
        return self if other == 0 else other.__add__(self)

    def K(self, X, X2=None):
        return _device_K(self, X, X2)

    def Kdiag(self, X):
        return np.full(np.asarray(X).shape[0], sum(p.variance for p in self.parts))

    def __repr__(self):
        return " + ".join(repr(p) for p in self._parts)


def _device_K(kern, X, X2):
    from .device import shared_context
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    if X.ndim != 2 or X.shape[1] != kern.input_dim:
        raise ValueError("X must be (n, %d)" % kern.input_dim)
    parts = kern.parts
    # stateless native call: the shared context's resident data / posterior are not touched
    return shared_context().hip.kernel_matrix([p.kind for p in parts], [p.variance for p in parts],
                                              np.array([p.expanded_lengthscale() for p in parts]), X,
                                              None if X2 is None else np.asarray(X2, dtype=np.float64))


def from_names(kernels, kernel_options, input_dim):
    """The reference's ``_make_kernel`` (gp_model/base.py:111-147): ``None`` ->
    default RBF; kernel object passthrough; name(s) + option dict(s) -> sum."""
    if kernels is None:
        return RBF(input_dim)
    if isinstance(kernels, (Stationary, Add)):
        return kernels.copy()
    if isinstance(kernels, str):
        kernels = [kernels]
    if isinstance(kernel_options, dict):
        kernel_options = [kernel_options]
    classes = []
    for name in kernels:
        cls = globals().get(name) if isinstance(name, str) else None
        if cls is None or not (isinstance(cls, type) and issubclass(cls, Stationary)) or cls is Stationary:
            raise AttributeError("kernel %r is not available on the MI355X path (have %s)" % (name, ", ".join(KINDS)))
        classes.append(cls)
    if kernel_options is None:
        options = [{} for _ in classes]
    elif len(classes) == len(kernel_options):
        options = [dict(o) for o in kernel_options]
    else:
        raise ValueError
    for o in options:
        o["input_dim"] = input_dim
    parts = [cls(**o) for cls, o in zip(classes, options)]
    return parts[0] if len(parts) == 1 else Add(parts)
