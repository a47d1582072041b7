"""Synthetic data generators with the signatures of ``edrgp.datasets``
(edrgp/datasets.py:7-91).  They draw from numpy's global RNG, as the reference does."""
from __future__ import annotations

from math import pi

import numpy as np
from scipy.stats import special_ortho_group
from sklearn.utils import check_array


def get_gaussian_inputs(sample_size, eig_values, eig_vectors=None, mean=None):
    """N(mean, V diag(eig_values) V^T) samples; V random orthogonal when not given."""
    dim = len(eig_values)
    if eig_vectors is None:
        V = special_ortho_group.rvs(dim)
    else:
        V = check_array(eig_vectors)
        if V.shape != (dim, dim):
            raise ValueError("eig_vectors shape must be ({0},{0})".format(dim))
    cov = (V * np.asarray(eig_values, dtype=float)) @ V.T
    mu = np.zeros(dim) if mean is None else mean
    return np.random.multivariate_normal(mu, cov, sample_size)


def get_tanh_targets(X, coefs, bias=0, noise_std=0.05):
    """tanh(X w + b) + noise_std * eps."""
    if X.shape[1] != len(coefs):
        raise ValueError("Dimensionality of input ({}) and coefs ({}) are mismatched".format(X.shape[1], len(coefs)))
    return np.tanh(X @ np.asarray(coefs) + bias) + noise_std * np.random.randn(X.shape[0])


def get_beta_inputs(sample_size, ndim, tau=1):
    """2 * Beta(1, tau) - 1 per coordinate."""
    return 2 * np.random.beta(1, tau, size=(sample_size, ndim)) - 1


def get_edr_target(X, sigma=None):
    """Test functions of 1, 2 or 3 effective coordinates."""
    d = X.shape[1]
    u = X.T
    if d == 1:
        out = u[0] * np.sin(np.sqrt(5) * u[0])
    elif d == 2:
        out = (u[0] ** 3 + u[1]) * (u[0] - u[1] ** 3)
    elif d == 3:
        out = (u[0] ** 3 + u[1]) * (u[0] - u[1] ** 3) + u[2]
    else:
        raise ValueError("effective dimension must be 1, 2 or 3")
    out = np.array(out, dtype=float).ravel()
    if sigma is not None:
        out += sigma * np.random.randn(out.size)
    return out


def get_branin_targets(X, noise_std=None):
    """Branin function on [0, 1]^2 (https://www.sfu.ca/~ssurjano/branin.html)."""
    a, b, c, r, s, t = 1, 5.1 / (4 * pi ** 2), 5 / pi, 6, 10, 1 / (8 * pi)
    x0 = 15 * X[:, 0] - 5
    x1 = 15 * X[:, 1]
    y = a * (x1 - b * x0 ** 2 + c * x0 - r) ** 2 + s * (1 - t) * np.cos(x0) + s
    if noise_std is not None:
        y += noise_std * np.random.randn(X.shape[0])
    return y
