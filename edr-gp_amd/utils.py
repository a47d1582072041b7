"""Subspace helpers used by the EDR estimators (API of ``edrgp.utils``:
edrgp/utils.py:8-175).  Small dense linear algebra on the (N x D) gradient
matrix; stays on the host."""
from __future__ import annotations

import numpy as np
import scipy.linalg
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_array


def ort_space(A):
    """Orthonormal basis (columns) of the orthogonal complement of span(A),
    A is [n_features, n_components] (edrgp/utils.py:8-24)."""
    U, s, _ = scipy.linalg.svd(A, full_matrices=True)
    rank = int(np.sum(np.abs(s) > 1e-10))
    return U[:, rank:]


def subspace_variance_ratio(X, V):
    """Norm of X projected on the columns of V, and its squared share of ||X||_F.
    Per column when V is orthonormal, otherwise one number for span(V)
    (edrgp/utils.py:27-55)."""
    gram = V.T @ V
    if np.allclose(gram, np.eye(V.shape[1])):
        proj = np.linalg.norm(X @ V, axis=0)
    else:
        Q = np.linalg.qr(V)[0]
        proj = np.linalg.norm(X @ Q)
    return proj, (proj / np.linalg.norm(X)) ** 2


def discrepancy(B, V):
    """||B B^T (I - V V^T)||_F / d, d = number of true components (edrgp/utils.py:58-78)."""
    residual = np.eye(B.shape[0]) - V @ V.T
    return np.linalg.norm((B @ B.T) @ residual) / B.shape[1]


class SVDTransformer(BaseEstimator, TransformerMixin):
    """PCA without centring or scaling: right singular vectors of the data
    (edrgp/utils.py:81-175).

    n_components : None (all), int, or float in (0, 1) = smallest number of
    components whose cumulative variance share reaches that value.

    Attributes: components_ (n_components, n_features), subspace_variance_,
    subspace_variance_ratio_.
    """

    def __init__(self, n_components=None):
        self.n_components = n_components

    def fit(self, X, y=None):
        Xc = check_array(X, copy=True)
        # thin SVD: only S and V^T are used.  (The reference calls np.linalg.svd with full_matrices=True,
        # edrgp/utils.py:140, which also builds the n_samples x n_samples U: 20 GB and 13 s per call at
        # 50 000 gradient rows; S and V^T are the same.)
        _, S, Vh = np.linalg.svd(Xc, full_matrices=False)
        var = S ** 2
        ratio = var / var.sum()
        k = Xc.shape[1]
        nc = self.n_components
        if isinstance(nc, (int, np.integer)) and not isinstance(nc, bool) and 0 < nc <= Xc.shape[1]:
            k = int(nc)
        elif isinstance(nc, float) and 0 < nc < 1:
            # (the reference raises NameError here: undefined `expl_var_ratio`, utils.py:148-150)
            k = int(np.sum(np.cumsum(ratio) < nc)) + 1
        elif nc is True:
            k = 1                      # examples/regression.py passes True positionally
        k = min(Xc.shape[0], k)
        self.components_ = Vh[:k, :]
        self.subspace_variance_ = var[:k]
        self.subspace_variance_ratio_ = ratio[:k]
        return self

    def transform(self, X):
        return np.asarray(X).dot(self.components_.T)


def effective_cpus():
    """CPUs this process can really use: the affinity mask capped by the cgroup CPU quota.
    A container may show every core of the host (``os.cpu_count()`` = 256 on an MI355X node)
    while its quota is a fraction of them."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, quota // int(f2.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


class host_threads:
    """Context manager: cap the BLAS / OpenMP pools of this process while the optimiser drives the device.
    The host side of an optimiser step is a few vector operations; an uncapped BLAS pool (one thread per
    visible core, spinning after each call) starves the HIP runtime's own threads inside a CPU quota and
    was measured to stretch every native evaluation from 28 to 87 ms."""

    def __init__(self, limit=None):
        self.limit = limit if limit is not None else min(4, effective_cpus())
        self._ctl = None

    def __enter__(self):
        try:
            from threadpoolctl import threadpool_limits
            self._ctl = threadpool_limits(limits=self.limit)
        except Exception:
            self._ctl = None
        return self

    def __exit__(self, *exc):
        if self._ctl is not None:
            try:
                self._ctl.restore_original_limits()
            except Exception:
                pass
        return False
