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


class GradientGram:
    """An (n_samples x n_features) gradient matrix known through its Gram matrix G^T G (and column sums): what the
    EDR step needs of the posterior-mean gradients at the training rows (edrgp/base.py:459-463 feeds them to a
    linear decomposition, edrgp/utils.py:27-55 to norms of projections).  The MI355X estimators reduce G^T G on the
    device (``edrgp_predict_train_gram``), so the N x D matrix itself never crosses PCIe; ``materialize()`` fetches
    it on demand (row-indexed refits, BlockEDR, user access) through the callable given at construction."""

    def __init__(self, gram, n_samples, gsum=None, fetch=None):
        self.gram = np.asarray(gram, dtype=np.float64)
        self.gsum = None if gsum is None else np.asarray(gsum, dtype=np.float64)
        self.shape = (int(n_samples), self.gram.shape[0])
        self._fetch = fetch
        self._dense = None

    def materialize(self):
        if self._dense is None:
            if self._fetch is None:
                raise ValueError("the gradient matrix behind this Gram matrix is not available")
            self._dense = np.asarray(self._fetch(), dtype=np.float64)
        return self._dense

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def mapped(self, P):
        """Gram matrix of G P (P: n_features x k)."""
        fetch = None if self._fetch is None else (lambda: self.materialize() @ P)
        return GradientGram(P.T @ self.gram @ P, self.shape[0], None if self.gsum is None else self.gsum @ P, fetch)


def subspace_variance_ratio(X, V):
    """Norm of X projected on the columns of V, and its squared share of ||X||_F.
    Per column when V is orthonormal, otherwise one number for span(V)
    (edrgp/utils.py:27-55).  X may be a ``GradientGram``: ||X v||^2 = v^T (X^T X) v."""
    gram = V.T @ V
    ortho = np.allclose(gram, np.eye(V.shape[1]))
    if isinstance(X, GradientGram) and X._dense is not None:
        X = X._dense          # already fetched (e.g. by fit_gram below its accuracy floor): the reference's own arithmetic
    if isinstance(X, GradientGram):
        if ortho:
            proj = np.sqrt(np.clip(np.einsum("ij,ik,kj->j", V, X.gram, V), 0.0, None))
        else:
            Q = np.linalg.qr(V)[0]
            proj = np.sqrt(max(np.trace(Q.T @ X.gram @ Q), 0.0))
        return proj, (proj / np.sqrt(np.trace(X.gram))) ** 2
    if ortho:
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
        return self._select(S ** 2, Vh, Xc.shape[0])

    def _select(self, var, Vh, n_rows):
        ratio = var / var.sum()
        k = Vh.shape[1]
        nc = self.n_components
        if isinstance(nc, (int, np.integer)) and not isinstance(nc, bool) and 0 < nc <= Vh.shape[1]:
            k = int(nc)
        elif isinstance(nc, float) and 0 < nc < 1:
            # (the reference raises NameError here: undefined `expl_var_ratio`, utils.py:148-150)
            k = int(np.sum(np.cumsum(ratio) < nc)) + 1
        elif nc is True:
            k = 1                      # examples/regression.py passes True positionally
        k = min(n_rows, k)
        self.components_ = Vh[:k, :]
        self.subspace_variance_ = var[:k]
        self.subspace_variance_ratio_ = ratio[:k]
        return self

    def fit_gram(self, G):
        """The same decomposition from the Gram matrix X^T X alone (a ``GradientGram``): the right singular vectors
        of X are the eigenvectors of X^T X, the squared singular values its eigenvalues.  Signs: a singular vector
        is defined up to sign; here each component is oriented along the column sums of X (largest entry positive
        when those are unknown or orthogonal to it)."""
        w, V = np.linalg.eigh(G.gram)
        # Accuracy floor: forming X^T X squares the condition number, so singular values below ~1e-6 of the largest
        # (eigenvalues below 1e-12 of the largest) are rounding noise here where the SVD of X resolves them.  A
        # numerically rank-deficient gradient matrix therefore goes back to the dense decomposition whenever the
        # matrix itself can be fetched (the reference's path, edrgp/utils.py:140); without it the noise eigenvalues
        # are clipped to 0 and only their (arbitrary, orthonormal) directions remain.
        if w.max() > 0 and w.min() < 1e-12 * w.max() and getattr(G, "_fetch", None) is not None:
            return self.fit(G.materialize())
        order = np.argsort(w)[::-1]
        var = np.clip(w[order], 0.0, None)
        Vh = V[:, order].T.copy()
        for i in range(Vh.shape[0]):
            d = 0.0 if G.gsum is None else float(Vh[i] @ G.gsum)
            if abs(d) <= 1e-12 * max(1.0, float(np.abs(G.gsum).max()) if G.gsum is not None else 1.0):
                d = Vh[i, np.argmax(np.abs(Vh[i]))]
            if d < 0:
                Vh[i] = -Vh[i]
        return self._select(var, Vh, G.shape[0])

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
