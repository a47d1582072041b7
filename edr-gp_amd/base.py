"""EDR estimators: gradient-based effective dimensionality reduction.

Public behaviour of ``edrgp.base`` (edrgp/base.py:13-766): fit a GP, take the
gradient of its posterior mean at the training points (the MI355X hot path:
``estimator_.predict_gradient``), decompose the (N x D) gradient matrix with a
linear transformer and keep its leading directions; optionally iterate on the
projected data (IterativeEDR) or work block-wise (BlockEDR).  This layer is
O(N D^2) numpy and stays on the host.
"""
from __future__ import annotations

import warnings
from copy import deepcopy

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin, clone
from sklearn.preprocessing import normalize as _row_normalize
from sklearn.utils import check_array, check_X_y
from sklearn.utils.validation import check_is_fitted

from .utils import GradientGram, subspace_variance_ratio


def _require_components(transformer):
    if not hasattr(transformer, "components_"):
        raise AttributeError('The transformer does not expose "components_" attribute')


def _drop_zero_rows(components):
    """Remove all-zero component rows, warning about which ones."""
    norms = np.linalg.norm(components, axis=1)
    dead = [int(i) for i in np.flatnonzero(norms == 0)]
    if dead:
        warnings.warn("Components with numbers {} will be droped because they contains only zeros".format(dead),
                      RuntimeWarning)
    return components[norms != 0]


class BaseEDR(BaseEstimator, TransformerMixin):
    """One-shot EDR: GP fit -> gradients -> linear decomposition -> projection.

    Parameters: estimator (exposes fit / predict_gradient), transformer (exposes
    fit and ``components_``), n_components (None = all).  ``step`` is accepted and
    ignored, as in the reference (edrgp/base.py:71-75).

    Attributes after fit: components_, estimator_, first_estimator_, transformer_,
    subspace_gradients_, subspace_variance_, subspace_variance_ratio_.
    """

    # Device-side reduction (MI355X addition, no constructor parameter: the reference signature stays): when the
    # estimator offers ``predict_gradient_gram`` and the transformer ``fit_gram`` (SVDTransformer), the gradients at
    # the training rows are consumed as their D x D Gram matrix, reduced on the device -- no N x D copy to the host,
    # no N x D SVD.  Components agree with the host path up to the sign of a singular vector.  False = always the
    # reference's dense path.
    device_reduction = True

    def __init__(self, estimator=None, transformer=None, n_components=None, step=None):
        self.estimator = estimator
        self.transformer = transformer
        self.n_components = n_components
        self.step = step               # unused here; stored so get_params()/clone() work

    # -- pieces ------------------------------------------------------------------
    def _check_init(self, n_features):
        if self.estimator is None:
            raise ValueError("Estimator should be speciified")
        if self.transformer is None:
            raise ValueError("transformer should be specified")
        self.n_components_ = n_features if self.n_components is None else self.n_components

    def _fit_estimator(self, X, y, **opt_kws):
        """clone + fit the GP on (X, y) (edrgp/base.py:115-140)."""
        X, y = check_X_y(X, y, accept_sparse=False)
        if y is not None:
            self.estimator_ = clone(self.estimator)
            self.estimator_.fit(X, y, **opt_kws)
            if self.num_iter == 0:
                self.first_estimator_ = clone(self.estimator_)
        elif not hasattr(self, "estimator_"):
            self.estimator_ = clone(self.estimator)
        return self

    def _get_estimator_gradients(self, X):
        check_is_fitted(self, "estimator_")
        return self.estimator_.predict_gradient(X)

    def get_estimator_gradients(self, X):
        return self._get_estimator_gradients(check_array(X))

    def _get_estimator_gradient_gram(self, X):
        """``GradientGram`` of the estimator's gradients at X, or None when the device-side reduction does not
        apply (estimator / transformer without the Gram interface, X not the estimator's training set)."""
        if not (self.device_reduction and hasattr(self.estimator_, "predict_gradient_gram")
                and hasattr(self.transformer, "fit_gram")):
            return None
        res = self.estimator_.predict_gradient_gram(X)
        if res is None:
            return None
        gram, gsum, n = res
        est, Xk = self.estimator_, X                # the dense matrix stays fetchable (refit by rows, user access)
        return GradientGram(gram, n, gsum, fetch=lambda: est.predict_gradient(Xk))

    def _decompose(self, X):
        """Gradients at X -> freshly fitted transformer and its components."""
        check_is_fitted(self, "estimator_")
        self.transformer_ = clone(self.transformer)
        grad = self._get_estimator_gradient_gram(X)
        if grad is not None:
            self.transformer_.fit_gram(grad)
        else:
            grad = self._get_estimator_gradients(X)
            self.transformer_.fit(grad)
        if self.num_iter == 0:
            self._first_gradients_ = grad
        _require_components(self.transformer_)
        return grad, deepcopy(self.transformer_.components_)

    def _chain(self, components):
        """Compose new components (in projected coordinates) with the current map."""
        return components if self.components_ is None else components @ self.components_

    def _fit_transformer(self, X):
        _, comps = self._decompose(X)
        self.components_ = self._chain(comps[:self.n_components_, :])
        return self

    def _last_fit(self, X, y, **opt_kws):
        """Final GP on the projected data and the subspace statistics (edrgp/base.py:172-200)."""
        self._fit_estimator(X, y, **opt_kws)
        self.subspace_gradients_ = self._get_estimator_gradients(X)
        self._recovered_gradients_ = self.subspace_gradients_ @ self.components_
        self.subspace_variance_, self.subspace_variance_ratio_ = subspace_variance_ratio(
            self._first_gradients_, self.components_.T)
        return self

    # -- sklearn API -----------------------------------------------------------------
    def fit(self, X, y=None, **opt_kws):
        self._check_init(X.shape[1])
        self.components_ = None
        self.num_iter = 0
        self._fit_estimator(X, y, **opt_kws)
        self._fit_transformer(X)
        X_proj = self.transform(X)
        self.num_iter += 1
        self._last_fit(X_proj, y, **opt_kws)
        return self

    def refit(self, refit_transformer, index=None):
        """Decompose the first-iteration gradients again with another transformer."""
        check_is_fitted(self, "components_")
        rows = slice(None) if index is None else index
        grads = self._first_gradients_[rows, :]
        self.refit_transformer_ = clone(refit_transformer)
        self.refit_transformer_.fit(grads)
        _require_components(self.refit_transformer_)
        comps = _row_normalize(deepcopy(self.refit_transformer_.components_), axis=1)
        self.refit_components_ = _drop_zero_rows(comps)
        self.refit_subspace_variance_, self.refit_subspace_variance_ratio_ = subspace_variance_ratio(
            grads, self.refit_components_.T)
        return self

    def transform(self, X, refitted=False):
        check_is_fitted(self, "components_")
        X = check_array(X)
        if refitted:
            check_is_fitted(self, ["refit_transformer_", "refit_components_"])
            return X @ self.refit_components_.T
        return self._project(X, self.components_.T)

    def _project(self, X, W):
        """X W on the host; the product is also announced to the device layer, so that a GP fitted on it next
        (the EDR loop, edrgp/base.py:459-463) projects the resident rows on the device instead of uploading them."""
        Xp = np.ascontiguousarray(X @ W)
        if self.device_reduction:
            try:
                from .device import announce_projection
                announce_projection(Xp, X, W)
            except Exception:      # noqa: BLE001  (no native library: nothing to announce to)
                pass
        return Xp

    def inverse_transform(self, X):
        check_is_fitted(self, "components_")
        return check_array(X) @ np.linalg.pinv(self.components_).T

    @property
    def feature_importances_(self):
        check_is_fitted(self, "components_")
        return self.components_

    # kept for subclasses / callers that use the reference's helper names
    def _check_transformer(self, transformer):
        _require_components(transformer)

    def _remove_zero_components(self, components):
        return _drop_zero_rows(components)


class IterativeEDR(BaseEDR):
    """EDR that drops directions step by step, refitting the GP on the projected
    data each time (edrgp/base.py:346-517).

    step : None (single pass), int > 0 (directions dropped per iteration; needs
    n_components < n_features) or float in (0, 1) (keep the directions explaining
    that share of the gradient variance; n_components must be None).
    """

    def __init__(self, estimator=None, transformer=None, n_components=None, step=None):
        self.estimator = estimator
        self.transformer = transformer
        self.n_components = n_components
        self.step = step

    def _check_step(self, n_features):
        self.adaptive_step = False
        step = self.step
        if step is None:
            self.step_ = self.n_components_
        elif isinstance(step, int) and step > 0:
            if self.n_components_ == n_features:
                raise ValueError("If step is int (n_components < n_features) must be True")
            self.step_ = step
        elif isinstance(step, float) and 0 < step < 1:
            if self.n_components is not None:
                raise ValueError("If step is float n_components should be None")
            self.adaptive_step = True
            self.step_ = step
        else:
            raise ValueError("Step should be None or int > 0 or float from 0 to 1")

    def fit(self, X, y=None, **opt_kws):
        n_features = X.shape[1]
        self._check_init(n_features)
        self._check_step(n_features)
        self.components_ = None
        self.continue_iteration = True
        self.num_iter = 0
        X_proj = X.copy()
        while self.continue_iteration:
            self._fit_estimator(X_proj, y, **opt_kws)
            self._fit_transformer(X_proj)
            X_proj = self.transform(X)
            self.num_iter += 1
        self._last_fit(X_proj, y, **opt_kws)
        return self

    def _fit_transformer(self, X):
        grad, comps = self._decompose(X)
        keep = self._select_n_components(grad, comps)
        self.components_ = self._select_best_components(comps, keep)
        return self

    def _select_n_components(self, grad, components):
        width = grad.shape[1]
        if self.adaptive_step:
            _, share = subspace_variance_ratio(grad, components.T)
            keep = int(np.sum(np.cumsum(share) < self.step_)) + 1
            if keep == width:
                self.continue_iteration = False
        else:
            keep = max(self.n_components_, width - self.step_)
            if keep == self.n_components_:
                self.continue_iteration = False
        return keep

    def _select_best_components(self, components, n_components):
        self.components_ = self._chain(components)
        _, share = subspace_variance_ratio(self._first_gradients_, self.components_.T)
        order = np.argsort(share)[-n_components:][::-1]
        return self.components_[order, :]


class BlockEDR(BaseEDR):
    """EDR with one transformer per block of features; the result is block diagonal
    (edrgp/base.py:520-766).

    blocks : list of column-index lists (None = one block with every feature);
    n_components : int or list (one per block).
    """

    def __init__(self, estimator, transformer, n_components=None, blocks=None):
        self.estimator = estimator
        self.transformer = transformer
        self.blocks = blocks
        self.n_components = n_components

    def _make_blocks(self, n_features):
        nc = self.n_components_
        if self.blocks is None:
            if not isinstance(nc, int):
                raise ValueError("blocks should be specified if n_components is list")
            self.blocks_ = [{"columns": np.arange(n_features), "n_components": nc}]
        elif isinstance(self.blocks, list):
            if isinstance(nc, list):
                self.blocks_ = [{"columns": b, "n_components": k} for b, k in zip(self.blocks, nc)]
            elif isinstance(nc, int):
                self.blocks_ = [{"columns": b, "n_components": max(self.n_components, len(b))} for b in self.blocks]
        return self

    def _fit_single_block(self, transformer, grads, block, index=None, params=None):
        rows = np.ones(grads.shape[0], dtype=bool) if index is None else index
        sub = grads[np.ix_(rows, block["columns"])]
        transformer.set_params(n_components=block["n_components"])
        if params is not None:
            transformer.set_params(**params)
        transformer.fit(sub)
        return transformer.components_.T

    def _merge_components(self, components):
        n_features = self._first_gradients_.shape[1]
        total = sum(c.shape[1] for c in components)
        merged = np.zeros((n_features, total))
        start = 0
        for blk, comp in zip(self.blocks_, components):
            stop = start + comp.shape[1]
            merged[blk["columns"], start:stop] = comp
            blk["columns"] = np.arange(start, stop)
            start = stop
        return merged.T

    def _fit_transformer(self, X):
        check_is_fitted(self, "estimator_")
        self._make_blocks(X.shape[1])
        grad = self._get_estimator_gradients(X)           # block-wise decompositions need the dense matrix
        if self.num_iter == 0:
            self._first_gradients_ = grad
        per_block = [self._fit_single_block(clone(self.transformer), grad, blk) for blk in self.blocks_]
        self.components_ = self._merge_components(per_block)
        return self

    def refit(self, refit_transformer, index=None, params=None):
        self._make_blocks(self._first_gradients_.shape[1])
        self.refit_transformer_ = clone(refit_transformer)
        n = self._first_gradients_.shape[0]
        self.refit_index_ = np.ones(n, dtype=bool) if index is None else index
        per_block = [self._fit_single_block(refit_transformer, self._first_gradients_, blk, index=self.refit_index_,
                                            params=None if params is None else params[i])
                     for i, blk in enumerate(self.blocks_)]
        self.refit_components_ = _drop_zero_rows(_row_normalize(self._merge_components(per_block)))
        self.refit_subspace_variance_, self.refit_subspace_variance_ratio_ = subspace_variance_ratio(
            self._first_gradients_[self.refit_index_, :], self.refit_components_.T)
        return self
