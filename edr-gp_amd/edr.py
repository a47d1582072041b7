"""``EffectiveDimensionalityReduction``: the public EDR estimator
(edrgp/edr.py:9-289) -- IterativeEDR plus input standardisation and an optional
linear preprocessor (PCA ...)."""
from __future__ import annotations

import numpy as np
from sklearn.base import clone
from sklearn.preprocessing import StandardScaler
from sklearn.utils import check_array
from sklearn.utils.validation import check_is_fitted

from .base import IterativeEDR, _require_components


class EffectiveDimensionalityReduction(IterativeEDR):
    """Parameters
    ----------
    estimator : GP regressor exposing ``fit`` / ``predict_gradient``
        (``edrgp_amd.gp_model.GaussianProcessRegressor`` / ``Sparse...``).
    dr_transformer : linear transformer exposing ``components_`` (PCA, SVDTransformer ...).
    n_components, step : see ``IterativeEDR``.
    normalize : standardise inputs before the GP (components are mapped back).
    preprocessor : optional linear map applied after scaling (needs normalize=True).
    """

    def __init__(self, estimator=None, dr_transformer=None, n_components=None, step=None, normalize=True,
                 preprocessor=None):
        self.estimator = estimator
        self.dr_transformer = dr_transformer
        self.n_components = n_components
        self.step = step
        self.normalize = normalize
        self.preprocessor = preprocessor

    # the base classes talk to ``self.transformer``
    @property
    def transformer(self):
        return self.dr_transformer

    # -- fitting -----------------------------------------------------------------
    def fit(self, X, y=None, **opt_kws):
        self.fitted = False
        Xp = self._preprocessing_fit(X)
        super().fit(Xp, y, **opt_kws)
        if self.normalize:
            # express the directions in the original (unscaled) coordinates
            self.components_ = self.components_ @ self._reverse_scaling_
        self.fitted = True
        return self

    def refit(self, refit_transformer, rows=None):
        super().refit(refit_transformer, rows)
        if hasattr(self, "refit_components_") and self.normalize:
            self.refit_components_ = self.refit_components_ @ self._reverse_scaling_

    def _preprocessing_fit(self, X, transform=True):
        if not self.normalize:
            if self.preprocessor is not None:
                raise ValueError("To apply prerpocessing, normalize should be True")
            return X
        self.scaler_ = StandardScaler()
        Xs = self.scaler_.fit_transform(X)
        self._scaling_ = np.diag(self.scaler_.scale_)
        self._reverse_scaling_ = np.diag(1.0 / self.scaler_.scale_)
        if self.preprocessor is not None:
            self.preprocessor_ = clone(self.preprocessor)
            Xs = self.preprocessor_.fit_transform(Xs)
            _require_components(self.preprocessor_)
            self._preprocessing_ = self.preprocessor_.components_
        return Xs if transform else None

    def _preprocessing_transform(self, X):
        X = check_array(X)
        if self.normalize is True:
            check_is_fitted(self, "scaler_")
            X = self.scaler_.transform(X) @ self._scaling_
        return X @ self.components_.T

    # -- gradients ------------------------------------------------------------------
    def get_estimator_gradients(self, X):
        return self._get_estimator_gradients(check_array(X), True)

    def _get_estimator_gradients(self, X, preprocessing_transform=False):
        if preprocessing_transform:
            X = self._preprocessing_transform(X)
        check_is_fitted(self, "estimator_")
        grad = self.estimator_.predict_gradient(X)
        if self.preprocessor is not None and self.num_iter == 0 and not preprocessing_transform:
            # first iteration: gradients live in preprocessor coordinates, map them back
            check_is_fitted(self, "preprocessor_")
            grad = grad @ self._preprocessing_
        if preprocessing_transform:
            grad = grad @ self.components_
        return grad

    def _get_estimator_gradient_gram(self, X):
        G = super()._get_estimator_gradient_gram(X)
        if G is not None and self.preprocessor is not None and self.num_iter == 0:
            check_is_fitted(self, "preprocessor_")
            G = G.mapped(self._preprocessing_)       # first iteration: gradients live in preprocessor coordinates
        return G

    @property
    def feature_importances_(self):
        check_is_fitted(self, "components_")
        imp = self.components_
        return imp @ self._scaling_ if self.normalize is True else imp

    def transform(self, X, refitted=False):
        check_is_fitted(self, "components_")
        X = check_array(X)
        if refitted:
            check_is_fitted(self, ["refit_transformer_", "refit_components_"])
            return X @ self.refit_components_.T
        comps = self.components_
        if not self.fitted and self.preprocessor is not None:
            comps = comps @ self._preprocessing_.T     # still inside fit: X is in preprocessor coordinates
        return self._project(X, comps.T) if not self.fitted else X @ comps.T
