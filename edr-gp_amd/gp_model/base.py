"""Estimator base: the reference's ``_BaseGP`` contract (edrgp/gp_model/base.py:15-257)
-- ``fit / predict / predict_variance / predict_gradient / save / load`` and the
attributes ``estimator_``, ``n_features_`` -- on top of the HIP models."""
from __future__ import annotations

from abc import ABCMeta, abstractmethod

import numpy as np
from sklearn.base import BaseEstimator
from sklearn.utils import assert_all_finite, check_array, check_X_y
from sklearn.utils.validation import check_is_fitted

from .. import kern as _kern
from ..models import load_model


class _BaseGP(BaseEstimator, metaclass=ABCMeta):
    """Parameters / attributes: see the concrete regressors.

    ``fit`` forwards ``**opt_kws`` to ``estimator_.optimize`` (or
    ``optimize_restarts`` when ``method='optimize_restarts'``) with the reference's
    defaults ``messages=False, max_iters=1000`` (edrgp/gp_model/base.py:67-69).
    """

    def fit(self, X, y, **opt_kws):
        X, y = self._check_data(X, y)
        self.n_features_ = X.shape[1]
        kernel = self._make_kernel()
        self.estimator_ = self._get_model(X, y, kernel)
        opt_kws.setdefault("messages", False)
        opt_kws.setdefault("max_iters", 1000)
        getattr(self.estimator_, self.method)(**opt_kws)
        return self

    def _check_data(self, X, y):
        X, y = check_X_y(X, y, accept_sparse=False)
        return X, y[:, np.newaxis]

    def _check_input(self, X):
        X = check_array(X, accept_sparse=False)
        if X.shape[1] != self.n_features_:
            raise ValueError("X has {} features per sample; expecting {}".format(X.shape[1], self.n_features_))
        return X

    def _make_kernel(self):
        """Kernel names / option dicts -> kernel object, as the reference does with
        ``GPy.kern`` (base.py:111-147), including its in-place normalisation of
        ``kernels`` / ``kernel_options`` to lists."""
        if self.kernels is None:
            return None
        if isinstance(self.kernels, (_kern.Stationary, _kern.Add)):
            return self.kernels.copy()
        if isinstance(self.kernels, str):
            self.kernels = [self.kernels]
        if isinstance(self.kernel_options, dict):
            self.kernel_options = [self.kernel_options]
        return _kern.from_names(self.kernels, self.kernel_options, self.n_features_)

    @abstractmethod
    def _get_model(self, X, y, kernel):
        pass

    def _check_predict(self, X):
        X = self._check_input(X)
        check_is_fitted(self, "estimator_")
        return X

    def predict(self, X):
        """Posterior mean at X, shape (n_samples,)."""
        X = self._check_predict(X)
        y_pred = self.estimator_.predict(X)[0][:, 0]
        assert_all_finite(y_pred)
        return y_pred

    def predict_variance(self, X):
        """Predictive variance (noise included) at X, shape (n_samples, 1)."""
        X = self._check_predict(X)
        return self.estimator_.predict(X)[1]

    def predict_gradient(self, X):
        """Gradient of the posterior mean at X, shape (n_samples, n_features)."""
        X = self._check_predict(X)
        return self.estimator_.predictive_gradients(X)[0][:, :, 0]

    def predict_gradient_gram(self, X):
        """An addition to the reference interface: ``(G^T G, column sums of G, n_samples)`` for G = predict_gradient(X)
        when X is the training set -- reduced on the device, the n_samples x n_features matrix G stays in HBM -- and
        None otherwise.  The EDR estimators use it when their transformer can decompose a Gram matrix
        (``SVDTransformer.fit_gram``)."""
        X = self._check_predict(X)
        return self.estimator_.predictive_gradient_gram(X)

    def save(self, model_path):
        """Writes the model to ``model_path`` (+ '.pickle', the reference's suffix rule,
        base.py:237-239).  The file is an .npz archive of arrays, not a Python pickle."""
        if not model_path.endswith(".pickle"):
            model_path += ".pickle"
        self.estimator_.pickle(model_path)

    def load(self, model_path):
        """Restores ``estimator_`` (and, unlike the reference, ``n_features_`` too, so
        ``predict`` works on a fresh object)."""
        if not model_path.endswith(".pickle"):
            model_path += ".pickle"
        self.estimator_ = load_model(model_path)
        self.n_features_ = self.estimator_.input_dim
