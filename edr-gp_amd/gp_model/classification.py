"""GP classifiers with the reference's constructor signatures (edrgp/gp_model/classification.py:38-45, :135-144).

``GaussianProcessClassifier`` stands on the dense EP model (``edrgp_amd.models.GPClassificationModel``: Bernoulli
likelihood, probit link, expectation propagation on the device), ``SparseGaussianProcessClassifier`` on
``SparseGPClassificationModel`` (GPy's ``SparseGPClassification``: EP on the DTC approximation, EPDTC).
"""
from __future__ import annotations

from sklearn.base import ClassifierMixin

from ..models import GPClassificationModel, SparseGPClassificationModel
from .base import _BaseGP


class GaussianProcessClassifier(ClassifierMixin, _BaseGP):
    """Parameters: kernels, kernel_options (as the regressors), Y_metadata / mean_function (must be None),
    method ('optimize' | 'optimize_restarts').  Attributes: ``estimator_``, ``n_features_``.

    ``predict`` returns the class (posterior probability of the second class > 0.5), ``predict_proba`` that
    probability, ``predict_gradient`` the gradient of the latent posterior mean (what the EDR estimators consume)."""

    def __init__(self, kernels=None, kernel_options=None, Y_metadata=None, mean_function=None, method="optimize"):
        self.kernels = kernels
        self.kernel_options = kernel_options
        self.Y_metadata = Y_metadata
        self.mean_function = mean_function
        self.method = method

    def _get_model(self, X, y, kernel):
        return GPClassificationModel(X, y, kernel, self.Y_metadata, self.mean_function)

    def predict(self, X):
        return super().predict(X) > 0.5

    def predict_proba(self, X):
        return super().predict(X)


class SparseGaussianProcessClassifier(ClassifierMixin, _BaseGP):
    """Parameters: kernels, kernel_options (as the regressors), likelihood (must be None: Bernoulli with the probit
    link, GPy's default), Z / num_inducing (inducing inputs, or how many training rows to draw for them; default 10),
    Y_metadata (must be None), method ('optimize' | 'optimize_restarts').  Attributes: ``estimator_``, ``n_features_``.

    Stands on ``edrgp_amd.models.SparseGPClassificationModel`` (EPDTC on the device).  ``predict`` returns the class,
    ``predict_proba`` the posterior probability of the second class, ``predict_gradient`` the gradient of the latent
    posterior mean."""

    def __init__(self, kernels=None, kernel_options=None, likelihood=None, Z=None, num_inducing=10, Y_metadata=None,
                 method="optimize"):
        self.kernels = kernels
        self.kernel_options = kernel_options
        self.likelihood = likelihood
        self.Z = Z
        self.num_inducing = num_inducing
        self.Y_metadata = Y_metadata
        self.method = method

    def _get_model(self, X, y, kernel):
        return SparseGPClassificationModel(X, y, likelihood=self.likelihood, kernel=kernel, Z=self.Z,
                                           num_inducing=self.num_inducing, Y_metadata=self.Y_metadata)

    def predict(self, X):
        return super().predict(X) > 0.5

    def predict_proba(self, X):
        return super().predict(X)
