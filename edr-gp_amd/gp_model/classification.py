"""GP classifiers with the reference's constructor signatures (edrgp/gp_model/classification.py:38-45, :135-144).

``GaussianProcessClassifier`` stands on the dense EP model (``edrgp_amd.models.GPClassificationModel``: Bernoulli
likelihood, probit link, expectation propagation on the device).  ``SparseGaussianProcessClassifier`` keeps the
reference's interface but is not built: GPy's ``SparseGPClassification`` runs EP on the DTC approximation (EPDTC), a
second inference algorithm outside the scope table's last row.
"""
from __future__ import annotations

from sklearn.base import ClassifierMixin

from ..models import GPClassificationModel
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
    """Interface of the reference's sparse classifier; fitting raises ``NotImplementedError`` (EPDTC is not built)."""

    def __init__(self, kernels=None, kernel_options=None, likelihood=None, Z=None, num_inducing=10, Y_metadata=None,
                 method="optimize"):
        self.kernels = kernels
        self.kernel_options = kernel_options
        self.likelihood = likelihood
        self.Z = Z
        self.num_inducing = num_inducing
        self.Y_metadata = Y_metadata
        self.method = method

    _MESSAGE = ("SparseGPClassification (EP on the DTC approximation) is not built on the MI355X path; "
                "use GaussianProcessClassifier (dense EP) or a sparse regressor")

    def fit(self, X, y, **opt_kws):
        # refused before any validation or device work, so that clone()d copies inside EDR pipelines fail at once
        raise NotImplementedError(self._MESSAGE)

    def _get_model(self, X, y, kernel):
        raise NotImplementedError(self._MESSAGE)

    def predict(self, X):
        return super().predict(X) > 0.5

    def predict_proba(self, X):
        return super().predict(X)
