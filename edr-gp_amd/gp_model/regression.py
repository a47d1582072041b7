"""GP regressors with the reference's constructor signatures
(edrgp/gp_model/regression.py:47-49 and :122-124)."""
from __future__ import annotations

from sklearn.base import RegressorMixin

from ..models import GPRegressionModel, SparseGPRegressionModel
from .base import _BaseGP


class GaussianProcessRegressor(RegressorMixin, _BaseGP):
    """Exact GP regressor (dense Cholesky on the device).

    Parameters
    ----------
    kernels : None, str, list of str or kernel object
        ``None`` -> RBF; names from {"RBF", "Matern32", "Matern52"}; a list is summed.
    kernel_options : dict or list of dict
        ``variance``, ``lengthscale``, ``ARD`` per kernel.
    Y_metadata, mean_function : must be None (GPy-specific).
    normalizer : True -> standardise y (mean / std, ddof 0); False / None -> no.
    noise_var : initial Gaussian noise variance (default 1).
    method : 'optimize' or 'optimize_restarts'.

    Attributes: ``estimator_`` (fitted model), ``n_features_``.
    """

    def __init__(self, kernels=None, kernel_options=None, Y_metadata=None, normalizer=True, noise_var=1.0,
                 mean_function=None, method="optimize"):
        self.normalizer = normalizer
        self.noise_var = noise_var
        self.kernels = kernels
        self.kernel_options = kernel_options
        self.Y_metadata = Y_metadata
        self.mean_function = mean_function
        self.method = method

    def _get_model(self, X, y, kernel):
        return GPRegressionModel(X, y, kernel, self.Y_metadata, self.normalizer, self.noise_var, self.mean_function)


class SparseGaussianProcessRegressor(RegressorMixin, _BaseGP):
    """Inducing-point (VarDTC / Titsias) GP regressor -- the MI355X hot path.

    Parameters as above, plus
    Z : array (num_inducing, n_features) or None -- inducing inputs; ``None`` picks
        ``num_inducing`` random training rows (global numpy RNG, as GPy does).
    num_inducing : int, default 10 (ignored when Z is given).
    X_variance : must be None.
    precision : 'fp64' (default) or 'fp32' -- an addition to the reference signature.  'fp32' runs the
        gradient contraction of the native path on the fp32 MFMA (BASELINE config 5); the bound stays fp64.
    """

    def __init__(self, kernels=None, kernel_options=None, Z=None, num_inducing=10, Y_metadata=None,
                 X_variance=None, normalizer=True, mean_function=None, method="optimize", precision="fp64"):
        self.kernels = kernels
        self.kernel_options = kernel_options
        self.Z = Z
        self.num_inducing = num_inducing
        self.Y_metadata = Y_metadata
        self.X_variance = X_variance
        self.normalizer = normalizer
        self.mean_function = mean_function
        self.method = method
        self.precision = precision

    def _get_model(self, X, y, kernel):
        if self.precision not in ("fp64", "fp32"):
            raise ValueError("precision must be 'fp64' or 'fp32'")
        if self.Y_metadata is not None:
            raise NotImplementedError("Y_metadata is GPy-specific and not supported")
        return SparseGPRegressionModel(X, y, kernel=kernel, Z=self.Z, num_inducing=self.num_inducing,
                                       X_variance=self.X_variance, mean_function=self.mean_function,
                                       normalizer=self.normalizer, precision=32 if self.precision == "fp32" else 64)
