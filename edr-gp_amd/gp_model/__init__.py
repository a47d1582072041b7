"""sklearn-style GP estimators: same names and signatures as ``edrgp.gp_model``
(edrgp/gp_model/__init__.py:1-9), computed on MI355X."""
from .classification import GaussianProcessClassifier, SparseGaussianProcessClassifier
from .regression import GaussianProcessRegressor, SparseGaussianProcessRegressor

__all__ = ["GaussianProcessRegressor", "SparseGaussianProcessRegressor", "GaussianProcessClassifier",
           "SparseGaussianProcessClassifier"]
