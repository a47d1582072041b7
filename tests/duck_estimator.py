"""Deterministic sklearn-clonable stand-in for a GP estimator (tests only): fits
y ~ tanh(X w) by least squares on atanh(y) and returns the analytic gradient.
Lets the EDR layer be compared with the reference's EDR layer on identical
gradient matrices, with no GP in the loop."""
import numpy as np
from sklearn.base import BaseEstimator


class TanhRidgeEstimator(BaseEstimator):
    def __init__(self, ridge=1e-3):
        self.ridge = ridge

    def fit(self, X, y, **kw):
        X = np.asarray(X, dtype=float)
        t = np.arctanh(np.clip(np.asarray(y, dtype=float).ravel(), -0.98, 0.98))
        A = X.T @ X + self.ridge * np.eye(X.shape[1])
        self.w_ = np.linalg.solve(A, X.T @ t)
        return self

    def predict(self, X):
        return np.tanh(np.asarray(X) @ self.w_)

    def predict_gradient(self, X):
        s = 1.0 - np.tanh(np.asarray(X) @ self.w_) ** 2
        return s[:, None] * self.w_[None, :]
