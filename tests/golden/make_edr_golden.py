"""Generates tests/golden/edr_golden.npz from the REFERENCE's own EDR layer
(edrgp.edr / edrgp.base / edrgp.utils, importable in the build container; its
gp_model is not, GPy being absent).  A deterministic duck-typed estimator
(tests/duck_estimator.py) supplies the gradients, so the fixtures pin the EDR
algebra: components, subspace variances, projections, SVDTransformer,
subspace_variance_ratio, discrepancy, ort_space.  Inputs and expected outputs
only.  Run from the repo root:  python tests/golden/make_edr_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
from duck_estimator import TanhRidgeEstimator  # noqa: E402
from edrgp.base import BaseEDR, BlockEDR  # noqa: E402   (reference)
from edrgp.edr import EffectiveDimensionalityReduction  # noqa: E402   (reference)
from edrgp.utils import SVDTransformer, discrepancy, ort_space, subspace_variance_ratio  # noqa: E402
import numpy.matlib  # noqa: E402,F401  (the reference's BlockEDR relies on it being imported)
from sklearn.decomposition import PCA  # noqa: E402

out = {}
rng = np.random.default_rng(42)
N, D = 400, 6
A = rng.standard_normal((D, D))
X = rng.standard_normal((N, D)) @ A * 0.5 + np.array([3.0, -2.0, 0.0, 10.0, 0.5, -7.0])
w = np.array([0.6, -0.4, 0.0, 0.05, 0.0, 0.02])
y = np.tanh((X - X.mean(0)) @ w) + 0.01 * rng.standard_normal(N)
out["X"], out["y"] = X, y

configs = {
    "oneshot": dict(n_components=2, step=None, normalize=True, preprocessor=None),
    "nonorm": dict(n_components=2, step=None, normalize=False, preprocessor=None),
    "iter_int": dict(n_components=2, step=2, normalize=True, preprocessor=None),
    "iter_float": dict(n_components=None, step=0.9, normalize=True, preprocessor=None),
    "pca_pre": dict(n_components=1, step=None, normalize=True, preprocessor=PCA(n_components=4)),
}
for name, cfg in configs.items():
    edr = EffectiveDimensionalityReduction(TanhRidgeEstimator(), SVDTransformer(), cfg["n_components"], cfg["step"],
                                           cfg["normalize"], cfg["preprocessor"])
    edr.fit(X, y)
    out[name + "_components"] = edr.components_
    out[name + "_subspace_variance"] = np.atleast_1d(edr.subspace_variance_)
    out[name + "_subspace_variance_ratio"] = np.atleast_1d(edr.subspace_variance_ratio_)
    out[name + "_transform"] = edr.transform(X)
    out[name + "_num_iter"] = np.array(edr.num_iter)
    out[name + "_feature_importances"] = edr.feature_importances_
    out[name + "_gradients"] = edr.get_estimator_gradients(X[:10])

base = BaseEDR(TanhRidgeEstimator(), SVDTransformer(), n_components=3)
base.fit(X, y)
out["base_components"] = base.components_
out["base_inverse"] = base.inverse_transform(base.transform(X[:5]))
base.refit(PCA(n_components=2))
out["base_refit_components"] = base.refit_components_
out["base_refit_ratio"] = np.atleast_1d(base.refit_subspace_variance_ratio_)

blk = BlockEDR(TanhRidgeEstimator(), SVDTransformer(), n_components=[1, 2], blocks=[[0, 1, 2], [3, 4, 5]])
blk.fit(X, y)
out["block_components"] = blk.components_
out["block_transform"] = blk.transform(X)

G = rng.standard_normal((50, 5))
svd = SVDTransformer(n_components=3).fit(G)
out["svd_in"], out["svd_components"] = G, svd.components_
out["svd_variance"], out["svd_ratio"] = svd.subspace_variance_, svd.subspace_variance_ratio_
V = np.linalg.qr(rng.standard_normal((5, 2)))[0]
sv, svr = subspace_variance_ratio(G, V)
out["svr_V"], out["svr_variance"], out["svr_ratio"] = V, sv, svr
Vn = rng.standard_normal((5, 2))
sv2, svr2 = subspace_variance_ratio(G, Vn)
out["svr_Vn"], out["svr_variance_n"], out["svr_ratio_n"] = Vn, np.atleast_1d(sv2), np.atleast_1d(svr2)
B = np.linalg.qr(rng.standard_normal((5, 2)))[0]
out["disc_B"], out["disc"] = B, np.array(discrepancy(B, V))
out["ort"] = ort_space(V)

path = os.path.join(HERE, "edr_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
