"""The reference's own tests (edrgp/tests/test_edr.py:33-117, plus the ones it
ships commented out at :120-222 that still make sense), run end to end on the
MI355X estimators: GP fit (L-BFGS-B over HIP evaluations) -> predict_gradient ->
EDR.  Same data generators, same assertions and tolerances."""
from copy import deepcopy

import numpy as np
import pytest
from sklearn.decomposition import PCA
from sklearn.feature_selection import mutual_info_regression
from sklearn.preprocessing import StandardScaler

from edrgp_amd import kern
from edrgp_amd.base import BlockEDR
from edrgp_amd.datasets import get_beta_inputs, get_edr_target, get_gaussian_inputs, get_tanh_targets
from edrgp_amd.edr import EffectiveDimensionalityReduction
from edrgp_amd.gp_model import GaussianProcessRegressor, SparseGaussianProcessRegressor
from edrgp_amd.utils import SVDTransformer, discrepancy

pytestmark = pytest.mark.gpu


def get_2d_data(mean=None):
    if mean is None:
        mean = [0, 0]
    X = get_gaussian_inputs(eig_values=[1, 0.3], sample_size=500, eig_vectors=np.array([[1, 1], [-1, 1]]), mean=mean)
    y = get_tanh_targets(X, [0.5, 0.5])
    return X, y


def test_sparse_regression():
    np.random.seed(101)
    N = 50
    noise_var = 0.05
    X = np.linspace(0, 10, 50)[:, None]
    k = kern.RBF(1)
    y = np.random.multivariate_normal(np.zeros(N), k.K(X) + np.eye(N) * np.sqrt(noise_var)).reshape(-1, 1)
    gp = GaussianProcessRegressor()
    gp.fit(X, y)
    sgp = SparseGaussianProcessRegressor(num_inducing=12)
    sgp.fit(X, y)
    assert abs(gp.estimator_.log_likelihood() - sgp.estimator_.log_likelihood()[0][0]) < 0.5


@pytest.mark.parametrize("mean", [[0, 0], [10, -10]])
def test_mi(mean):
    np.random.seed(11)
    X, y = get_2d_data(mean)
    edr = EffectiveDimensionalityReduction(GaussianProcessRegressor(), SVDTransformer(), n_components=1,
                                           normalize=True)
    edr.fit(X, y)
    mi = mutual_info_regression(edr.transform(X), y)[0]
    assert mi > 1


@pytest.mark.parametrize("normalize", [True, False])
def test_translation(normalize):
    np.random.seed(12)
    X, y = get_2d_data(mean=[10, -10])
    edr = EffectiveDimensionalityReduction(GaussianProcessRegressor(), SVDTransformer(), n_components=1,
                                           normalize=normalize)
    edr.fit(X, y)
    components_shift = edr.components_
    X -= X.mean(0)
    edr2 = deepcopy(edr)
    edr2.fit(X, y)
    assert np.allclose(components_shift, edr2.components_, rtol=1e-3)


@pytest.mark.parametrize("mean", [[0, 0, 0, 0], [10, -10, 100, -100]])
def test_preprocess(mean):
    np.random.seed(13)
    X = get_gaussian_inputs(eig_values=[1, 0.3, 0.001, 0.001], sample_size=500, mean=mean)
    y = get_tanh_targets(X, [0.5, 0.5, 0, 0])
    make = lambda: EffectiveDimensionalityReduction(GaussianProcessRegressor(), SVDTransformer(), n_components=1,  # noqa: E731
                                                    normalize=True, preprocessor=PCA(n_components=2))
    edr = make().fit(X, y)
    components_shift = edr.components_
    X -= X.mean(0)
    edr = make().fit(X, y)
    assert np.allclose(components_shift, edr.components_, rtol=1e-3)


@pytest.mark.parametrize("mean", [[0, 0], [10, -10]])
def test_scaling(mean):
    np.random.seed(14)
    X, y = get_2d_data(mean)
    edr_sc = EffectiveDimensionalityReduction(GaussianProcessRegressor(), SVDTransformer(), normalize=True)
    edr_sc.fit(X, y)
    x1 = edr_sc.transform(X - np.mean(X, axis=0))
    edr = EffectiveDimensionalityReduction(GaussianProcessRegressor(), SVDTransformer(), normalize=False)
    X_scaled = StandardScaler().fit_transform(X)
    x2 = edr.fit_transform(X_scaled, y)
    assert np.allclose(x1, x2)


@pytest.mark.parametrize("n_components,step", [(3, 1), (None, 0.99)])
def test_iterative(n_components, step):
    """edrgp/tests/test_edr.py:120-133 (shipped commented out upstream), with upstream's own set-up and
    assertions: beta inputs 300 x 10, B = qr(random_sparse(10, 3, density 0.3, random_state 0)),
    y = get_edr_target(X B, 0.1), ARD RBF GP, SVDTransformer, normalize=False; the iterative EDR must end
    with exactly 3 directions, discrepancy < 0.1, variance ratios sorted.

    Round 1 ran a variant (PCA() + normalize=True) and relaxed the assertions when it kept all 10 directions.
    Diagnosis (round 2, CPU, the reference's own EDR classes imported from /root/reference and driven by the
    oracle-backed GP on identical data): with upstream's set-up the reference and this repository's EDR layer
    give the same result -- 3 x 10 components, discrepancy 0.026 (step 1) / 0.034 (step 0.99), 7 / 3 iterations
    -- while with PCA() + normalize=True BOTH keep 10 directions at step 0.99: PCA centres the gradient matrix,
    so the cumulative-variance rule of _select_n_components (edrgp/base.py:498-509) sees the variance about the
    mean gradient and never reaches 0.99 before the last direction.  It was the test that deviated, not
    `_select_n_components`; the assertions are upstream's again."""
    from scipy.sparse import random as random_sparse
    np.random.seed(0)
    X = get_beta_inputs(300, 10)
    B = np.linalg.qr(random_sparse(10, 3, density=0.3, random_state=0).toarray())[0]
    y = get_edr_target(X.dot(B), 0.1)
    gp_model = GaussianProcessRegressor(['RBF'], [{'ARD': True}])
    edr = EffectiveDimensionalityReduction(gp_model, SVDTransformer(), n_components=n_components, step=step,
                                           normalize=False)
    edr.fit(X, y)
    assert discrepancy(B, edr.components_.T) < 1e-1
    assert edr.components_.shape[0] == 3
    assert np.all(edr.subspace_variance_ratio_ == np.sort(edr.subspace_variance_ratio_)[::-1])


def test_get_gradients_shapes_and_sparse_estimator():
    """predict_gradient / get_estimator_gradients shapes with the sparse GP inside EDR."""
    np.random.seed(2)
    X = get_beta_inputs(300, 6)
    y = np.tanh(X[:, 0] - 0.5 * X[:, 1]) + 0.05 * np.random.randn(300)
    edr = EffectiveDimensionalityReduction(SparseGaussianProcessRegressor(num_inducing=40), SVDTransformer(),
                                           n_components=2, normalize=True)
    edr.fit(X, y, max_iters=200)
    assert edr.components_.shape == (2, 6)
    assert edr.get_estimator_gradients(X).shape == (300, 6)
    assert edr.transform(X).shape == (300, 2)
    lead = np.abs(edr.components_[0]) / np.linalg.norm(edr.components_[0])
    assert lead[:2].sum() > lead[2:].sum()


def test_block_edr_zero_structure():
    """BlockEDR (edrgp/tests/test_edr.py:178-196, disabled upstream): components are block diagonal."""
    np.random.seed(3)
    X = get_beta_inputs(300, 6)
    y = np.tanh(X[:, 0] + X[:, 1]) + X[:, 4] ** 2 + 0.05 * np.random.randn(300)
    blocks = [[0, 1, 2], [3, 4, 5]]
    edr = BlockEDR(GaussianProcessRegressor(), SVDTransformer(), n_components=[1, 1], blocks=blocks)
    edr.fit(X, y, max_iters=200)
    assert edr.components_.shape == (2, 6)
    assert np.all(edr.components_[0, 3:] == 0) and np.all(edr.components_[1, :3] == 0)


def test_large_sparse_fit_end_to_end():
    """The estimator API at a size GPy's CPU path would take minutes on: real L-BFGS-B fit
    (hyper-parameters and inducing inputs), prediction quality, gradient direction."""
    rng = np.random.default_rng(0)
    N, D, M = 20000, 8, 256
    X = rng.standard_normal((N, D))
    w = np.zeros(D)
    w[:2] = [0.8, -0.6]
    y = np.tanh(X @ w) + 0.05 * rng.standard_normal(N)
    np.random.seed(0)
    est = SparseGaussianProcessRegressor(kernels="RBF", kernel_options={"ARD": True}, num_inducing=M)
    est.fit(X, y, max_iters=40)
    Xt = rng.standard_normal((2000, D))
    yt = np.tanh(Xt @ w)
    pred = est.predict(Xt)
    assert pred.shape == (2000,)
    assert np.mean((pred - yt) ** 2) < 0.02 * np.var(yt) + 1e-3
    assert est.predict_variance(Xt[:10]).shape == (10, 1)
    grad = est.predict_gradient(Xt[:500])
    true_grad = (1 - np.tanh(Xt[:500] @ w) ** 2)[:, None] * w[None, :]
    cos = np.sum(grad * true_grad, 1) / (np.linalg.norm(grad, axis=1) * np.linalg.norm(true_grad, axis=1) + 1e-12)
    assert np.median(cos) > 0.95
    ll = est.estimator_.log_likelihood()
    assert ll.shape == (1, 1) and np.isfinite(ll[0, 0])
    # ARD found the two active inputs: their lengthscales are the shortest
    ls = est.estimator_.kern.parts[0].lengthscale
    assert set(np.argsort(ls)[:2]) == {0, 1}
