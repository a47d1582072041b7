"""CPU tests of the host side of the product (no GPU): Logexp, parameter
packing in GPy's order, the transformed objective gradient, the L-BFGS-B driver,
normaliser handling, estimator API and errors.  Device calls are answered by a
test-only evaluator built on the oracle (tests/oracle_evaluator.py)."""
import numpy as np
import pytest
from sklearn.base import clone

from oracle import gp_oracle as go
from edrgp_amd import kern as K
from edrgp_amd import models
from edrgp_amd.gp_model import GaussianProcessRegressor, SparseGaussianProcessRegressor
from oracle_evaluator import OracleEvaluator


def test_logexp_matches_paramz_restatement():
    x = np.array([-50.0, -3.0, 0.0, 0.5, 10.0, 35.9, 36.1, 100.0])
    assert np.allclose(models.logexp_f(x), go.Logexp.f(x), rtol=1e-15, atol=0)
    f = np.array([1e-6, 0.3, 1.0, 5.0, 40.0])
    assert np.allclose(models.logexp_finv(f), go.Logexp.finv(f), rtol=1e-15)
    assert np.allclose(models.logexp_f(models.logexp_finv(f)), f, rtol=1e-12)
    assert np.allclose(models.logexp_gradfactor(f, 2.0), go.Logexp.gradfactor(f, 2.0))


def test_kernel_constructor_rules():
    k = K.RBF(3)
    assert k.variance == 1.0 and k.lengthscale.tolist() == [1.0] and not k.ARD
    k = K.Matern52(3, variance=2.0, lengthscale=0.5, ARD=True)
    assert k.lengthscale.tolist() == [0.5, 0.5, 0.5]
    with pytest.raises(AssertionError):
        K.RBF(3, lengthscale=[1.0, 2.0])
    s = np.sum([K.RBF(2), K.Matern32(2)])
    assert [p.kind for p in s.parts] == ["RBF", "Matern32"]
    assert K.from_names(None, None, 4).kind == "RBF"
    assert [p.kind for p in K.from_names(["RBF", "Matern52"], None, 4).parts] == ["RBF", "Matern52"]
    with pytest.raises(ValueError):
        K.from_names(["RBF", "Matern52"], [{"variance": 1.0}], 4)
    with pytest.raises(AttributeError):
        K.from_names("Linear", None, 4)


def _problem(N=60, D=3, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.tanh(X @ np.array([0.8, -0.5, 0.1])[:D]) + 0.05 * rng.standard_normal(N)
    return X, y


@pytest.mark.parametrize("sparse", [False, True])
def test_objective_gradient_in_optimizer_space(sparse):
    X, y = _problem()
    kern = K.RBF(3, variance=0.7, lengthscale=[1.0, 1.5, 2.0], ARD=True) + K.Matern32(3, variance=0.2, lengthscale=0.9)
    if sparse:
        np.random.seed(1)
        m = models.SparseGPRegressionModel(X, y, kern, num_inducing=9, normalizer=True, evaluator=OracleEvaluator())
        assert m.parameter_names()[:27] == ["inducing_inputs"] * 27
    else:
        m = models.GPRegressionModel(X, y, kern, normalizer=True, noise_var=0.3, evaluator=OracleEvaluator())
    assert m.parameter_names()[-7:] == ["sum.rbf.variance"] + ["sum.rbf.lengthscale"] * 3 + \
        ["sum.matern32.variance", "sum.matern32.lengthscale", "Gaussian_noise.variance"]
    x0 = m.optimizer_array.copy()
    f0, g0 = m._objective_grads(x0)
    assert np.isclose(f0, m.objective_function())
    for i in list(range(min(4, x0.size))) + list(range(x0.size - 7, x0.size)):
        h = 1e-6
        xp, xm = x0.copy(), x0.copy()
        xp[i] += h
        xm[i] -= h
        fd = (m._objective_grads(xp)[0] - m._objective_grads(xm)[0]) / (2 * h)
        assert abs(fd - g0[i]) <= 1e-5 * max(1.0, abs(g0[i])), (i, fd, g0[i])
    m.optimizer_array = x0
    assert np.allclose(m.optimizer_array, x0)


def test_reference_sparse_vs_dense_likelihood():
    """edrgp/tests/test_edr.py:33-50 (test_sparse_regression) on the host logic:
    optimised dense and sparse (12 inducing points) likelihoods agree within 0.5."""
    np.random.seed(101)
    N, noise_var = 50, 0.05
    X = np.linspace(0, 10, 50)[:, None]
    Kx = go.Stationary("RBF", 1).K(X)
    y = np.random.multivariate_normal(np.zeros(N), Kx + np.eye(N) * np.sqrt(noise_var)).reshape(-1, 1)
    gp = models.GPRegressionModel(X, y, normalizer=True, evaluator=OracleEvaluator())
    gp.optimize(messages=False, max_iters=1000)
    sgp = models.SparseGPRegressionModel(X, y, num_inducing=12, normalizer=True, evaluator=OracleEvaluator())
    sgp.optimize(messages=False, max_iters=1000)
    ll_d, ll_s = gp.log_likelihood(), sgp.log_likelihood()
    assert isinstance(ll_d, float) and ll_s.shape == (1, 1)
    assert abs(ll_d - ll_s[0][0]) < 0.5
    assert ll_s[0][0] <= ll_d + 1e-6            # variational lower bound


def test_predict_unnormalises_and_shapes():
    X, y = _problem(N=40)
    y = 5.0 + 3.0 * y
    m = models.GPRegressionModel(X, y, normalizer=True, noise_var=0.1, evaluator=OracleEvaluator())
    mu, var = m.predict(X[:7])
    assert mu.shape == (7, 1) and var.shape == (7, 1)
    k = go.Stationary("RBF", 3)
    norm = go.Standardize()
    norm.scale_by(y[:, None])
    inf = go.exact_inference(k, X, 0.1, norm.normalize(y[:, None]))
    mu_ref, var_ref = go.predict(k, X[:7], X, inf["woodbury_vector"], inf["woodbury_inv"], 0.1, norm)
    assert np.allclose(mu, mu_ref) and np.allclose(var, var_ref)
    jac, none = m.predictive_gradients(X[:7])
    assert jac.shape == (7, 3, 1) and none is None
    assert np.allclose(jac, go.predictive_gradients_mean(k, X[:7], X, inf["woodbury_vector"], norm))
    m2 = models.GPRegressionModel(X, y, normalizer=False, noise_var=0.1, evaluator=OracleEvaluator())
    assert m2.normalizer is None and np.array_equal(m2.Y_normalized[:, 0], y)


def test_optimize_restarts_keeps_best():
    X, y = _problem(N=30)
    m = models.GPRegressionModel(X, y, normalizer=True, evaluator=OracleEvaluator())
    np.random.seed(3)
    runs = m.optimize_restarts(num_restarts=3, verbose=False, max_iters=50)
    assert len(runs) == 3
    best = min(r.f_opt for r in runs)
    assert np.isclose(m.objective_function(), best, rtol=1e-6)


def test_estimator_api_and_errors(monkeypatch):
    monkeypatch.setattr(models, "HipEvaluator", lambda device=None: OracleEvaluator())
    X, y = _problem(N=40)
    est = SparseGaussianProcessRegressor(kernels="Matern52", kernel_options={"ARD": True}, num_inducing=8)
    assert clone(est).get_params()["num_inducing"] == 8
    np.random.seed(0)
    est.fit(X, y, max_iters=30)
    assert est.n_features_ == 3 and est.kernels == ["Matern52"] and est.kernel_options == [{"ARD": True}]
    assert est.predict(X[:5]).shape == (5,)
    assert est.predict_variance(X[:5]).shape == (5, 1)
    assert est.predict_gradient(X[:5]).shape == (5, 3)
    with pytest.raises(ValueError, match="X has 2 features per sample; expecting 3"):
        est.predict(X[:5, :2])
    with pytest.raises(ValueError):
        GaussianProcessRegressor(kernels=["RBF", "Matern32"], kernel_options=[{}]).fit(X, y)
    with pytest.raises(NotImplementedError):
        GaussianProcessRegressor(Y_metadata={"a": 1}).fit(X, y)
    with pytest.raises(NotImplementedError):
        SparseGaussianProcessRegressor(X_variance=np.ones_like(X)).fit(X, y)
    with pytest.raises(ValueError):
        est.fit(X, np.full(y.shape, np.nan))


def test_save_load_roundtrip(tmp_path, monkeypatch):
    monkeypatch.setattr(models, "HipEvaluator", lambda device=None: OracleEvaluator())
    X, y = _problem(N=40)
    np.random.seed(0)
    est = SparseGaussianProcessRegressor(num_inducing=6).fit(X, y, max_iters=20)
    path = str(tmp_path / "model")
    est.save(path)
    fresh = SparseGaussianProcessRegressor()
    fresh.load(path)
    assert fresh.n_features_ == 3
    assert np.allclose(fresh.predict(X[:5]), est.predict(X[:5]))
    assert np.allclose(fresh.predict_gradient(X[:5]), est.predict_gradient(X[:5]))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports every symbol of include/edrgp_hip.h."""
    import re, os
    from edrgp_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "edrgp_hip.h")).read()
    declared = set(re.findall(r"\b(edrgp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)


def test_product_fails_loudly_without_gpu():
    from edrgp_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from edrgp_amd.backend import HipGP
    with pytest.raises(_lib.EdrgpError):
        HipGP(0)


def test_svd_transformer_thin_svd_equals_full_and_scales():
    """SVDTransformer uses the thin SVD (the reference's full_matrices=True builds an n x n U): same
    singular values / right singular vectors on a small matrix, and a 200 000-row gradient matrix is
    decomposed in well under a second instead of needing 320 GB."""
    import time
    from edrgp_amd.utils import SVDTransformer
    rng = np.random.default_rng(0)
    A = rng.standard_normal((300, 7)) * np.arange(1, 8)
    t = SVDTransformer().fit(A)
    _, S, Vh = np.linalg.svd(A)                      # the reference's call (edrgp/utils.py:140)
    assert np.allclose(t.subspace_variance_, S ** 2, rtol=1e-12)
    assert np.allclose(np.abs(np.sum(t.components_ * Vh, axis=1)), 1.0, atol=1e-12)
    G = rng.standard_normal((200000, 16))
    t0 = time.perf_counter()
    SVDTransformer(n_components=3).fit(G)
    assert time.perf_counter() - t0 < 5.0


@pytest.mark.parametrize("kind", ["RBF", "Matern32", "Matern52"])
def test_oracle_gradients_x_gemm_form(kind):
    """The oracle's reordered gradients_X (one BLAS product instead of input_dim passes; used by the full-size
    parity tests only) equals GPy's loop form, both with X2 given and with X2 = None, ARD or not."""
    from oracle import gp_oracle as go
    rng = np.random.default_rng(3)
    for ard in (False, True):
        D = 5
        ls = rng.uniform(0.7, 1.9, D) if ard else 1.3
        k = go.Stationary(kind, D, variance=1.7, lengthscale=ls, ARD=ard)
        X, X2 = rng.standard_normal((40, D)), rng.standard_normal((23, D))
        X2[3] = X[7]                                    # a zero distance: the 1/r convention must agree
        dL = rng.standard_normal((40, 23))
        a = k.gradients_X(dL, X, X2)
        b = k.gradients_X(dL, X, X2, gemm_form=True)
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())
        dLs = rng.standard_normal((40, 40))
        a = k.gradients_X(dLs, X)
        b = k.gradients_X(dLs, X, gemm_form=True)
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())


def test_fit_gram_falls_back_on_rank_deficient_gradients():
    """SVDTransformer.fit_gram works on G^T G, which squares the condition number: a numerically rank-deficient
    gradient matrix (singular values below 1e-6 of the largest) goes back to the dense SVD when the matrix can be
    fetched, and then reproduces ``fit`` exactly; a full-rank one stays on the Gram path."""
    from edrgp_amd.utils import GradientGram, SVDTransformer
    rng = np.random.default_rng(11)
    B = rng.standard_normal((400, 3))
    G = B @ rng.standard_normal((3, 6)) + 1e-9 * rng.standard_normal((400, 6))      # rank 3 + noise at 1e-9
    fetched = []
    gram = GradientGram(G.T @ G, 400, G.sum(0), fetch=lambda: fetched.append(1) or G)
    a = SVDTransformer(n_components=0.999999999999).fit_gram(gram)
    b = SVDTransformer(n_components=0.999999999999).fit(G)
    assert fetched, "the dense matrix was not fetched"
    assert a.components_.shape == b.components_.shape
    assert np.allclose(a.components_, b.components_) and np.allclose(a.subspace_variance_, b.subspace_variance_)
    # full rank: no fetch, same leading subspace
    G2 = rng.standard_normal((400, 6))
    fetched.clear()
    a2 = SVDTransformer(n_components=2).fit_gram(GradientGram(G2.T @ G2, 400, G2.sum(0), fetch=lambda: fetched.append(1) or G2))
    b2 = SVDTransformer(n_components=2).fit(G2)
    assert not fetched
    assert np.allclose(np.abs(a2.components_ @ b2.components_.T), np.eye(2), atol=1e-8)


def test_sparse_classifier_is_clonable_and_refuses_other_likelihoods():
    from edrgp_amd.gp_model import SparseGaussianProcessClassifier
    est = clone(SparseGaussianProcessClassifier(num_inducing=5, likelihood="Gaussian"))      # constructible and clonable
    assert est.num_inducing == 5
    with pytest.raises(NotImplementedError):                 # only GPy's default (Bernoulli, probit) is built
        est.fit(np.random.default_rng(0).standard_normal((8, 2)), np.array([0, 1, 0, 1, 1, 0, 1, 0]))
