"""GP classification (edrgp/gp_model/classification.py): Bernoulli likelihood, probit link, expectation
propagation.  CPU: the oracle's EP restatement satisfies EP's own fixed-point conditions and the variational sanity
checks available without GPy (parity with GPy itself is unpinned, as for the rest of the path); the host model /
estimator logic runs on the oracle evaluator.  GPU: the device's sites, marginal likelihood, gradients and predictions
against the oracle for the SAME site order, and the estimators end to end."""
import numpy as np
import pytest

from oracle import gp_oracle as go
from oracle_evaluator import OracleEvaluator


def _data(N, D, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    f = np.sin(2.0 * X[:, 0]) + 0.7 * X[:, 1 % D]
    y = (f + 0.3 * rng.standard_normal(N) > 0).astype(float)
    return X, y


class _Orders:
    """The same sequence of site permutations for the oracle and the device."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.used = []

    def __call__(self, n):
        o = self.rng.permutation(n)
        self.used.append(o)
        return o


def test_oracle_ep_fixed_point_and_gradients():
    X, y = _data(200, 2, 0)
    kern = go.make_kernel(["RBF"], 2, [dict(variance=1.3, lengthscale=0.9)])
    sites = go.expectation_propagation(kern.K(X), y, order_fn=_Orders(1))
    assert 2 <= sites["sweeps"] <= 30
    inf = go.ep_inference(kern, X, sites)
    # EP fixed point: the Gaussian marginals match the moments of the tilted distributions
    assert np.abs(inf["mu"] - sites["mu_hat"]).max() < 5e-3
    assert np.abs(inf["Sigma_diag"] - sites["sigma2_hat"]).max() < 5e-3
    assert np.all(sites["tau"] > 0)
    # Z_EP is below the number of label configurations' trivial bound and finite
    assert np.isfinite(inf["log_marginal"]) and inf["log_marginal"] < 0
    # gradient of the fixed-site marginal w.r.t. the kernel parameters: central differences
    g = kern.param_gradients_full(inf["dL_dK"], X, None)[0]
    for name, base, an in (("variance", 1.3, g[0]), ("lengthscale", 0.9, g[1][0])):
        e = 1e-5 * base
        def lml(val):
            kw = dict(variance=1.3, lengthscale=0.9); kw[name] = val
            return go.ep_inference(go.make_kernel(["RBF"], 2, [kw]), X, sites)["log_marginal"]
        fd = (lml(base + e) - lml(base - e)) / (2 * e)
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (name, fd, an)
    # predictions: high training accuracy, probabilities in (0, 1)
    mu, var = go.raw_predict(kern, X, X, inf["woodbury_vector"], inf["woodbury_inv"])
    p = go.bernoulli_predictive_mean(mu[:, 0], var[:, 0])
    assert np.all((p > 0) & (p < 1)) and np.mean((p > 0.5) == (y == 1)) > 0.85


def test_classifier_host_logic_on_oracle_evaluator(monkeypatch, tmp_path):
    from edrgp_amd import models
    from edrgp_amd.gp_model import GaussianProcessClassifier, SparseGaussianProcessClassifier
    monkeypatch.setattr(models, "HipEvaluator", lambda device=None: OracleEvaluator())
    X, y = _data(150, 3, 2)
    np.random.seed(3)
    clf = GaussianProcessClassifier(kernels="RBF", kernel_options={"ARD": True})
    clf.fit(X, y, max_iters=15)
    m = clf.estimator_
    assert m.parameter_names() == ["rbf.variance"] + ["rbf.lengthscale"] * 3       # kernel parameters only
    assert m.ep_sweeps >= 2 and m._sites is not None
    proba = clf.predict_proba(X)
    assert proba.shape == (150,) and np.all((proba > 0) & (proba < 1))
    assert np.mean(clf.predict(X) == (y == 1)) > 0.85
    assert clf.predict_gradient(X[:7]).shape == (7, 3)
    assert np.all(np.isnan(clf.predict_variance(X[:3])))
    # the objective's gradient in optimiser space against central differences (sites fixed, as during optimize())
    x0 = m.optimizer_array.copy()
    f0, g0 = m._objective_grads(x0)
    for i in range(x0.size):
        e = np.zeros_like(x0); e[i] = 1e-6
        fd = (m._objective_grads(x0 + e)[0] - m._objective_grads(x0 - e)[0]) / 2e-6
        assert abs(fd - g0[i]) <= 1e-4 * max(1.0, abs(g0[i]))
    m.optimizer_array = x0
    # persistence keeps the sites
    clf.save(str(tmp_path / "clf"))
    clf2 = GaussianProcessClassifier()
    clf2.load(str(tmp_path / "clf"))
    assert np.allclose(clf2.predict_proba(X[:20]), clf.predict_proba(X[:20]), atol=1e-12)
    with pytest.raises(NotImplementedError):
        SparseGaussianProcessClassifier().fit(X, y)
    with pytest.raises(ValueError):
        GaussianProcessClassifier().fit(X, y + 2.0)


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,kind", [(150, 2, "RBF"), (700, 5, "Matern52"), (1100, 3, "RBF")])
def test_device_ep_matches_oracle(hip, N, D, kind):
    X, y = _data(N, D, N)
    var, ls = 1.2, 0.8 * np.sqrt(D)
    kern = go.make_kernel([kind], D, [dict(variance=var, lengthscale=ls)])
    orders = _Orders(7)
    sites = go.expectation_propagation(kern.K(X), y, order_fn=orders)
    hip.set_data(X, y)
    hip.set_kernel([kind], [var], np.full((1, D), ls), 1.0)
    hip.ep_begin(y)
    sweeps, stop = 0, False
    while not stop:
        d_tau, d_v = hip.ep_sweep(orders.used[sweeps])           # the oracle's site order, sweep by sweep
        if sweeps > 0:
            stop = d_tau < 1e-6 and d_v < 1e-6
        sweeps += 1
        assert sweeps <= len(orders.used)
    assert sweeps == sites["sweeps"]
    logZt = hip.ep_finish()
    tau, v = hip.ep_get_sites()
    assert np.abs(tau - sites["tau"]).max() <= 1e-8 * np.abs(sites["tau"]).max()
    assert np.abs(v - sites["v"]).max() <= 1e-8 * np.abs(sites["v"]).max()
    assert abs(logZt - sites["log_Z_tilde"]) <= 1e-9 * max(abs(sites["log_Z_tilde"]), float(N))
    # fixed sites, other hyper-parameters (what the optimiser does): marginal, gradients, posterior
    var2, ls2 = 0.9, 1.1 * np.sqrt(D)
    kern2 = go.make_kernel([kind], D, [dict(variance=var2, lengthscale=ls2)])
    inf = go.ep_inference(kern2, X, sites)
    hip.set_kernel([kind], [var2], np.full((1, D), ls2), 1.0)
    lml, dvar, dlen = hip.ep_lml_grad()
    assert abs(lml - inf["log_marginal"]) <= 1e-9 * max(abs(inf["log_marginal"]), float(N))
    assert abs(hip.ep_lml() - lml) <= 1e-12 * abs(lml)
    g = kern2.param_gradients_full(inf["dL_dK"], X, None)[0]
    gs = max(abs(g[0]), abs(g[1][0]))
    assert abs(dvar[0] - g[0]) <= 1e-7 * gs and abs(dlen[0].sum() - g[1][0]) <= 1e-7 * gs
    hip.ep_posterior()
    Xs = np.vstack([np.random.default_rng(1).standard_normal((40, D)), X[:8]])
    mu_ref, var_ref = go.raw_predict(kern2, Xs, X, inf["woodbury_vector"], inf["woodbury_inv"])
    mu, vr = hip.predict(Xs)
    assert np.abs(mu - mu_ref[:, 0]).max() <= 1e-8 * max(1.0, np.abs(mu_ref).max())
    assert np.abs(vr - var_ref[:, 0]).max() <= 1e-8 * max(1.0, np.abs(var_ref).max())
    jac_ref = go.predictive_gradients_mean(kern2, Xs, X, inf["woodbury_vector"])[:, :, 0]
    assert np.abs(hip.predict_grad(Xs) - jac_ref).max() <= 1e-8 * max(1.0, np.abs(jac_ref).max())


@pytest.mark.gpu
def test_classifier_estimator_and_edr_end_to_end():
    from edrgp_amd import EffectiveDimensionalityReduction
    from edrgp_amd.gp_model import GaussianProcessClassifier
    from edrgp_amd.utils import SVDTransformer
    rng = np.random.default_rng(5)
    X = rng.standard_normal((600, 4))
    w = np.array([0.8, -0.6, 0.0, 0.0])
    y = ((X @ w) + 0.2 * rng.standard_normal(600) > 0).astype(float)
    np.random.seed(0)
    clf = GaussianProcessClassifier(kernels="RBF", kernel_options={"ARD": True}).fit(X, y, max_iters=60)
    assert np.mean(clf.predict(X) == (y == 1)) > 0.9
    p = clf.predict_proba(X)
    assert np.all((p > 0) & (p < 1))
    assert clf.estimator_.ep_sweeps >= 2
    # the latent-mean gradients point along the separating direction: EDR with the classifier finds it
    np.random.seed(1)
    edr = EffectiveDimensionalityReduction(GaussianProcessClassifier(), SVDTransformer(), n_components=1, normalize=True)
    edr.fit(X, y, max_iters=40)
    c = edr.components_[0] / np.linalg.norm(edr.components_[0])
    assert abs(c @ w) > 0.97


@pytest.mark.gpu
def test_device_ep_fixed_point_beyond_oracle_sizes(hip):
    """N = 3000 (the oracle's sequential sweeps would take minutes): size-independent properties.  At convergence the
    Gaussian marginals (mu, diag Sigma -- read back as the posterior's predictions at the training inputs) match the
    moments of the tilted distributions rebuilt on the host from the device's sites; the gradient of the fixed-site
    marginal agrees with central differences of the device's own marginal."""
    import time
    N, D = 3000, 4
    X, y = _data(N, D, 11)
    var, ls = 1.0, 1.5
    hip.set_data(X, y)
    hip.set_kernel(["RBF"], [var], np.full((1, D), ls), 1.0)
    t0 = time.perf_counter()
    hip.ep_begin(y)
    rng = np.random.default_rng(0)
    sweeps, stop = 0, False
    while not stop and sweeps < 40:
        d_tau, d_v = hip.ep_sweep(rng.permutation(N))
        stop = sweeps > 0 and d_tau < 1e-6 and d_v < 1e-6
        sweeps += 1
    hip.ep_finish()
    t1 = time.perf_counter()
    print(f"EP N={N}: {sweeps} sweeps in {t1 - t0:.2f} s")
    assert stop and sweeps <= 12
    tau, v = hip.ep_get_sites()
    assert np.all(tau > 0)
    hip.ep_posterior()
    mu, s2 = hip.predict(X)
    tau_c, v_c = 1.0 / s2 - tau, mu / s2 - v
    mom = np.array([go.bernoulli_moments_match_ep(y[i], tau_c[i], v_c[i]) for i in range(N)])
    assert np.abs(mom[:, 1] - mu).max() < 5e-3 and np.abs(mom[:, 2] - s2).max() < 5e-3
    lml, dvar, dlen = hip.ep_lml_grad()
    for name, an in (("var", dvar[0]), ("ls", dlen[0].sum())):
        e = 1e-5
        vals = []
        for sgn in (1, -1):
            hip.set_kernel(["RBF"], [var + sgn * e * (name == "var")], np.full((1, D), ls + sgn * e * (name == "ls")), 1.0)
            vals.append(hip.ep_lml())
        fd = (vals[0] - vals[1]) / (2 * e)
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (name, fd, an)
