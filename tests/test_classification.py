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
    with pytest.raises(ValueError):
        GaussianProcessClassifier().fit(X, y + 2.0)


def test_oracle_epdtc_fixed_point_and_gradients():
    """EPDTC restatement: EP's fixed point holds on the DTC posterior (the Gaussian marginals the VarDTC posterior
    predicts at the training rows are the ones the site updates tracked, and they match the tilted moments), the
    gradients of the fixed-site bound w.r.t. kernel parameters and Z agree with central differences, and with
    Z = X the inducing-point posterior is the dense one: the sites of the dense EP are a fixed point of EPDTC too."""
    X, y = _data(220, 2, 4)
    rng = np.random.default_rng(0)
    Z = X[rng.permutation(220)[:25]].copy()
    kw = dict(variance=1.3, lengthscale=0.9)
    kern = go.make_kernel(["RBF"], 2, [kw])
    sites = go.expectation_propagation_dtc(kern.K(Z), kern.K(Z, X), y, order_fn=_Orders(1))
    assert 2 <= sites["sweeps"] <= 30 and np.all(sites["tau"] > 0)
    inf = go.epdtc_inference(kern, X, Z, sites)
    mu, var = go.raw_predict(kern, X, Z, inf["woodbury_vector"], inf["woodbury_inv"])
    Qnn = np.sum(go.dtrtrs(inf["Lm"], kern.K(Z, X), lower=1) ** 2, 0)
    # the VarDTC posterior at the training rows = EP's (mu, Sigma_diag) + the DTC residual Knn - Qnn
    assert np.abs(mu[:, 0] - sites["mu"]).max() < 1e-10
    assert np.abs(var[:, 0] - (kern.Kdiag(X) - Qnn + sites["Sigma_diag"])).max() < 1e-10
    assert np.abs(sites["mu"] - sites["mu_hat"]).max() < 5e-3
    assert np.abs(sites["Sigma_diag"] - sites["sigma2_hat"]).max() < 5e-3
    kg, _, dZ = go.sparse_gradients(kern, X, Z, inf)

    def lml(var_=1.3, ls_=0.9, Z_=Z):
        return go.epdtc_inference(go.make_kernel(["RBF"], 2, [dict(variance=var_, lengthscale=ls_)]), X, Z_, sites)["log_marginal"]
    e = 1e-6
    assert abs((lml(var_=1.3 + e) - lml(var_=1.3 - e)) / (2 * e) - kg[0][0]) <= 1e-6 * max(1.0, abs(kg[0][0]))
    assert abs((lml(ls_=0.9 + e) - lml(ls_=0.9 - e)) / (2 * e) - kg[0][1][0]) <= 1e-6 * max(1.0, abs(kg[0][1][0]))
    for (m, q) in ((0, 0), (7, 1), (24, 0)):
        Zp, Zm = Z.copy(), Z.copy()
        Zp[m, q] += e; Zm[m, q] -= e
        assert abs((lml(Z_=Zp) - lml(Z_=Zm)) / (2 * e) - dZ[m, q]) <= 1e-5 * max(1.0, np.abs(dZ).max())
    # Z = X: Qnn = Knn, and one more EPDTC sweep from the dense sites leaves them where they are
    Xs, ys = X[:60], y[:60]
    dense = go.expectation_propagation(kern.K(Xs), ys, order_fn=_Orders(2), epsilon=1e-12)
    sp = go.expectation_propagation_dtc(kern.K(Xs), kern.K(Xs), ys, order_fn=_Orders(2), epsilon=1e-12)
    assert np.abs(sp["tau"] - dense["tau"]).max() <= 1e-4 * dense["tau"].max()
    assert np.abs(sp["v"] - dense["v"]).max() <= 1e-4 * np.abs(dense["v"]).max()


def test_sparse_classifier_host_logic_on_oracle_evaluator(monkeypatch, tmp_path):
    from edrgp_amd import models
    from edrgp_amd.gp_model import SparseGaussianProcessClassifier
    monkeypatch.setattr(models, "HipEvaluator", lambda device=None: OracleEvaluator())
    X, y = _data(160, 3, 6)
    np.random.seed(4)
    clf = SparseGaussianProcessClassifier(kernels="RBF", num_inducing=12)
    clf.fit(X, y, max_iters=10)
    m = clf.estimator_
    assert m.Z.shape == (12, 3) and m.is_sparse and m.is_classifier
    assert m.parameter_names() == ["inducing_inputs"] * 36 + ["rbf.variance", "rbf.lengthscale"]
    assert m.ep_sweeps >= 2 and m._sites is not None
    proba = clf.predict_proba(X)
    assert proba.shape == (160,) and np.all((proba > 0) & (proba < 1))
    assert np.mean(clf.predict(X) == (y == 1)) > 0.8
    assert clf.predict_gradient(X[:7]).shape == (7, 3)
    x0 = m.optimizer_array.copy()
    f0, g0 = m._objective_grads(x0)
    for i in (0, 5, 17, 35, 36, 37):
        e = np.zeros_like(x0); e[i] = 1e-6
        fd = (m._objective_grads(x0 + e)[0] - m._objective_grads(x0 - e)[0]) / 2e-6
        assert abs(fd - g0[i]) <= 1e-4 * max(1.0, abs(g0[i]))
    m.optimizer_array = x0
    clf.save(str(tmp_path / "sclf"))
    clf2 = SparseGaussianProcessClassifier()
    clf2.load(str(tmp_path / "sclf"))
    assert np.allclose(clf2.predict_proba(X[:20]), clf.predict_proba(X[:20]), atol=1e-12)
    with pytest.raises(NotImplementedError):
        SparseGaussianProcessClassifier(likelihood="other").fit(X, y)


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


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,M,kind", [(150, 2, 10, "RBF"), (600, 5, 40, "Matern52"), (900, 3, 130, "RBF"),
                                        (400, 4, 30, "RBF+Matern32"), (70, 2, 1, "RBF"), (90, 3, 3, "Matern32")])
def test_device_epdtc_matches_oracle(hip, N, D, M, kind):
    """EPDTC on the device (rank-one updates of B^-1 in the whitened basis) against the oracle's restatement of
    upstream's per-site refactorisation, for the SAME site orders: sweeps, sites, log Z_tilde; then the fixed-site
    bound, its gradients (kernel parameters and Z) and the posterior at other hyper-parameters."""
    X, y = _data(N, D, N + M)
    kinds = kind.split("+")
    rng = np.random.default_rng(M)
    Z = X[rng.permutation(N)[:M]].copy()

    def kern_of(scale):
        opts = [dict(variance=1.2 * scale / (i + 1), lengthscale=(0.8 + 0.3 * i) * np.sqrt(D) * scale) for i in range(len(kinds))]
        return go.make_kernel(kinds, D, opts), [o["variance"] for o in opts], np.array([[o["lengthscale"]] * D for o in opts])

    kern, var, ls = kern_of(1.0)
    orders = _Orders(7)
    sites = go.expectation_propagation_dtc(kern.K(Z), kern.K(Z, X), y, order_fn=orders)
    hip.set_data(X, y)
    hip.set_inducing(Z)
    hip.set_kernel(kinds, var, ls, 1.0)
    hip.epdtc_begin(y)
    sweeps, stop = 0, False
    while not stop:
        d_tau, d_v = hip.epdtc_sweep(orders.used[sweeps])
        if sweeps > 0:
            stop = d_tau < 1e-6 and d_v < 1e-6
        sweeps += 1
        assert sweeps <= len(orders.used)
    assert sweeps == sites["sweeps"]
    logZt = hip.ep_finish()
    tau, v = hip.ep_get_sites()
    assert np.abs(tau - sites["tau"]).max() <= 1e-7 * np.abs(sites["tau"]).max()
    assert np.abs(v - sites["v"]).max() <= 1e-7 * np.abs(sites["v"]).max()
    assert abs(logZt - sites["log_Z_tilde"]) <= 1e-8 * max(abs(sites["log_Z_tilde"]), float(N))
    # fixed sites, other hyper-parameters and inducing inputs (what the optimiser does)
    kern2, var2, ls2 = kern_of(1.15)
    Z2 = Z + 0.05 * rng.standard_normal(Z.shape)
    dev_sites = dict(tau=tau, v=v, log_Z_tilde=logZt)
    inf = go.epdtc_inference(kern2, X, Z2, dev_sites)
    hip.set_inducing(Z2)
    hip.set_kernel(kinds, var2, ls2, 1.0)
    lml, dvar, dlen, dZ = hip.epdtc_lml_grad()
    assert abs(lml - inf["log_marginal"]) <= 1e-9 * max(abs(inf["log_marginal"]), float(N))
    assert abs(hip.epdtc_lml() - lml) <= 1e-12 * abs(lml)
    kg, _, dZ_ref = go.sparse_gradients(kern2, X, Z2, inf)
    gs = max(max(abs(g[0]), abs(g[1][0])) for g in kg)
    for p, g in enumerate(kg):
        assert abs(dvar[p] - g[0]) <= 1e-7 * gs and abs(dlen[p].sum() - g[1][0]) <= 1e-7 * gs
    assert np.abs(dZ - dZ_ref).max() <= 1e-7 * max(1.0, np.abs(dZ_ref).max())
    hip.epdtc_posterior()
    Xs = np.vstack([np.random.default_rng(1).standard_normal((40, D)), X[:8]])
    mu_ref, var_ref = go.raw_predict(kern2, Xs, Z2, inf["woodbury_vector"], inf["woodbury_inv"])
    mu, vr = hip.predict(Xs)
    assert np.abs(mu - mu_ref[:, 0]).max() <= 1e-8 * max(1.0, np.abs(mu_ref).max())
    assert np.abs(vr - var_ref[:, 0]).max() <= 1e-8 * max(1.0, np.abs(var_ref).max())
    jac_ref = go.predictive_gradients_mean(kern2, Xs, Z2, inf["woodbury_vector"])[:, :, 0]
    assert np.abs(hip.predict_grad(Xs) - jac_ref).max() <= 1e-8 * max(1.0, np.abs(jac_ref).max())


@pytest.mark.gpu
def test_sparse_classifier_estimator_and_edr_end_to_end():
    from edrgp_amd import EffectiveDimensionalityReduction
    from edrgp_amd.gp_model import SparseGaussianProcessClassifier
    from edrgp_amd.utils import SVDTransformer
    rng = np.random.default_rng(5)
    X = rng.standard_normal((800, 4))
    w = np.array([0.8, -0.6, 0.0, 0.0])
    y = ((X @ w) + 0.2 * rng.standard_normal(800) > 0).astype(float)
    np.random.seed(0)
    clf = SparseGaussianProcessClassifier(kernels="RBF", kernel_options={"ARD": True}, num_inducing=30).fit(X, y, max_iters=60)
    assert np.mean(clf.predict(X) == (y == 1)) > 0.9
    p = clf.predict_proba(X)
    assert np.all((p > 0) & (p < 1))
    m = clf.estimator_
    assert m.ep_sweeps >= 2 and m.Z.shape == (30, 4)
    # the optimiser-space gradient of the device objective against its own central differences (sites fixed)
    x0 = m.optimizer_array.copy()
    f0, g0 = m._objective_grads(x0)
    for i in (0, 41, 119, 120, 122):
        e = np.zeros_like(x0); e[i] = 1e-5
        fd = (m._objective_grads(x0 + e)[0] - m._objective_grads(x0 - e)[0]) / 2e-5
        assert abs(fd - g0[i]) <= 1e-4 * max(1.0, np.abs(g0).max())
    m.optimizer_array = x0
    np.random.seed(1)
    edr = EffectiveDimensionalityReduction(SparseGaussianProcessClassifier(num_inducing=30), SVDTransformer(), n_components=1,
                                           normalize=True)
    edr.fit(X, y, max_iters=40)
    c = edr.components_[0] / np.linalg.norm(edr.components_[0])
    assert abs(c @ w) > 0.95


@pytest.mark.gpu
def test_device_epdtc_fixed_point_beyond_oracle_sizes(hip):
    """N = 20000, M = 256 (the oracle refactors an M x M matrix and solves against K_mn per site: hours).  Properties:
    convergence in a few sweeps; at the fixed point the VarDTC posterior's marginals at the training rows, less the DTC
    residual Knn - Qnn, match the tilted moments rebuilt on the host from the device's sites."""
    import time
    N, D, M = 20000, 4, 256
    X, y = _data(N, D, 13)
    rng = np.random.default_rng(2)
    Z = X[rng.permutation(N)[:M]].copy()
    var, ls = 1.0, 1.5
    hip.set_data(X, y)
    hip.set_inducing(Z)
    hip.set_kernel(["RBF"], [var], np.full((1, D), ls), 1.0)
    t0 = time.perf_counter()
    hip.epdtc_begin(y)
    sweeps, stop = 0, False
    while not stop and sweeps < 40:
        d_tau, d_v = hip.epdtc_sweep(rng.permutation(N))
        stop = sweeps > 0 and d_tau < 1e-6 and d_v < 1e-6
        sweeps += 1
    hip.ep_finish()
    print(f"EPDTC N={N} M={M}: {sweeps} sweeps in {time.perf_counter() - t0:.2f} s")
    assert stop and sweeps <= 15
    tau, v = hip.ep_get_sites()
    assert np.all(tau > 0)
    hip.epdtc_posterior()
    mu, s2 = hip.predict(X)
    kern = go.make_kernel(["RBF"], D, [dict(variance=var, lengthscale=ls)])
    Lm = go.jitchol(kern.K(Z))
    Qnn = np.sum(go.dtrtrs(Lm, kern.K(Z, X), lower=1) ** 2, 0)
    sd = s2 - (var - Qnn)                                  # Sigma_diag of the DTC posterior EP worked on
    tau_c, v_c = 1.0 / sd - tau, mu / sd - v
    idx = rng.permutation(N)[:2000]
    mom = np.array([go.bernoulli_moments_match_ep(y[i], tau_c[i], v_c[i]) for i in idx])
    assert np.abs(mom[:, 1] - mu[idx]).max() < 5e-3 and np.abs(mom[:, 2] - sd[idx]).max() < 5e-3
