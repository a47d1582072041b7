"""GPU parity of the inference pipelines (C ABI) against the CPU oracle.

Sparse: Titsias bound, its gradients, posterior factors, predictions
(GPy VarDTC.inference / SparseGP._update_gradients / Posterior._raw_predict).
Dense: exact LML, gradients, predictions (ExactGaussianInference).

Tolerances (fp64): bound <= 1e-10 relative, posterior mean <= 1e-8 relative
(SURVEY.md section 8d), gradients <= 1e-7 relative to the gradient norm.
"""
import numpy as np
import pytest

from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def set_kernel(hip, kern, noise):
    D = kern.input_dim
    kinds = [p.kind for p in kern.parts]
    var = [p.variance for p in kern.parts]
    ls = np.array([np.broadcast_to(p.lengthscale, (D,)) for p in kern.parts])
    hip.set_kernel(kinds, var, ls, noise)


def oracle_sparse(kern, X, y, Z, noise):
    inf = go.vardtc_inference(kern, X, Z, noise, y[:, None])
    kg, dn, dZ = go.sparse_gradients(kern, X, Z, inf)
    return inf, kg, dn, dZ


def fold_dlen(kern, dlen):
    """device returns per-dimension lengthscale gradients; sum them for isotropic parts"""
    out = []
    for p, part in enumerate(kern.parts):
        out.append(dlen[p] if part.ARD else np.array([dlen[p].sum()]))
    return out


CASES = [
    # (N, M, D, kinds, options, noise)
    (50, 12, 1, ["RBF"], [dict(variance=1.0, lengthscale=1.0)], 1.0),
    (256, 32, 4, ["RBF"], [dict(variance=0.8, lengthscale=1.7)], 0.05),
    (256, 32, 4, ["Matern32"], [dict(variance=1.3, lengthscale=2.0)], 0.1),
    (512, 64, 8, ["Matern52"], [dict(variance=1.1, lengthscale=2.5)], 0.02),
    (700, 100, 6, ["RBF"], [dict(variance=0.9, lengthscale=[1.5, 2.0, 2.5, 3.0, 3.5, 4.0], ARD=True)], 0.05),
    (600, 70, 5, ["RBF", "Matern52"], [dict(variance=0.6, lengthscale=2.0), dict(variance=0.3, lengthscale=1.2)], 0.05),
    (3000, 200, 16, ["RBF"], [dict(variance=1.0, lengthscale=4.0)], 0.01),
    # D > 32: RBF keeps the fused gradient epilogue (k read back from K_nm), Matern takes the unfused path
    (1500, 130, 40, ["RBF"], [dict(variance=1.2, lengthscale=6.0)], 0.05),
    (1500, 130, 40, ["Matern52"], [dict(variance=1.2, lengthscale=6.0)], 0.05),
]


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("N,M,D,kinds,opts,noise", CASES)
def test_sparse_bound_gradients_posterior(hip, N, M, D, kinds, opts, noise, mode):
    """mode 0 = automatic choice between the Gram form and the stable form
    (pivot-ratio proxy of cond(Kmm)); mode 2 = stable form forced."""
    X, y, Z = go.synthetic_problem(N, D, M, seed=N + M)
    kern = go.make_kernel(kinds, D, opts)
    inf, kg, dn, dZ = oracle_sparse(kern, X, y, Z, noise)

    hip.set_option("stats_mode", mode)
    try:
        hip.set_data(X, y)
        hip.set_inducing(Z)
        set_kernel(hip, kern, noise)

        lml = hip.sparse_lml()
        ref = inf["log_marginal"][0, 0]
        assert abs(lml - ref) <= 1e-10 * abs(ref), (lml, ref, hip.get("MODE", (1,)))
        assert rel(hip.get("LM", (M, M)), inf["Lm"]) < 1e-9
        assert rel(hip.get("A", (M, M)), inf["A"]) < 1e-8

        lml2, dvar, dlen, dnoise, dZd = hip.sparse_lml_grad()
        assert abs(lml2 - ref) <= 1e-10 * abs(ref)
        gscale = max(abs(dn), max(abs(g[0]) for g in kg), max(np.abs(g[1]).max() for g in kg))
        for p, (gv, gl) in enumerate(kg):
            assert abs(dvar[p] - gv) <= 1e-7 * gscale, ("dvar", p, dvar[p], gv)
            assert np.abs(fold_dlen(kern, dlen)[p] - gl).max() <= 1e-7 * gscale, ("dlen", p)
        assert abs(dnoise - dn) <= 1e-7 * gscale, (dnoise, dn)
        assert np.abs(dZd - dZ).max() <= 1e-6 * max(np.abs(dZ).max(), 1e-12), "dZ"

        lml3 = hip.sparse_posterior()
        assert abs(lml3 - ref) <= 1e-10 * abs(ref)
        # woodbury factors are conditioning-sensitive (errors live in the small
        # eigen-directions of Kmm); the posterior mean below is the pinned quantity
        assert rel(hip.get("WV", (M,)), inf["woodbury_vector"][:, 0]) < 1e-5
        assert rel(hip.get("WINV", (M, M)), inf["woodbury_inv"]) < 1e-3

        rng = np.random.default_rng(1)
        Xs = np.vstack([rng.standard_normal((40, D)), X[:8], Z[:5]])
        mu_ref, var_ref = go.raw_predict(kern, Xs, Z, inf["woodbury_vector"], inf["woodbury_inv"])
        jac_ref = go.predictive_gradients_mean(kern, Xs, Z, inf["woodbury_vector"])[:, :, 0]
        mu, var = hip.predict(Xs)
        assert rel(mu, mu_ref[:, 0]) < 1e-8
        assert np.abs(var - var_ref[:, 0]).max() < 1e-7 * max(1.0, np.abs(var_ref).max())
        jac = hip.predict_grad(Xs)
        assert rel(jac, jac_ref) < 1e-8

        mu_t, jac_t = hip.predict_train()
        mu_tr, _ = go.raw_predict(kern, X, Z, inf["woodbury_vector"], inf["woodbury_inv"])
        assert rel(mu_t, mu_tr[:, 0]) < 1e-8
        assert rel(jac_t, go.predictive_gradients_mean(kern, X, Z, inf["woodbury_vector"])[:, :, 0]) < 1e-8
    finally:
        hip.set_option("stats_mode", 0)


WELL_CONDITIONED = [c for i, c in enumerate(CASES) if i in (2, 3, 5, 6)]


@pytest.mark.parametrize("N,M,D,kinds,opts,noise", WELL_CONDITIONED)
def test_sparse_gram_form_statistics(hip, N, M, D, kinds, opts, noise):
    """Gram form forced: Phi = K_mn K_nm and b = K_mn y are the all-reduced
    statistics of the multi-GPU path; check them and the bound."""
    X, y, Z = go.synthetic_problem(N, D, M, seed=N + M)
    kern = go.make_kernel(kinds, D, opts)
    hip.set_option("stats_mode", 1)
    try:
        hip.set_data(X, y)
        hip.set_inducing(Z)
        set_kernel(hip, kern, noise)
        lml = hip.sparse_lml()
        assert hip.get("MODE", (1,))[0] == 1.0
        st = go.gram_stats(kern, X, Z, y[:, None])
        assert rel(hip.get("PHI", (M, M)), st["Phi"]) < 1e-12
        assert rel(hip.get("BVEC", (M,)), st["b"][:, 0]) < 1e-12
        ref = go.vardtc_inference(kern, X, Z, noise, y[:, None])["log_marginal"][0, 0]
        assert abs(lml - ref) <= 1e-10 * abs(ref)
    finally:
        hip.set_option("stats_mode", 0)


def test_sparse_chunked_rows_equal_unchunked(hip):
    N, M, D = 1500, 96, 7
    X, y, Z = go.synthetic_problem(N, D, M, seed=3)
    kern = go.make_kernel(["Matern52"], D, [dict(variance=1.2, lengthscale=2.2)])
    hip.set_data(X, y)
    hip.set_inducing(Z)
    set_kernel(hip, kern, 0.03)
    hip.set_option("chunk_rows", 0)
    a = hip.sparse_lml_grad()
    hip.set_option("chunk_rows", 400)          # 4 ragged chunks
    b = hip.sparse_lml_grad()
    hip.set_option("chunk_rows", 0)
    inf, kg, dn, dZ = oracle_sparse(kern, X, y, Z, 0.03)
    ref = inf["log_marginal"][0, 0]
    for r in (a, b):
        assert abs(r[0] - ref) <= 1e-10 * abs(ref)
        assert abs(r[1][0] - kg[0][0]) <= 1e-7 * abs(kg[0][0])
        assert np.abs(r[4] - dZ).max() <= 1e-7 * np.abs(dZ).max()


def test_automatic_chunking_keeps_all_rows_resident_when_they_fit(hip):
    """N above the 131072-row chunk of the chunked path: the automatic choice ("chunk_rows" = 0, the default) keeps the
    whole K_nm resident (one row pass, K_nm and h(r) reused by the gradient pass and by predictions on the training
    rows) and gives the numbers of the explicitly chunked run."""
    N, M, D = 300000, 256, 8
    X, y, Z = go.synthetic_problem(N, D, M, seed=5)
    hip.set_data(X, y)
    hip.set_inducing(Z)
    hip.set_kernel(["Matern32"], [1.1], np.full((1, D), 1.3 * np.sqrt(D)), 0.02)
    out = {}
    for chunk in (0, 131072):
        hip.set_option("chunk_rows", chunk)
        g = hip.sparse_lml_grad()
        hip.sparse_posterior()
        mu, jac = hip.predict_train(True, True)
        if chunk == 0:                              # all N rows of K_nm are resident: a 131072-row buffer is refused
            from edrgp_amd._lib import EdrgpError
            with pytest.raises(EdrgpError):
                hip.get("KNM", (131072, M))
            hip.get("KNM", (N, M))
        else:
            hip.get("KNM", (131072, M))
        out[chunk] = (g, mu, jac)
    hip.set_option("chunk_rows", 0)
    a, b = out[0], out[131072]
    assert abs(a[0][0] - b[0][0]) <= 1e-11 * abs(b[0][0])
    assert np.abs(a[0][2] - b[0][2]).max() <= 1e-8 * np.abs(b[0][2]).max()
    assert np.abs(a[0][4] - b[0][4]).max() <= 1e-8 * np.abs(b[0][4]).max()
    assert np.abs(a[1] - b[1]).max() <= 1e-9 * max(1.0, np.abs(b[1]).max())
    assert np.abs(a[2] - b[2]).max() <= 1e-8 * max(1.0, np.abs(b[2]).max())


@pytest.mark.parametrize("N,M,D,mode", [(900, 65, 3, 0), (1500, 130, 6, 0), (2500, 192, 5, 2), (4000, 449, 9, 0)])
def test_bound_only_rhs_row_equals_explicit_inverse(hip, N, M, D, mode):
    """Bound-only evaluations carry beta*v through the second factorisation as an extra matrix row (M + 1
    rows, ragged last block) instead of forming L_B^-1: same bound as the explicit form and as the oracle."""
    X, y, Z = go.synthetic_problem(N, D, M, seed=N + M)
    kern = go.make_kernel(["RBF"], D, [dict(variance=1.1, lengthscale=1.5 * np.sqrt(D))])
    hip.set_data(X, y)
    hip.set_inducing(Z)
    set_kernel(hip, kern, 0.04)
    hip.set_option("stats_mode", mode)
    try:
        explicit = hip.sparse_posterior()          # posterior evaluations form L_B^-1 explicitly
        row = hip.sparse_lml()
        again = hip.sparse_lml()
    finally:
        hip.set_option("stats_mode", 0)
    ref = go.vardtc_inference(kern, X, Z, 0.04, y[:, None])["log_marginal"][0, 0]
    assert row == again
    assert abs(row - explicit) <= 1e-12 * abs(explicit)
    assert abs(row - ref) <= 1e-10 * abs(ref)


@pytest.mark.parametrize("N,M,D,kinds,mode", [(1200, 300, 4, ["RBF"], 1), (1500, 449, 6, ["Matern52"], 0),
                                              (1300, 257, 3, ["RBF", "Matern32"], 1), (1400, 390, 5, ["RBF"], 2)])
def test_padded_inducing_set(hip, N, M, D, kinds, mode):
    """M > 256 and not a multiple of 128: the evaluation works on the next multiple with decoupled phantom
    inducing points.  Bound, gradients, posterior and the debug accessors must be those of the M-point problem
    (oracle), and equal to the unpadded evaluation."""
    X, y, Z = go.synthetic_problem(N, D, M, seed=N + M)
    opts = [dict(variance=0.9 + 0.2 * i, lengthscale=(0.5 + 0.3 * i) * np.sqrt(D)) for i in range(len(kinds))]
    kern = go.make_kernel(kinds, D, opts)
    noise = 0.05
    hip.set_data(X, y)
    hip.set_inducing(Z)
    set_kernel(hip, kern, noise)
    hip.set_option("stats_mode", mode)
    # both sides lose digits in proportion to cond(Kmm) (as in tools/fuzz.py)
    amp = max(1.0, np.linalg.cond(kern.K(Z) + 1e-8 * np.eye(M)) / 1e6)
    res = {}
    try:
        for pad in (0, 1):
            hip.set_option("pad_inducing", pad)
            lml_only = hip.sparse_lml()
            out = hip.sparse_lml_grad()
            hip.sparse_posterior()
            mu, var = hip.predict(X[:40])
            jac = hip.predict_grad(X[:40])
            res[pad] = (lml_only, out, mu, var, jac, hip.get("LM", M * M).reshape(M, M), hip.get("WV", M),
                        hip.get("BVEC", M))
    finally:
        hip.set_option("pad_inducing", 1)
        hip.set_option("stats_mode", 0)
    inf, kg, dn, dZ = oracle_sparse(kern, X, y, Z, noise)
    ref = inf["log_marginal"][0, 0]
    mu_ref, var_ref = go.raw_predict(kern, X[:40], Z, inf["woodbury_vector"], inf["woodbury_inv"])
    jac_ref = go.predictive_gradients_mean(kern, X[:40], Z, inf["woodbury_vector"])[:, :, 0]
    gscale = max(abs(dn), max(abs(g[0]) for g in kg), max(np.abs(g[1]).max() for g in kg))
    for pad in (0, 1):
        lml_only, out, mu, var, jac, Lm, wv, bvec = res[pad]
        assert abs(lml_only - ref) <= 1e-9 * amp * abs(ref)
        assert abs(out[0] - ref) <= 1e-9 * amp * abs(ref)
        for p in range(len(kinds)):
            assert abs(out[1][p] - kg[p][0]) <= 1e-6 * amp * gscale
            assert abs(out[2][p].sum() - np.sum(kg[p][1])) <= 1e-6 * amp * gscale
        assert abs(out[3] - dn) <= 1e-6 * amp * gscale
        assert out[4].shape == (M, D)
        assert np.abs(out[4] - dZ).max() <= 1e-5 * amp * max(np.abs(dZ).max(), 1e-3 * gscale)
        assert np.abs(mu - mu_ref[:, 0]).max() <= 1e-7 * amp * np.abs(mu_ref).max()
        assert np.abs(var - var_ref[:, 0]).max() <= 1e-6 * amp * max(1.0, np.abs(var_ref).max())
        assert np.abs(jac - jac_ref).max() <= 1e-7 * amp * np.abs(jac_ref).max()
        assert np.abs(wv - inf["woodbury_vector"][:, 0]).max() <= 1e-6 * amp * np.abs(inf["woodbury_vector"]).max()
    # the real blocks of the padded evaluation are the unpadded ones
    assert np.abs(res[1][5] - res[0][5]).max() <= 1e-12 * np.abs(res[0][5]).max()
    assert np.abs(res[1][7] - res[0][7]).max() <= 1e-12 * np.abs(res[0][7]).max()
    assert abs(res[1][0] - res[0][0]) <= 1e-11 * abs(res[0][0])


@pytest.mark.parametrize("M", [96, 300])
def test_posterior_reuses_last_gradient_evaluation(hip, M):
    """A posterior request right after a bound+gradient evaluation of the same state is served from that
    evaluation's workspaces (GPy returns the posterior with every inference call): same numbers as a full
    posterior evaluation; any change of state in between turns the shortcut off."""
    N, D = 1500, 5
    X, y, Z = go.synthetic_problem(N, D, M, seed=40 + M)
    kern = go.make_kernel(["RBF"], D, [dict(variance=1.1, lengthscale=0.7 * np.sqrt(D))])
    hip.set_data(X, y)
    hip.set_inducing(Z)
    set_kernel(hip, kern, 0.05)
    lml_g = hip.sparse_lml_grad()[0]
    lml_p = hip.sparse_posterior()                       # shortcut
    wv1, winv1 = hip.get("WV", M), hip.get("WINV", M * M)
    mu1, var1 = hip.predict(X[:64])
    jac1 = hip.predict_grad(X[:64])
    assert lml_p == lml_g
    try:
        hip.set_option("posterior_reuse", 0)
        hip.sparse_lml_grad()
        lml_f = hip.sparse_posterior()                   # full evaluation
    finally:
        hip.set_option("posterior_reuse", 1)
    wv2, winv2 = hip.get("WV", M), hip.get("WINV", M * M)
    mu2, var2 = hip.predict(X[:64])
    jac2 = hip.predict_grad(X[:64])
    assert abs(lml_f - lml_g) <= 1e-13 * abs(lml_g)
    assert np.array_equal(wv1, wv2) and np.array_equal(winv1, winv2)
    assert np.array_equal(mu1, mu2) and np.array_equal(var1, var2) and np.array_equal(jac1, jac2)
    # a change of state between the two calls: the posterior must be the one of the new state
    hip.sparse_lml_grad()
    kern2 = go.make_kernel(["RBF"], D, [dict(variance=0.8, lengthscale=1.1 * np.sqrt(D))])
    set_kernel(hip, kern2, 0.08)
    hip.sparse_posterior()
    inf = go.vardtc_inference(kern2, X, Z, 0.08, y[:, None])
    mu_ref, _ = go.raw_predict(kern2, X[:64], Z, inf["woodbury_vector"], inf["woodbury_inv"])
    amp = max(1.0, np.linalg.cond(kern2.K(Z) + 1e-8 * np.eye(M)) / 1e6)
    assert np.abs(hip.predict(X[:64])[0] - mu_ref[:, 0]).max() <= 1e-7 * amp * np.abs(mu_ref).max()


def test_sparse_jitchol_retry_duplicate_inducing(hip):
    """Z with exact duplicates: Kmm + 1e-8 I is numerically singular for a wide RBF;
    the jitter ladder (GPy jitchol) must take over or succeed; result must be finite."""
    N, M, D = 200, 24, 2
    X, y, Z = go.synthetic_problem(N, D, M, seed=9)
    Z[1] = Z[0]
    Z[3] = Z[2]
    kern = go.make_kernel(["RBF"], D, [dict(variance=1.0, lengthscale=30.0)])
    hip.set_data(X, y)
    hip.set_inducing(Z)
    set_kernel(hip, kern, 0.1)
    lml = hip.sparse_lml()
    assert np.isfinite(lml)
    ref = go.vardtc_inference(kern, X, Z, 0.1, y[:, None])["log_marginal"][0, 0]
    assert abs(lml - ref) <= 1e-3 * abs(ref)      # ill-conditioned: loose, documented


DENSE_CASES = [
    (50, 1, ["RBF"], [dict(variance=1.0, lengthscale=1.0)], 1.0),
    (300, 4, ["RBF"], [dict(variance=0.7, lengthscale=1.3)], 0.05),
    (500, 2, ["Matern52"], [dict(variance=1.5, lengthscale=0.9)], 0.1),
    (333, 5, ["RBF"], [dict(variance=0.7, lengthscale=[0.5, 1.0, 2.0, 4.0, 1.5], ARD=True)], 0.02),
    (260, 3, ["Matern32", "RBF"], [dict(variance=0.5, lengthscale=1.0), dict(variance=0.4, lengthscale=3.0)], 0.05),
]


@pytest.mark.parametrize("N,D,kinds,opts,noise", DENSE_CASES)
def test_dense_lml_gradients_predict(hip, N, D, kinds, opts, noise):
    X, y, _ = go.synthetic_problem(N, D, 4, seed=N)
    kern = go.make_kernel(kinds, D, opts)
    inf = go.exact_inference(kern, X, noise, y[:, None])
    kg, dn = go.dense_gradients(kern, X, inf)
    hip.set_data(X, y)
    set_kernel(hip, kern, noise)
    assert abs(hip.dense_lml() - inf["log_marginal"]) <= 1e-10 * abs(inf["log_marginal"])
    lml, dvar, dlen, dnoise = hip.dense_lml_grad()
    assert abs(lml - inf["log_marginal"]) <= 1e-10 * abs(inf["log_marginal"])
    gscale = max(abs(dn), max(abs(g[0]) for g in kg), max(np.abs(g[1]).max() for g in kg))
    for p, (gv, gl) in enumerate(kg):
        assert abs(dvar[p] - gv) <= 1e-8 * gscale
        assert np.abs(fold_dlen(kern, dlen)[p] - gl).max() <= 1e-8 * gscale
    assert abs(dnoise - dn) <= 1e-8 * gscale
    hip.dense_posterior()
    rng = np.random.default_rng(2)
    Xs = np.vstack([rng.standard_normal((30, D)), X[:6]])
    mu_ref, var_ref = go.raw_predict(kern, Xs, X, inf["woodbury_vector"], inf["woodbury_inv"])
    jac_ref = go.predictive_gradients_mean(kern, Xs, X, inf["woodbury_vector"])[:, :, 0]
    mu, var = hip.predict(Xs)
    assert rel(mu, mu_ref[:, 0]) < 1e-8
    assert np.abs(var - var_ref[:, 0]).max() < 1e-8 * max(1.0, np.abs(var_ref).max())
    assert rel(hip.predict_grad(Xs), jac_ref) < 1e-8


@pytest.mark.parametrize("N,D,kind", [(2500, 6, "RBF"), (4100, 8, "Matern52")])
def test_dense_exact_gp_beyond_the_plumbing_size(hip, N, D, kind):
    """the exact GP (edrgp/gp_model/regression.py:10-77) at sizes beyond config 1's N = 500.  N = 2500
    runs the fused factorisation + inverse (one launch per 64-column step), N = 4100 the path beyond 3072
    (two launches per step, recursive-doubling inverse).  Bound, gradients, posterior mean / variance / Jacobian
    against the oracle."""
    X, y, _ = go.synthetic_problem(N, D, 4, seed=N)
    kern = go.make_kernel([kind], D, [dict(variance=0.9, lengthscale=0.8 * np.sqrt(D))])
    noise = 0.05
    inf = go.exact_inference(kern, X, noise, y[:, None])
    kg, dn = go.dense_gradients(kern, X, inf)
    hip.set_data(X, y)
    set_kernel(hip, kern, noise)
    lml, dvar, dlen, dnoise = hip.dense_lml_grad()
    # (a sum of N terms of order one that nearly cancels: measured against max(|LML|, N))
    assert abs(lml - inf["log_marginal"]) <= 1e-10 * max(abs(inf["log_marginal"]), float(N))
    gscale = max(abs(dn), abs(kg[0][0]), np.abs(kg[0][1]).max())
    assert abs(dvar[0] - kg[0][0]) <= 1e-8 * gscale
    assert np.abs(fold_dlen(kern, dlen)[0] - kg[0][1]).max() <= 1e-8 * gscale
    assert abs(dnoise - dn) <= 1e-8 * gscale
    hip.dense_posterior()
    Xs = np.vstack([np.random.default_rng(2).standard_normal((40, D)), X[:8]])
    mu_ref, var_ref = go.raw_predict(kern, Xs, X, inf["woodbury_vector"], inf["woodbury_inv"])
    jac_ref = go.predictive_gradients_mean(kern, Xs, X, inf["woodbury_vector"])[:, :, 0]
    mu, var = hip.predict(Xs)
    assert rel(mu, mu_ref[:, 0]) < 1e-8
    assert np.abs(var - var_ref[:, 0]).max() < 1e-8 * max(1.0, np.abs(var_ref).max())
    assert rel(hip.predict_grad(Xs), jac_ref) < 1e-8


def test_bitwise_reproducible_across_overlap_modes(hip):
    """Fixed-order reductions, no float atomics: the same evaluation gives the same bits on every repeat,
    whichever side-stream overlap policy schedules it."""
    X, y, Z = go.synthetic_problem(20000, 16, 512, seed=4)
    hip.set_data(X, y)
    ref = {}
    try:
        for i in range(24):
            s = 1.0 + 1e-3 * (i % 3)
            hip.set_option("overlap", (3, 0, 2, 1)[i % 4])
            hip.set_inducing(Z)
            hip.set_kernel(["RBF"], [s], np.full((1, 16), 4.0 * s), 0.01 * s)
            out = hip.sparse_lml_grad() if i % 2 else (hip.sparse_lml(),)
            key = (i % 3, len(out))
            if key in ref:
                for u, v in zip(ref[key], out):
                    assert np.array_equal(np.asarray(u), np.asarray(v)), (i, key)
            else:
                ref[key] = out
    finally:
        hip.set_option("overlap", 3)
