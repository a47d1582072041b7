"""Parity with the CPU oracle AT BASELINE.json's full sizes (SURVEY 8d gate: bound <= 1e-10 rel, posterior mean
on the N training rows <= 1e-8 rel), through the C ABI, in the library's default (automatic) statistics mode:
  Cm   N = 100 000, M = 2048, D = 32, RBF        (the metric's configuration; Gram form)
  C2   N = 100 000, M = 1024, D = 16, RBF        (stable form: its Kmm is the worse conditioned one)
  C3s  N = 125 000, M = 2048, D = 32, Matern-5/2 (one GPU's shard of config 3)
  C5c  N = 131 072, M = 4096, D = 64, RBF        (one row chunk of config 5), `precision` 64 and 32
One oracle inference per configuration (GPy's operation sequence in numpy/scipy on the host cores: tens of
seconds), the oracle's posterior mean / variance on the rows, and the oracle's gradients of the bound
(SparseGP._update_gradients: d/dvariance, d/dlengthscale, d/dnoise, dZ; the inducing-input gradient in the
oracle's reordered form -- one BLAS product instead of D passes over the N x M array, itself held to GPy's loop form in
tests/test_host_logic.py::test_oracle_gradients_x_gemm_form).  C2 adds a directional derivative of the ORACLE's bound
along a random direction of Z (two more oracle inferences) against the device's analytic dZ.
Inputs follow SURVEY 8(d): Z = X[perm[:M]].  Run last (module name sorts after test_full_size.py)."""
import time

import numpy as np
import pytest

from oracle import gp_oracle as go

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]


def synthetic(N, D, M, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    Z = X[rng.permutation(N)[:M]].copy()
    return X, y, Z


class _host_cores:
    def __enter__(self):
        from edrgp_amd.utils import effective_cpus
        from threadpoolctl import threadpool_limits
        self._l = threadpool_limits(limits=effective_cpus())
        return self

    def __exit__(self, *a):
        self._l.restore_original_limits()
        return False


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def device_run(hip, X, y, Z, kind, ell, noise, precision, n_var):
    D = X.shape[1]
    hip.set_option("stats_mode", 0)
    hip.set_option("chunk_rows", 0)
    hip.set_option("precision", precision)
    hip.set_data(X, y)
    hip.set_inducing(Z)
    hip.set_kernel([kind], [1.0], np.full((1, D), ell), noise)
    out = {"lml_only": hip.sparse_lml()}
    out["lml"], out["dvar"], out["dlen"], out["dnoise"], out["dZ"] = hip.sparse_lml_grad()
    out["mode"] = int(hip.get("MODE", (1,))[0])
    out["lml_post"] = hip.sparse_posterior()
    out["mu"], _ = hip.predict_train(want_mean=True, want_grad=False)
    out["mu_p"], out["var_p"] = hip.predict(X[:n_var])
    out["wv"] = hip.get("WV", (Z.shape[0],))
    return out


# name, N, M, D, kernel, precisions, directional derivative of the oracle's bound along Z, gate (bound, posterior mean)
# The fp64 BASELINE configurations are held to SURVEY 8(d)'s gate, 1e-10 / 1e-8.  The chunk of config 5 -- BASELINE's
# fp32 configuration, for which SURVEY asks for "the achieved tolerance" -- lands at 1.1e-10 / 2.1e-8 in fp64 and is
# gated at 3e-10 / 5e-8: its two statistics forms agree with each other to 7e-12 / 1.2e-8 (tools/cond_probe.py) and the
# oracle moves by 3e-15 / 5e-13 when the last bit of its inputs is perturbed (tools/oracle_floor.py), so the difference
# is the device's own and common to both forms -- the explicit Lm^-1 (cond(Lm) ~ 1e4 at M = 4096) where GPy solves
# with the triangular factor.
CONFIGS = [
    ("Cm", 100000, 2048, 32, "RBF", (64,), False, (1e-10, 1e-8)),
    ("C2", 100000, 1024, 16, "RBF", (64,), True, (1e-10, 1e-8)),
    ("C3s", 125000, 2048, 32, "Matern52", (64,), False, (1e-10, 1e-8)),
    ("C5c", 131072, 4096, 64, "RBF", (64, 32), False, (3e-10, 5e-8)),
]
# precision 32: Phi comes from the fp64 row pass (the bound is the fp64 one); K_nm and the gradient contraction are
# fp32 -- measured here against the oracle: 6.4e-3 / 2.8e-3 / 5.4e-2 for d/dvariance, d/dlengthscale, dZ (max-norm relative)
# and, after a change of the fp64 slab summation order that moves S2 by an ulp, 2.4e-3 / 1.1e-3 / 9.2e-2: the fp32
# contraction K_nm S2 cancels ~17 bits at this shape, so these figures are one draw of its rounding noise (which fp32
# values S2 rounds to), not constants -- the tolerances leave a factor of about two above the larger draw
TOL = {64: dict(dvar=1e-7, dlen=1e-7, dnoise=1e-7, dZ=1e-6), 32: dict(dvar=2e-2, dlen=2e-2, dnoise=1e-7, dZ=2e-1)}


@pytest.mark.parametrize("name,N,M,D,kind,precisions,fd_on_oracle,gate", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_full_size_against_oracle(hip, name, N, M, D, kind, precisions, fd_on_oracle, gate):
    X, y, Z = synthetic(N, D, M, seed=0)
    ell, noise = float(np.sqrt(D)), 0.01
    kern = go.make_kernel([kind], D, [dict(variance=1.0, lengthscale=ell)])
    n_var = 4096
    try:
        dev = {p: device_run(hip, X, y, Z, kind, ell, noise, p, n_var) for p in precisions}
    finally:
        hip.set_option("precision", 64)

    with _host_cores():
        t0 = time.perf_counter()
        inf = go.vardtc_inference(kern, X, Z, noise, y[:, None])
        t_inf = time.perf_counter() - t0
        ref = float(inf["log_marginal"][0, 0])
        wv = inf["woodbury_vector"]
        mu_ref = (inf["psi1"] @ wv)[:, 0]                  # Posterior._raw_predict mean: K(X*, Z) woodbury_vector
        mu4, var4 = go.raw_predict(kern, X[:n_var], Z, wv, inf["woodbury_inv"])
        t0 = time.perf_counter()
        kg, dn_ref, dZ_ref = go.sparse_gradients(kern, X, Z, inf, gemm_form=True)
        t_grad = time.perf_counter() - t0
        dvar_ref, dlen_ref = float(kg[0][0]), float(np.sum(kg[0][1]))
        for p in precisions:
            d = dev[p]
            print("\n[%s, precision %d] oracle inference %.1f s + gradients %.1f s; bound %.9f (device %.9f, mode %d); rel err bound "
                  "%.2e / %.2e / %.2e, mean %.2e, var %.2e, dvar %.2e, dlen %.2e, dnoise %.2e, dZ %.2e"
                  % (name, p, t_inf, t_grad, ref, d["lml"], d["mode"], abs(d["lml"] - ref) / abs(ref),
                     abs(d["lml_only"] - ref) / abs(ref), abs(d["lml_post"] - ref) / abs(ref), rel(d["mu"], mu_ref),
                     np.abs(d["var_p"] - var4[:, 0]).max(), abs(d["dvar"][0] - dvar_ref) / abs(dvar_ref),
                     abs(d["dlen"][0].sum() - dlen_ref) / abs(dlen_ref), abs(d["dnoise"] - dn_ref) / abs(dn_ref),
                     rel(d["dZ"], dZ_ref)))
            # SURVEY 8(d) gate (the fp32 row pass keeps the fp64 bound and posterior)
            for v in (d["lml"], d["lml_only"], d["lml_post"]):
                assert abs(v - ref) <= gate[0] * abs(ref), (p, v, ref)
            assert rel(d["mu"], mu_ref) <= gate[1]
            assert np.abs(d["mu_p"] - mu4[:, 0]).max() <= gate[1] * np.abs(mu_ref).max()
            assert np.abs(d["var_p"] - var4[:, 0]).max() <= 1e-7
            assert np.abs(d["wv"] - wv[:, 0]).max() <= 1e-6 * np.abs(wv).max()
            # gradients of the bound against SparseGP._update_gradients
            tol = TOL[p]
            assert abs(d["dvar"][0] - dvar_ref) <= tol["dvar"] * abs(dvar_ref), (p, d["dvar"], dvar_ref)
            assert abs(d["dlen"][0].sum() - dlen_ref) <= tol["dlen"] * abs(dlen_ref), (p, d["dlen"][0].sum(), dlen_ref)
            assert abs(d["dnoise"] - dn_ref) <= tol["dnoise"] * abs(dn_ref), (p, d["dnoise"], dn_ref)
            assert rel(d["dZ"], dZ_ref) <= tol["dZ"], (p, rel(d["dZ"], dZ_ref))
        del inf, kg, dZ_ref

        if fd_on_oracle:
            d = dev[64]
            V = np.random.default_rng(5).standard_normal(Z.shape)
            an = float(np.sum(d["dZ"] * V))
            t = 1e-4
            up = go.vardtc_inference(kern, X, Z + t * V, noise, y[:, None])["log_marginal"][0, 0]
            dn = go.vardtc_inference(kern, X, Z - t * V, noise, y[:, None])["log_marginal"][0, 0]
            fd = float(up - dn) / (2 * t)
            print("[%s] dZ . V analytic %.6e, central difference of the oracle's bound %.6e" % (name, an, fd))
            assert abs(fd - an) <= 1e-3 * abs(an), (fd, an)


def test_dense_exact_gp_16k_against_oracle(hip):
    """the exact GP (edrgp/gp_model/regression.py:10-77) at N = 16 384 -- 32x config 1's N, the blocked
    factorisation beyond the fused range (n > 3072: two launches per step, recursive-doubling inverse, K^-1 = L^-T L^-1
    as a Gram product).  Bound, gradients and posterior against the oracle's ExactGaussianInference (one N^3 Cholesky +
    dpotri on the host cores), and the bound of a 2 000-row subsample against scikit-learn's independent exact GP."""
    N, D, noise = 16384, 8, 0.05
    rng = np.random.default_rng(16)
    X = rng.standard_normal((N, D))
    y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    ell = 0.8 * float(np.sqrt(D))
    kern = go.make_kernel(["RBF"], D, [dict(variance=0.9, lengthscale=ell)])
    hip.set_data(X, y)
    hip.set_kernel(["RBF"], [0.9], np.full((1, D), ell), noise)
    t0 = time.perf_counter()
    lml, dvar, dlen, dnoise = hip.dense_lml_grad()
    t_dev = time.perf_counter() - t0
    hip.dense_posterior()
    Xs = np.vstack([rng.standard_normal((64, D)), X[:16]])
    mu, var = hip.predict(Xs)
    jac = hip.predict_grad(Xs)
    with _host_cores():
        t0 = time.perf_counter()
        inf = go.exact_inference(kern, X, noise, y[:, None])
        kg, dn = go.dense_gradients(kern, X, inf)
        t_cpu = time.perf_counter() - t0
        ref = float(inf["log_marginal"])
        mu_ref, var_ref = go.raw_predict(kern, Xs, X, inf["woodbury_vector"], inf["woodbury_inv"])
        jac_ref = go.predictive_gradients_mean(kern, Xs, X, inf["woodbury_vector"])[:, :, 0]
    print("\n[dense N=%d] device bound+gradients %.2f s, oracle %.1f s; bound %.6f (device %.6f): rel %.2e; dvar %.2e dlen %.2e "
          "dnoise %.2e; mean %.2e var %.2e jac %.2e"
          % (N, t_dev, t_cpu, ref, lml, abs(lml - ref) / max(abs(ref), N), abs(dvar[0] - kg[0][0]) / abs(kg[0][0]),
             abs(dlen[0].sum() - kg[0][1][0]) / abs(kg[0][1][0]), abs(dnoise - dn) / abs(dn), rel(mu, mu_ref[:, 0]),
             np.abs(var - var_ref[:, 0]).max(), rel(jac, jac_ref)))
    assert abs(lml - ref) <= 1e-10 * max(abs(ref), float(N))
    gscale = max(abs(dn), abs(kg[0][0]), abs(float(kg[0][1][0])))
    assert abs(dvar[0] - kg[0][0]) <= 1e-7 * gscale
    assert abs(dlen[0].sum() - kg[0][1][0]) <= 1e-7 * gscale
    assert abs(dnoise - dn) <= 1e-7 * gscale
    assert rel(mu, mu_ref[:, 0]) <= 1e-8
    assert np.abs(var - var_ref[:, 0]).max() <= 1e-7
    assert rel(jac, jac_ref) <= 1e-7
    del inf
    # an independent implementation on a subsample: sklearn's exact-GP log marginal likelihood
    from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel as C, WhiteKernel
    n = 2000
    hip.set_data(X[:n], y[:n])
    hip.set_kernel(["RBF"], [0.9], np.full((1, D), ell), noise)
    lml_n = hip.dense_lml_grad()[0]
    sk = SkGPR(C(0.9) * RBF(ell) + WhiteKernel(noise), alpha=1e-8, optimizer=None).fit(X[:n], y[:n])
    sk_lml = float(sk.log_marginal_likelihood(sk.kernel_.theta))
    assert abs(lml_n - sk_lml) <= 1e-9 * max(abs(sk_lml), float(n)), (lml_n, sk_lml)
