"""CPU oracle for the edr-gp GP inference path  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped path (``edr-gp_amd/``) never imports anything from
``oracle/`` and fails loudly when its HIP library is missing.

What it restates
----------------
The reference (``/root/reference``, neuro-ml/edr-gp 0.2.6) does no GP
arithmetic itself: ``edrgp/gp_model/regression.py:76-77`` and ``:153-157``
construct ``GPy.models.GPRegression`` / ``SparseGPRegression`` and
``edrgp/gp_model/base.py:69,187,206,222`` call ``optimize``, ``predict`` and
``predictive_gradients`` on them.  GPy (``gpy>=1.8.4``,
``requirements.txt:3``, unpinned; paramz transitively) is a third-party
dependency that is NOT under ``/root/reference`` and NOT installed in the build
container, so the algorithm below is a restatement of GPy's published
algorithm (GPy 1.9/1.10 sources: ``GPy/kern/src/stationary.py``, ``rbf.py``,
``GPy/inference/latent_function_inference/{var_dtc,exact_gaussian_inference,
posterior}.py``, ``GPy/core/{gp,sparse_gp}.py``, ``GPy/util/{linalg,
normalizer}.py``, ``paramz/transformations.py``), issuing the same LAPACK/BLAS
calls in the same order, with the same jitters (1e-8), clips and array shapes.

Pinning status
--------------
The reference holds NO golden vectors or known-answer values for this path
(``edrgp/tests/test_edr.py:33-117`` are five loose end-to-end invariants), and
GPy cannot be run here, so parity with GPy itself is **unpinned**.  The oracle
is pinned instead against independent known answers that exist in this
container (``tests/test_golden.py``): scikit-learn's exact GP
log-marginal-likelihood / posterior (|d| ~ 1e-13), the VarDTC bound with Z = X
versus the exact LML, the variational lower-bound property, central finite
differences of the bound for every hyper-parameter and for Z, and finite
differences of the posterior mean for ``predictive_gradients``.

Every function cites the reference call site that reaches it and the GPy unit
it restates.
"""
from __future__ import annotations

import numpy as np
from scipy import linalg
from scipy.linalg import lapack

LOG_2_PI = np.log(2.0 * np.pi)
CONST_JITTER = 1e-8          # GPy VarDTC.const_jitter; ExactGaussianInference adds 1e-8 too

KINDS = ("RBF", "Matern32", "Matern52")


# --------------------------------------------------------------------------
# GPy.util.linalg
# --------------------------------------------------------------------------
def jitchol(A, maxtries=5):
    """GPy.util.linalg.jitchol: dpotrf, on failure retry with jitter
    mean(diag)*1e-6*10^k, k < maxtries.  Reached from VarDTC.inference /
    pdinv on every ``fit`` iteration (gp_model/base.py:69)."""
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return L
    diagA = np.diag(A)
    if np.any(diagA <= 0.0):
        raise linalg.LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        try:
            return linalg.cholesky(A + np.eye(A.shape[0]) * jitter, lower=True)
        except linalg.LinAlgError:
            jitter *= 10
        finally:
            num_tries += 1
    raise linalg.LinAlgError("not positive definite, even with jitter.")


def dtrtrs(A, B, lower=1, trans=0):
    """GPy.util.linalg.dtrtrs -> LAPACK dtrtrs (solve op(A) X = B, A triangular)."""
    A = np.asfortranarray(A)
    return lapack.dtrtrs(A, B, lower=lower, trans=trans, unitdiag=0)[0]


def dpotrs(L, B, lower=1):
    """GPy.util.linalg.dpotrs -> LAPACK dpotrs."""
    return lapack.dpotrs(np.asfortranarray(L), B, lower=lower)[0]


def dpotri(L, lower=1):
    """GPy.util.linalg.dpotri -> LAPACK dpotri; returns the symmetrised inverse."""
    Ai = lapack.dpotri(np.asfortranarray(L), lower=lower)[0]
    Ai = np.tril(Ai) + np.tril(Ai, -1).T if lower else np.triu(Ai) + np.triu(Ai, 1).T
    return Ai


def tdot(A):
    """GPy.util.linalg.tdot: A A^T (dsyrk upstream)."""
    return A @ A.T


def backsub_both_sides(L, X, transpose="left"):
    """GPy.util.linalg.backsub_both_sides: L^-T X L^-1 ('left', the default)
    or L^-1 X L^-T, X symmetric, L lower Cholesky."""
    if transpose == "left":
        tmp = dtrtrs(L, X, lower=1, trans=1)
        return dtrtrs(L, tmp.T, lower=1, trans=1).T
    tmp = dtrtrs(L, X, lower=1, trans=0)
    return dtrtrs(L, tmp.T, lower=1, trans=0).T


def pdinv(A):
    """GPy.util.linalg.pdinv: (Ai, L, Li, logdet)."""
    L = jitchol(A)
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    Li = dtrtrs(L, np.eye(A.shape[0]), lower=1)
    Ai = dpotri(L, lower=1)
    return Ai, L, Li, logdet


# --------------------------------------------------------------------------
# GPy.util.normalizer.Standardize   (normalizer=True, regression.py:48,124)
# --------------------------------------------------------------------------
class Standardize:
    def scale_by(self, Y):
        Y = np.asarray(Y, dtype=np.float64)
        self.mean = Y.mean(0)
        self.std = Y.std(0)            # ddof = 0

    def normalize(self, Y):
        return (Y - self.mean) / self.std

    def inverse_mean(self, X):
        return X * self.std + self.mean

    def inverse_variance(self, var):
        return var * self.std ** 2


# --------------------------------------------------------------------------
# paramz.transformations.Logexp  (positive parameters under optimisation)
# --------------------------------------------------------------------------
_LIM_VAL = 36.0
_EPS = np.finfo(np.float64).resolution


class Logexp:
    @staticmethod
    def f(x):
        x = np.asarray(x, dtype=np.float64)
        return np.where(x > _LIM_VAL, x, np.log1p(np.exp(np.clip(x, -np.log(np.finfo(np.float64).max), _LIM_VAL))))

    @staticmethod
    def finv(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f > _LIM_VAL, f, np.log(np.expm1(f)))

    @staticmethod
    def gradfactor(f, df):
        f = np.asarray(f, dtype=np.float64)
        return df * np.where(f > _LIM_VAL, 1.0, -np.expm1(-f))


# --------------------------------------------------------------------------
# GPy.kern.src.stationary.Stationary / rbf.RBF / Matern32 / Matern52
# (built by _make_kernel, gp_model/base.py:111-147; default RBF, :120-122)
# --------------------------------------------------------------------------
class Stationary:
    def __init__(self, kind, input_dim, variance=1.0, lengthscale=None, ARD=False):
        assert kind in KINDS, kind
        self.kind = kind
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        self.variance = float(variance)
        if lengthscale is None:
            lengthscale = np.ones(self.input_dim if self.ARD else 1)
        lengthscale = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
        if self.ARD and lengthscale.size == 1:
            lengthscale = np.ones(self.input_dim) * lengthscale
        if not self.ARD:
            assert lengthscale.size == 1, "Only 1 lengthscale needed for non-ARD kernel"
        self.lengthscale = lengthscale.copy()

    # -- distances ---------------------------------------------------------
    def _unscaled_dist(self, X, X2=None):
        if X2 is None:
            Xsq = np.sum(np.square(X), 1)
            r2 = -2.0 * tdot(X) + (Xsq[:, None] + Xsq[None, :])
            r2[np.diag_indices_from(r2)] = 0.0      # util.diag.view(r2)[:,] = 0.
            r2 = np.clip(r2, 0, np.inf)
            return np.sqrt(r2)
        X1sq = np.sum(np.square(X), 1)
        X2sq = np.sum(np.square(X2), 1)
        r2 = -2.0 * np.dot(X, X2.T) + (X1sq[:, None] + X2sq[None, :])
        r2 = np.clip(r2, 0, np.inf)
        return np.sqrt(r2)

    def _scaled_dist(self, X, X2=None):
        if self.ARD:
            if X2 is not None:
                X2 = X2 / self.lengthscale
            return self._unscaled_dist(X / self.lengthscale, X2)
        return self._unscaled_dist(X, X2) / self.lengthscale

    def _inv_dist(self, X, X2=None):
        dist = self._scaled_dist(X, X2).copy()
        return 1.0 / np.where(dist != 0.0, dist, np.inf)

    # -- covariance as a function of r --------------------------------------
    def K_of_r(self, r):
        v = self.variance
        if self.kind == "RBF":
            return v * np.exp(-0.5 * r ** 2)
        if self.kind == "Matern32":
            return v * (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
        return v * (1 + np.sqrt(5.0) * r + 5.0 / 3 * r ** 2) * np.exp(-np.sqrt(5.0) * r)

    def dK_dr(self, r):
        v = self.variance
        if self.kind == "RBF":
            return -r * self.K_of_r(r)
        if self.kind == "Matern32":
            return -3.0 * v * r * np.exp(-np.sqrt(3.0) * r)
        return v * (10.0 / 3 * r - 5.0 * r - 5.0 * np.sqrt(5.0) / 3 * r ** 2) * np.exp(-np.sqrt(5.0) * r)

    def K(self, X, X2=None):
        return self.K_of_r(self._scaled_dist(X, X2))

    def Kdiag(self, X):
        ret = np.empty(X.shape[0])
        ret[:] = self.variance
        return ret

    def dK_dr_via_X(self, X, X2):
        return self.dK_dr(self._scaled_dist(X, X2))

    # -- gradients -----------------------------------------------------------
    def gradients_X(self, dL_dK, X, X2=None, gemm_form=False):
        """Stationary._gradients_X_pure (upstream also has a Cython/OpenMP twin).

        ``gemm_form``: the same sums reordered, sum_n tmp[m, n] (x[m, q] - x2[n, q]) =
        x[m, q] sum_n tmp[m, n] - (tmp @ X2)[m, q] -- one BLAS product instead of input_dim
        passes over the M x N array with a temporary each.  Not GPy's operation order; it
        exists so that the full-size parity tests (N = 1e5, M = 2048..4096, D up to 64: the
        loop form takes minutes there) can check dZ, and is itself held to the loop form in
        tests/test_host_logic.py::test_oracle_gradients_x_gemm_form."""
        invdist = self._inv_dist(X, X2)
        dL_dr = self.dK_dr_via_X(X, X2) * dL_dK
        tmp = invdist * dL_dr
        if X2 is None:
            tmp = tmp + tmp.T
            X2 = X
        if gemm_form:
            grad = X * np.sum(tmp, axis=1)[:, None] - tmp @ X2
            return grad / self.lengthscale ** 2
        grad = np.empty(X.shape, dtype=np.float64)
        for q in range(self.input_dim):
            np.sum(tmp * (X[:, q][:, None] - X2[:, q][None, :]), axis=1, out=grad[:, q])
        return grad / self.lengthscale ** 2

    def param_gradients_full(self, dL_dK, X, X2=None):
        """Stationary.update_gradients_full -> (d/dvariance, d/dlengthscale)."""
        dvar = np.sum(self.K(X, X2) * dL_dK) / self.variance
        dL_dr = self.dK_dr_via_X(X, X2) * dL_dK
        if self.ARD:
            tmp = dL_dr * self._inv_dist(X, X2)
            if X2 is None:
                X2 = X
            dl = -np.array([np.sum(tmp * np.square(X[:, q:q + 1] - X2[:, q:q + 1].T))
                            for q in range(self.input_dim)]) / self.lengthscale ** 3
        else:
            r = self._scaled_dist(X, X2)
            dl = np.atleast_1d(-np.sum(dL_dr * r) / self.lengthscale)
        return dvar, dl

    def param_gradients_diag(self, dL_dKdiag, X):
        """Stationary.update_gradients_diag."""
        return np.sum(dL_dKdiag), np.zeros_like(self.lengthscale)

    def n_params(self):
        return 1 + self.lengthscale.size


class Add:
    """GPy.kern.Add of stationary parts (``np.sum([...])`` at gp_model/base.py:146)."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.input_dim = self.parts[0].input_dim

    def K(self, X, X2=None):
        return sum(p.K(X, X2) for p in self.parts)

    def Kdiag(self, X):
        return sum(p.Kdiag(X) for p in self.parts)

    def gradients_X(self, dL_dK, X, X2=None, gemm_form=False):
        return sum(p.gradients_X(dL_dK, X, X2, gemm_form=gemm_form) for p in self.parts)

    def param_gradients_full(self, dL_dK, X, X2=None):
        return [p.param_gradients_full(dL_dK, X, X2) for p in self.parts]

    def param_gradients_diag(self, dL_dKdiag, X):
        return [p.param_gradients_diag(dL_dKdiag, X) for p in self.parts]


def as_add(kern):
    return kern if isinstance(kern, Add) else Add([kern])


def make_kernel(kinds, input_dim, options=None):
    """_BaseGP._make_kernel (gp_model/base.py:111-147) for the stationary family."""
    if kinds is None:
        kinds = ["RBF"]
    if isinstance(kinds, str):
        kinds = [kinds]
    if isinstance(options, dict):
        options = [options]
    if options is None:
        options = [{}] * len(kinds)
    if len(options) != len(kinds):
        raise ValueError
    return Add([Stationary(k, input_dim, **o) for k, o in zip(kinds, options)])


# --------------------------------------------------------------------------
# GPy.inference.latent_function_inference.var_dtc.VarDTC.inference
# (SparseGPRegression built at regression.py:153-157; evaluated once per
#  optimiser iteration from gp_model/base.py:69)
# --------------------------------------------------------------------------
def vardtc_inference(kern, X, Z, noise_var, Y):
    """Returns dict with the Titsias bound (shape (1,1), as the reference's
    test reads it: tests/test_edr.py:50), posterior factors and dL_d* terms.
    Same operation sequence as upstream: Kmm+jitter -> jitchol; psi1 = K(X,Z);
    dtrtrs; tdot; jitchol(B); dtrtrs x2; ..."""
    kern = as_add(kern)
    num_data, output_dim = Y.shape
    num_inducing = Z.shape[0]
    precision = 1.0 / np.fmax(np.atleast_1d(np.float64(noise_var)), CONST_JITTER)
    precision = precision[:, None]                                   # (1,1)
    VVT_factor = precision * Y
    trYYT = np.einsum("ij,ij->", Y, Y)

    Kmm = kern.K(Z).copy()
    Kmm[np.diag_indices_from(Kmm)] += CONST_JITTER
    Lm = jitchol(Kmm)

    psi0 = kern.Kdiag(X)
    psi1 = kern.K(X, Z)
    tmp = psi1 * np.sqrt(precision)
    tmp = dtrtrs(Lm, tmp.T, lower=1)
    A = tdot(tmp)

    B = np.eye(num_inducing) + A
    LB = jitchol(B)

    tmp = dtrtrs(Lm, psi1.T, lower=1, trans=0)
    _LBi_Lmi_psi1 = dtrtrs(LB, tmp, lower=1, trans=0)
    _LBi_Lmi_psi1Vf = np.dot(_LBi_Lmi_psi1, VVT_factor)
    tmp = dtrtrs(LB, _LBi_Lmi_psi1Vf, lower=1, trans=1)
    Cpsi1Vf = dtrtrs(Lm, tmp, lower=1, trans=1)

    delit = tdot(_LBi_Lmi_psi1Vf)
    data_fit = np.trace(delit)
    DBi_plus_BiPBi = backsub_both_sides(LB, output_dim * np.eye(num_inducing) + delit)
    delit = -0.5 * DBi_plus_BiPBi
    delit += -0.5 * B * output_dim
    delit += output_dim * np.eye(num_inducing)
    dL_dKmm = backsub_both_sides(Lm, delit)

    # _compute_dL_dpsi
    dL_dpsi0 = -0.5 * output_dim * (precision * np.ones([num_data, 1])).flatten()
    dL_dpsi1 = np.dot(VVT_factor, Cpsi1Vf.T)
    dL_dpsi2_beta = 0.5 * backsub_both_sides(Lm, output_dim * np.eye(num_inducing) - DBi_plus_BiPBi)
    dL_dpsi2 = precision * dL_dpsi2_beta
    dL_dpsi1 += 2.0 * np.dot(psi1, dL_dpsi2)

    # _compute_log_marginal_likelihood
    beta = precision
    lik_1 = -0.5 * num_data * output_dim * (LOG_2_PI - np.log(beta)) - 0.5 * beta * trYYT
    lik_2 = -0.5 * output_dim * (np.sum(beta * psi0) - np.trace(A))
    lik_3 = -output_dim * (np.sum(np.log(np.diag(LB))))
    lik_4 = 0.5 * data_fit
    log_marginal = lik_1 + lik_2 + lik_3 + lik_4                      # (1,1)

    # _compute_dL_dR  (homoscedastic)
    dL_dR = -0.5 * num_data * output_dim * beta + 0.5 * trYYT * beta ** 2
    dL_dR += 0.5 * output_dim * (psi0.sum() * beta ** 2 - np.trace(A) * beta)
    dL_dR += beta * (0.5 * np.sum(A * DBi_plus_BiPBi) - data_fit)
    dL_dthetaL = float(np.sum(dL_dR))      # Gaussian.exact_inference_gradients = sum

    woodbury_vector = Cpsi1Vf
    Bi = -dpotri(LB, lower=1)
    Bi[np.diag_indices_from(Bi)] += 1.0
    woodbury_inv = backsub_both_sides(Lm, Bi)

    return dict(log_marginal=log_marginal, woodbury_vector=woodbury_vector,
                woodbury_inv=woodbury_inv, Lm=Lm, LB=LB, A=A, Kmm=Kmm, psi1=psi1,
                dL_dKmm=dL_dKmm, dL_dKnm=dL_dpsi1, dL_dKdiag=dL_dpsi0,
                dL_dthetaL=dL_dthetaL, data_fit=data_fit)


def sparse_gradients(kern, X, Z, inf, gemm_form=False):
    """SparseGP._update_gradients (GPy/core/sparse_gp.py): kernel, noise and
    inducing-input gradients of the bound.  Returns (kern_grads, dnoise, dZ)
    with kern_grads = [(dvariance, dlengthscale), ...] per additive part.
    ``gemm_form``: see Stationary.gradients_X (full-size tests only)."""
    kern = as_add(kern)
    g_diag = kern.param_gradients_diag(inf["dL_dKdiag"], X)
    g_nm = kern.param_gradients_full(inf["dL_dKnm"], X, Z)
    g_mm = kern.param_gradients_full(inf["dL_dKmm"], Z, None)
    kern_grads = [(a[0] + b[0] + c[0], a[1] + b[1] + c[1]) for a, b, c in zip(g_diag, g_nm, g_mm)]
    dZ = kern.gradients_X(inf["dL_dKmm"], Z, gemm_form=gemm_form)
    dZ = dZ + kern.gradients_X(inf["dL_dKnm"].T, Z, X, gemm_form=gemm_form)
    return kern_grads, inf["dL_dthetaL"], dZ


# --------------------------------------------------------------------------
# GPy.inference.latent_function_inference.exact_gaussian_inference
# (GPRegression built at regression.py:76-77)
# --------------------------------------------------------------------------
def exact_inference(kern, X, noise_var, Y):
    kern = as_add(kern)
    K = kern.K(X)
    Ky = K.copy()
    Ky[np.diag_indices_from(Ky)] += noise_var + 1e-8
    Wi, LW, LWi, W_logdet = pdinv(Ky)
    alpha = dpotrs(LW, Y, lower=1)
    log_marginal = 0.5 * (-Y.size * LOG_2_PI - Y.shape[1] * W_logdet - np.sum(alpha * Y))
    dL_dK = 0.5 * (tdot(alpha) - Y.shape[1] * Wi)
    dL_dthetaL = float(np.sum(np.diag(dL_dK)))
    return dict(log_marginal=float(log_marginal), woodbury_vector=alpha, woodbury_chol=LW,
                woodbury_inv=Wi, K=K, dL_dK=dL_dK, dL_dthetaL=dL_dthetaL)


def dense_gradients(kern, X, inf):
    """GP.parameters_changed: kern.update_gradients_full(dL_dK, X)."""
    kern = as_add(kern)
    return kern.param_gradients_full(inf["dL_dK"], X, None), inf["dL_dthetaL"]


# --------------------------------------------------------------------------
# GP classification: GPy.models.GPClassification = GP + Bernoulli(probit) + EP
# (edrgp/gp_model/classification.py:63-64).  Restated: likelihoods/bernoulli.py
# (moments_match_ep, predictive_mean), util/univariate_Gaussian.py,
# inference/latent_function_inference/expectation_propagation.py (class EP:
# epsilon=1e-6, eta=1, delta=1, sequential site updates in a random order, full
# recomputation of the posterior after every sweep, ep_mode="alt": the sites are
# fitted ONCE per optimize() call -- on_optimization_start resets them -- and held
# fixed while the kernel hyper-parameters are optimised).
# Parity status: unpinned, as for the rest of the GPy surface (GPy is absent); the
# fixed point is checked by its own consistency conditions in tests/test_classification.py.
# --------------------------------------------------------------------------
from scipy import special as _special

_EPS = np.finfo(float).eps


def log_cdf_normal(z):
    return _special.log_ndtr(z)


def deriv_log_cdf_normal(z):
    """phi(z) / Phi(z) (GPy util.univariate_Gaussian.derivLogCdfNormal), stable in the lower tail."""
    return np.exp(-0.5 * z * z - 0.5 * LOG_2_PI - _special.log_ndtr(z))


def bernoulli_moments_match_ep(y, tau_cav, v_cav):
    """Bernoulli.moments_match_ep with the probit link: zeroth / first / second moment of the tilted
    distribution Phi(sign f) N(f | v/tau, 1/tau).  y in {1} -> +1, {0, -1} -> -1."""
    sign = 1.0 if y == 1 else -1.0
    z = sign * v_cav / np.sqrt(tau_cav ** 2 + tau_cav)
    phi_div_Phi = deriv_log_cdf_normal(z)
    log_Z_hat = log_cdf_normal(z)
    mu_hat = v_cav / tau_cav + sign * phi_div_Phi / np.sqrt(tau_cav ** 2 + tau_cav)
    sigma2_hat = 1.0 / tau_cav - (phi_div_Phi / (tau_cav ** 2 + tau_cav)) * (z + phi_div_Phi)
    return np.exp(log_Z_hat), mu_hat, sigma2_hat


def ep_recompute(K, tau, v):
    """posteriorParams._recompute: Sigma, mu and chol(B), B = I + S^1/2 K S^1/2."""
    n = tau.size
    root = np.sqrt(tau)
    SrK = root[:, None] * K
    B = np.eye(n) + SrK * root[None, :]
    L = jitchol(B)
    V = dtrtrs(L, SrK, lower=1)
    Sigma = K - np.dot(V.T, V)
    aux = dpotrs(L, root * np.dot(K, v), lower=1)
    alpha = v - root * aux
    mu = np.dot(K, alpha)
    return mu, Sigma, L


def expectation_propagation(K, y, order_fn=None, epsilon=1e-6, eta=1.0, delta=1.0, max_iters=10 ** 6):
    """EP.expectation_propagation.  ``order_fn(n)`` yields the site order of a sweep (upstream:
    np.random.permutation(n)).  Returns the site parameters, cavity parameters, the tilted moments of the
    last sweep, log_Z_tilde and the number of sweeps."""
    n = y.size
    if order_fn is None:
        order_fn = np.random.permutation
    tau, v = np.zeros(n), np.zeros(n)
    tau_cav, v_cav = np.zeros(n), np.zeros(n)
    Z_hat, mu_hat, s2_hat = np.zeros(n), np.zeros(n), np.zeros(n)
    mu, Sigma = np.zeros(n), K.copy()
    upd = np.empty_like(Sigma)
    tau_old = v_old = None
    it = 0
    stop = False
    while not stop and it < max_iters:
        for i in order_fn(n):
            tau_cav[i] = 1.0 / Sigma[i, i] - eta * tau[i]
            v_cav[i] = mu[i] / Sigma[i, i] - eta * v[i]
            Z_hat[i], mu_hat[i], s2_hat[i] = bernoulli_moments_match_ep(y[i], tau_cav[i], v_cav[i])
            d_tau = delta / eta * (1.0 / s2_hat[i] - 1.0 / Sigma[i, i])
            d_v = delta / eta * (mu_hat[i] / s2_hat[i] - mu[i] / Sigma[i, i])
            prev = tau[i]
            tau[i] += d_tau
            if tau[i] < _EPS:
                tau[i] = _EPS
                d_tau = tau[i] - prev
            v[i] += d_v
            ci = d_tau / (1.0 + d_tau * Sigma[i, i])
            s = Sigma[:, i].copy()
            np.multiply(s[:, None], (ci * s)[None, :], out=upd)    # upstream: DSYR on the lower triangle + symmetrify
            Sigma -= upd
            mu = np.dot(Sigma, v)
        mu, Sigma, _ = ep_recompute(K, tau, v)
        if it > 0:
            stop = (np.mean(np.square(tau - tau_old)) < epsilon) and (np.mean(np.square(v - v_old)) < epsilon)
        tau_old, v_old = tau.copy(), v.copy()
        it += 1
    log_Z_tilde = np.sum(np.log(Z_hat) + 0.5 * LOG_2_PI + 0.5 * np.log(1.0 + tau / tau_cav)
                         - 0.5 * (v ** 2) / (tau_cav + tau)
                         + 0.5 * (v_cav * ((tau / tau_cav) * v_cav - 2.0 * v) / (tau_cav + tau)))
    return dict(tau=tau, v=v, tau_cav=tau_cav, v_cav=v_cav, Z_hat=Z_hat, mu_hat=mu_hat, sigma2_hat=s2_hat,
                log_Z_tilde=float(log_Z_tilde), sweeps=it)


def ep_inference(kern, X, sites):
    """EP._inference / _ep_marginal for FIXED site parameters (what every objective evaluation of one optimize()
    call is in ep_mode "alt"): log marginal, woodbury vector / inverse, dL_dK."""
    kern = as_add(kern)
    K = kern.K(X)
    tau, v = sites["tau"], sites["v"]
    mu, Sigma, L = ep_recompute(K, tau, v)
    n = tau.size
    B_logdet = np.sum(2.0 * np.log(np.diag(L)))
    log_marginal = 0.5 * (-n * LOG_2_PI - B_logdet + np.sum(v * np.dot(Sigma, v))) + sites["log_Z_tilde"]
    root = np.sqrt(tau)
    SrK = root[:, None] * K
    aux = dpotrs(L, np.dot(SrK, v), lower=1)
    alpha = (v - root * aux)[:, None]
    LWi = dtrtrs(L, np.diag(root), lower=1)
    Wi = np.dot(LWi.T, LWi)
    dL_dK = 0.5 * (tdot(alpha) - Wi)
    return dict(log_marginal=float(log_marginal), woodbury_vector=alpha, woodbury_inv=Wi, K=K, dL_dK=dL_dK,
                mu=mu, Sigma_diag=np.diag(Sigma).copy())


# --------------------------------------------------------------------------
# Sparse GP classification: GPy.models.SparseGPClassification = SparseGP + Bernoulli(probit) + EPDTC
# (edrgp/gp_model/classification.py:147-149).  Restated: expectation_propagation.py (class EPDTC and
# posteriorParamsDTC: the posterior lives on the DTC covariance Sigma = K_nm (K_mm + K_mn T K_nm)^-1 K_mn; a site
# update refactors LLT = K_mm + K_mn T K_nm after its rank-one change; after every sweep everything is recomputed;
# the stop rule and log Z_tilde are EPBase's, shared with the dense EP above) followed by VarDTC.inference with the
# pseudo-targets mu_tilde = v / tau, the heteroscedastic precision tau and Z_tilde = log Z_tilde
# (var_dtc.py, het_noise branches of _compute_dL_dpsi and _compute_log_marginal_likelihood).  As upstream, Lm is
# jitchol(kern.K(Z)) WITHOUT the 1e-8 const_jitter (EPDTC.inference hands VarDTC its own Lm).
# Parity status: unpinned (GPy absent); tests/test_classification.py checks the fixed point's own consistency
# conditions and the gradients against finite differences of this bound.
# --------------------------------------------------------------------------
def epdtc_recompute(LLT0, Kmn, tau, v):
    """posteriorParamsDTC._recompute (mu and the diagonal of Sigma = V^T V without forming the N x N matrix)."""
    LLT = LLT0 + np.dot(Kmn * tau[None, :], Kmn.T)
    L = jitchol(LLT)
    V = dtrtrs(L, Kmn, lower=1)
    mu = np.dot(V.T, np.dot(V, v))
    return mu, np.sum(V * V, 0), LLT


def expectation_propagation_dtc(Kmm, Kmn, y, order_fn=None, epsilon=1e-6, eta=1.0, delta=1.0, max_iters=10 ** 6):
    """EPDTC.expectation_propagation (_init_approximations with no previous sites, _local_updates with
    posteriorParamsDTC._update_rank1, _recompute + the eps floor on Sigma_diag after every sweep)."""
    n = y.size
    if order_fn is None:
        order_fn = np.random.permutation
    LLT0 = Kmm.copy()
    Lm = jitchol(LLT0)
    Vm = dtrtrs(Lm, Kmn, lower=1)
    Sigma_diag = np.sum(Vm * Vm, 0) + 1e-8
    LLT = LLT0.copy()
    mu = np.zeros(n)
    tau, v = np.zeros(n), np.zeros(n)
    tau_cav, v_cav = np.zeros(n), np.zeros(n)
    Z_hat, mu_hat, s2_hat = np.zeros(n), np.zeros(n), np.zeros(n)
    tau_old = v_old = None
    it = 0
    stop = False
    while not stop and it < max_iters:
        for i in order_fn(n):
            tau_cav[i] = 1.0 / Sigma_diag[i] - eta * tau[i]
            v_cav[i] = mu[i] / Sigma_diag[i] - eta * v[i]
            Z_hat[i], mu_hat[i], s2_hat[i] = bernoulli_moments_match_ep(y[i], tau_cav[i], v_cav[i])
            d_tau = delta / eta * (1.0 / s2_hat[i] - 1.0 / Sigma_diag[i])
            d_v = delta / eta * (mu_hat[i] / s2_hat[i] - mu[i] / Sigma_diag[i])
            prev = tau[i]
            tau[i] += d_tau
            if tau[i] < _EPS:
                tau[i] = _EPS
                d_tau = tau[i] - prev
            v[i] += d_v
            # posteriorParamsDTC._update_rank1
            k = Kmn[:, i].copy()
            LLT += d_tau * np.outer(k, k)
            L = jitchol(LLT)
            V = dtrtrs(L, Kmn, lower=1)
            Sigma_diag = np.maximum(np.sum(V * V, 0), _EPS)
            si = np.dot(V.T, V[:, i])
            mu = mu + (d_v - d_tau * mu[i]) * si
        mu, Sigma_diag, LLT = epdtc_recompute(LLT0, Kmn, tau, v)
        Sigma_diag = np.maximum(Sigma_diag, _EPS)
        if it > 0:
            stop = (np.mean(np.square(tau - tau_old)) < epsilon) and (np.mean(np.square(v - v_old)) < epsilon)
        tau_old, v_old = tau.copy(), v.copy()
        it += 1
    log_Z_tilde = np.sum(np.log(Z_hat) + 0.5 * LOG_2_PI + 0.5 * np.log(1.0 + tau / tau_cav)
                         - 0.5 * (v ** 2) / (tau_cav + tau)
                         + 0.5 * (v_cav * ((tau / tau_cav) * v_cav - 2.0 * v) / (tau_cav + tau)))
    return dict(tau=tau, v=v, tau_cav=tau_cav, v_cav=v_cav, Z_hat=Z_hat, mu_hat=mu_hat, sigma2_hat=s2_hat,
                log_Z_tilde=float(log_Z_tilde), sweeps=it, mu=mu, Sigma_diag=Sigma_diag)


def epdtc_inference(kern, X, Z, sites):
    """EPDTC.inference for FIXED sites (ep_mode "alternated": every objective evaluation of one optimize() call):
    VarDTC.inference(Y = v / tau, precision = tau, Lm = jitchol(K(Z)), Z_tilde = log Z_tilde)."""
    kern = as_add(kern)
    tau, v = sites["tau"], sites["v"]
    num_data, num_inducing, output_dim = X.shape[0], Z.shape[0], 1
    Y = (v / tau)[:, None]
    precision = tau[:, None]
    VVT_factor = precision * Y
    Lm = jitchol(kern.K(Z))
    Kmm = tdot(Lm)
    psi0 = kern.Kdiag(X)
    psi1 = kern.K(X, Z)
    tmp = psi1 * np.sqrt(precision)
    tmp = dtrtrs(Lm, tmp.T, lower=1)
    A = tdot(tmp)
    B = np.eye(num_inducing) + A
    LB = jitchol(B)
    psi1Vf = np.dot(psi1.T, VVT_factor)
    tmp = dtrtrs(Lm, psi1Vf, lower=1, trans=0)
    _LBi_Lmi_psi1Vf = dtrtrs(LB, tmp, lower=1, trans=0)
    tmp = dtrtrs(LB, _LBi_Lmi_psi1Vf, lower=1, trans=1)
    Cpsi1Vf = dtrtrs(Lm, tmp, lower=1, trans=1)
    delit = tdot(_LBi_Lmi_psi1Vf)
    data_fit = np.trace(delit)
    DBi_plus_BiPBi = backsub_both_sides(LB, output_dim * np.eye(num_inducing) + delit)
    delit = -0.5 * DBi_plus_BiPBi
    delit += -0.5 * B * output_dim
    delit += output_dim * np.eye(num_inducing)
    dL_dKmm = backsub_both_sides(Lm, delit)
    # _compute_dL_dpsi, het_noise and certain inputs
    dL_dpsi0 = -0.5 * output_dim * (precision * np.ones([num_data, 1])).flatten()
    dL_dpsi1 = np.dot(VVT_factor, Cpsi1Vf.T)
    dL_dpsi2_beta = 0.5 * backsub_both_sides(Lm, output_dim * np.eye(num_inducing) - DBi_plus_BiPBi)
    dL_dpsi1 += 2.0 * np.dot(dL_dpsi2_beta, (psi1 * precision).T).T
    # _compute_log_marginal_likelihood, het_noise
    beta = precision
    lik_1 = (-0.5 * num_data * output_dim * LOG_2_PI + 0.5 * output_dim * np.sum(np.log(beta))
             - 0.5 * np.sum(beta.ravel() * np.square(Y).sum(axis=-1)))
    lik_2 = -0.5 * output_dim * (np.sum(beta.flatten() * psi0) - np.trace(A))
    lik_3 = -output_dim * (np.sum(np.log(np.diag(LB))))
    lik_4 = 0.5 * data_fit
    log_marginal = lik_1 + lik_2 + lik_3 + lik_4 + sites["log_Z_tilde"]
    Bi = -dpotri(LB, lower=1)
    Bi[np.diag_indices_from(Bi)] += 1.0
    woodbury_inv = backsub_both_sides(Lm, Bi)
    return dict(log_marginal=float(log_marginal), woodbury_vector=Cpsi1Vf, woodbury_inv=woodbury_inv, Lm=Lm, LB=LB,
                A=A, Kmm=Kmm, psi1=psi1, dL_dKmm=dL_dKmm, dL_dKnm=dL_dpsi1, dL_dKdiag=dL_dpsi0, dL_dthetaL=0.0,
                data_fit=data_fit)


def bernoulli_predictive_mean(mu, var):
    """Bernoulli.predictive_mean with the probit link: Phi(mu / sqrt(1 + var))  (GP.predict of a classifier,
    gp_model/base.py:187 -> classification.py:83,99)."""
    return _special.ndtr(mu / np.sqrt(1.0 + var))


# --------------------------------------------------------------------------
# Posterior._raw_predict + GP.predict + GP.predictive_gradients
# (gp_model/base.py:187, :206, :222)
# --------------------------------------------------------------------------
def raw_predict(kern, Xnew, pred_var, woodbury_vector, woodbury_inv):
    kern = as_add(kern)
    Kx = kern.K(pred_var, Xnew)
    mu = np.dot(Kx.T, woodbury_vector)
    Kxx = kern.Kdiag(Xnew)
    var = (Kxx - np.sum(np.dot(woodbury_inv.T, Kx) * Kx, 0))[:, None]
    var = np.clip(var, 1e-15, np.inf)
    return mu, var


def predict(kern, Xnew, pred_var, woodbury_vector, woodbury_inv, noise_var, normalizer=None):
    """GP.predict: latent moments -> + Gaussian noise -> Standardize inverse."""
    mu, var = raw_predict(kern, Xnew, pred_var, woodbury_vector, woodbury_inv)
    var = var + noise_var
    if normalizer is not None:
        mu = normalizer.inverse_mean(mu)
        var = normalizer.inverse_variance(var)
    return mu, var


def predictive_gradients_mean(kern, Xnew, pred_var, woodbury_vector, normalizer=None):
    """Mean part of GP.predictive_gradients -> (N*, D, 1).  The variance part
    is computed upstream and discarded by the reference (``[0]`` at
    gp_model/base.py:222).  Newer GPy rescales the Jacobian by the
    normaliser's std (gradient of the un-normalised mean); that behaviour is
    the one restated (a global scalar: EDR directions are invariant to it)."""
    kern = as_add(kern)
    mean_jac = np.empty((Xnew.shape[0], Xnew.shape[1], woodbury_vector.shape[1]))
    for i in range(woodbury_vector.shape[1]):
        mean_jac[:, :, i] = kern.gradients_X(woodbury_vector[:, i:i + 1].T, Xnew, pred_var)
    if normalizer is not None:
        mean_jac = mean_jac * normalizer.std
    return mean_jac


# --------------------------------------------------------------------------
# The lean (Gram-only) formulation the HIP path uses, restated on the CPU so
# tests can compare intermediate objects (Phi, b) as well as end results.
# Mathematically identical to vardtc_inference; see DESIGN.md.
# --------------------------------------------------------------------------
def gram_stats(kern, X, Z, Y):
    kern = as_add(kern)
    psi1 = kern.K(X, Z)
    return dict(Phi=psi1.T @ psi1, b=psi1.T @ Y, yy=float(np.sum(Y * Y)),
                psi0=float(np.sum(kern.Kdiag(X))))


def synthetic_problem(N, D, M, seed=0, dtype=np.float64):
    """SURVEY.md section 8(d) deterministic synthetic inputs: X ~ N(0, I);
    y = tanh(X w) + 0.05 eps, w = 1/sqrt(D) (functional form of
    edrgp/datasets.py:25-31), standardised (ddof 0); Z = X[perm[:M]]."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D)).astype(dtype)
    w = np.ones(D) / np.sqrt(D)
    y = np.tanh(X @ w) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    Z = X[rng.permutation(N)[:M]].copy()
    return X, y.astype(dtype), Z
