"""Model objects behind ``estimator_`` -- the GPy surface the reference touches.

The reference builds ``GPy.models.GPRegression`` / ``SparseGPRegression``
(edrgp/gp_model/regression.py:76-77, :153-157) and then only calls
``optimize`` / ``optimize_restarts`` (gp_model/base.py:69), ``predict`` (:187,
:206), ``predictive_gradients`` (:222), ``pickle`` (:239) and, from its tests,
``log_likelihood()`` (tests/test_edr.py:49-50: a float for the dense model, a
``(1, 1)`` array for the sparse one).  These classes offer exactly that surface.

Host side (this file): parameter bookkeeping in GPy's order, the paramz
``Logexp`` transform, the L-BFGS-B driver (``scipy.optimize.fmin_l_bfgs_b``, the
optimiser paramz wraps), restarts, Standardize.  Device side: every likelihood /
gradient / posterior / prediction evaluation is one call into the HIP library
through an *evaluator* (``HipEvaluator``).  There is no CPU evaluator in the
product; tests may inject one built on the oracle to exercise this host logic
without a GPU.
"""
from __future__ import annotations

import itertools

import numpy as np
from scipy import optimize as _sciopt

from . import kern as _kern
from ._lib import NotPositiveDefiniteError

_LIM_VAL = 36.0
_token_counter = itertools.count(1)


# ---------------------------------------------------------------------------
# paramz.transformations.Logexp
# ---------------------------------------------------------------------------
def logexp_f(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > _LIM_VAL, x, np.log1p(np.exp(np.clip(x, -700.0, _LIM_VAL))))


def logexp_finv(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f > _LIM_VAL, f, np.log(np.expm1(f)))


def logexp_gradfactor(f, df):
    f = np.asarray(f, dtype=np.float64)
    return df * np.where(f > _LIM_VAL, 1.0, -np.expm1(-f))


# ---------------------------------------------------------------------------
# GPy.util.normalizer.Standardize
# ---------------------------------------------------------------------------
class Standardize:
    def __init__(self):
        self.mean = None
        self.std = None

    def scale_by(self, Y):
        Y = np.asarray(Y, dtype=np.float64)
        self.mean = Y.mean(0)
        self.std = Y.std(0)

    def scaled(self):
        return self.mean is not None

    def normalize(self, Y):
        return (Y - self.mean) / self.std

    def inverse_mean(self, X):
        return X * self.std + self.mean

    def inverse_variance(self, var):
        return var * self.std ** 2


def make_normalizer(normalizer):
    """GPy GP.__init__: True -> Standardize, False/None -> none, object -> itself."""
    if normalizer is True:
        return Standardize()
    if normalizer is False or normalizer is None:
        return None
    return normalizer


# ---------------------------------------------------------------------------
# Device evaluator
# ---------------------------------------------------------------------------
class HipEvaluator:
    """Binds one model to the shared native context (edr-gp_amd/device.py)."""

    def __init__(self, device=None):
        from .device import shared_context
        self._device = device
        self._ctx = shared_context(device)
        self._token = next(_token_counter)
        self._sent = {}

    # a copied / unpickled model gets a fresh binding to the shared context
    # (native handles are never copied)
    def __deepcopy__(self, memo):
        return HipEvaluator(self._device)

    def __reduce__(self):
        return (HipEvaluator, (self._device,))

    @property
    def hip(self):
        return self._ctx.hip

    def _sync_state(self, model):
        resident = self._ctx.claim(self._token)
        if not resident:
            self._sent = {}
        h = self._ctx.hip
        prec = int(getattr(model, "precision", 64))
        if self._sent.get("prec") != prec:
            h.set_option("precision", prec)
            self._sent["prec"] = prec
            self._sent.pop("post", None)
        if self._sent.get("data") is not model._data_version:
            self._ctx.send_data(model.X, model.Y_normalized[:, 0])
            self._sent = {"data": model._data_version, "prec": prec}
        if model.is_sparse:
            zkey = model.Z.tobytes()
            if self._sent.get("Z") != zkey:
                h.set_inducing(model.Z)
                self._sent["Z"] = zkey
        parts = model.kern.parts
        kkey = (tuple(p.kind for p in parts), tuple(p.variance for p in parts),
                tuple(p.lengthscale.tobytes() for p in parts), float(model.noise_variance))
        if self._sent.get("kern") != kkey:
            h.set_kernel([p.kind for p in parts], [p.variance for p in parts],
                         np.array([p.expanded_lengthscale() for p in parts]), float(model.noise_variance))
            self._sent["kern"] = kkey
            self._sent.pop("post", None)
        return h

    def lml(self, model):
        h = self._sync_state(model)
        return h.sparse_lml() if model.is_sparse else h.dense_lml()

    def lml_grad(self, model):
        h = self._sync_state(model)
        if model.is_sparse:
            return h.sparse_lml_grad()
        lml, dvar, dlen, dnoise = h.dense_lml_grad()
        return lml, dvar, dlen, dnoise, None

    def _ensure_sites(self, model):
        """EP site parameters resident on the device: fitted once per optimize() call (GPy EP, ep_mode "alt"), or
        restored from the model when another model used the shared context in between."""
        h = self._sync_state(model)
        if model._sites is None:
            # dense EP on K(X, X), or EPDTC on the inducing-point posterior (same site updates and stop rule)
            begin, sweep = (h.epdtc_begin, h.epdtc_sweep) if model.is_sparse else (h.ep_begin, h.ep_sweep)
            begin(model.Y[:, 0])
            sweeps, stop = 0, False
            while not stop and sweeps < model.ep_max_iters:
                order = np.random.permutation(model.num_data)          # upstream: a fresh permutation per sweep
                d_tau, d_v = sweep(order, model.ep_eta, model.ep_delta)
                if sweeps > 0:
                    stop = (d_tau < model.ep_epsilon) and (d_v < model.ep_epsilon)
                sweeps += 1
            log_Z_tilde = h.ep_finish()
            tau, v = h.ep_get_sites()
            model._set_sites(tau, v, log_Z_tilde, sweeps)
            self._sent["sites"] = model._sites_version
            self._sent.pop("post", None)
        elif self._sent.get("sites") is not model._sites_version:
            h.ep_set_sites(*model._sites)
            self._sent["sites"] = model._sites_version
            self._sent.pop("post", None)
        return h

    def ep_lml(self, model):
        h = self._ensure_sites(model)
        return h.epdtc_lml() if model.is_sparse else h.ep_lml()

    def ep_lml_grad(self, model):
        """(lml, dvar, dlen, dZ); dZ is None for the dense classifier."""
        h = self._ensure_sites(model)
        if model.is_sparse:
            return h.epdtc_lml_grad()
        return h.ep_lml_grad() + (None,)

    def _ensure_posterior(self, model):
        classifier = getattr(model, "is_classifier", False)
        h = self._ensure_sites(model) if classifier else self._sync_state(model)
        key = (self._sent.get("kern"), self._sent.get("Z"), self._sent.get("sites"))
        if self._sent.get("post") != key:
            if classifier:
                model._lml_cache = h.epdtc_posterior() if model.is_sparse else h.ep_posterior()
            else:
                model._lml_cache = h.sparse_posterior() if model.is_sparse else h.dense_posterior()
            self._sent["post"] = key
        return h

    def predict(self, model, Xs, want_var=True):
        return self._ensure_posterior(model).predict(Xs, want_var)

    def predict_grad(self, model, Xs):
        return self._ensure_posterior(model).predict_grad(Xs)

    def predict_train(self, model, want_mean=True, want_grad=True):
        return self._ensure_posterior(model).predict_train(want_mean, want_grad)

    def predict_train_gram(self, model):
        """J^T J (D x D) and column sums of the mean Jacobian J at the training rows, reduced on the device."""
        _, gram, gsum = self._ensure_posterior(model).predict_train_gram(want_mean=False)
        return gram, gsum

    def posterior_factors(self, model):
        h = self._ensure_posterior(model)
        P = model.Z.shape[0] if model.is_sparse else model.X.shape[0]
        return h.get("WV", (P,)), h.get("WINV", (P, P))


class OptimizationResult:
    def __init__(self, x_opt, f_opt, funct_eval, status):
        self.x_opt, self.f_opt, self.funct_eval, self.status = x_opt, f_opt, funct_eval, status


# ---------------------------------------------------------------------------
# Models
# ---------------------------------------------------------------------------
class _GPModel:
    is_sparse = False
    _allowed_failures = 10

    def __init__(self, X, Y, kernel, noise_var, normalizer, evaluator=None, device=None):
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        Y = np.asarray(Y, dtype=np.float64)
        if Y.ndim == 1:
            Y = Y[:, None]
        if X.ndim != 2 or Y.shape[0] != X.shape[0]:
            raise ValueError("X must be (N, D) and Y (N, 1)")
        if Y.shape[1] != 1:
            raise NotImplementedError("the MI355X path supports a single output column")
        self.X, self.Y = X, Y
        self.num_data, self.input_dim = X.shape
        self.output_dim = 1
        self.normalizer = make_normalizer(normalizer)
        if self.normalizer is not None:
            self.normalizer.scale_by(Y)
            self.Y_normalized = self.normalizer.normalize(Y)
        else:
            self.Y_normalized = Y
        self._data_version = object()
        self.kern = _kern.RBF(self.input_dim) if kernel is None else kernel
        if self.kern.input_dim != self.input_dim:
            raise ValueError("kernel input_dim %d != data dimension %d" % (self.kern.input_dim, self.input_dim))
        self.noise_variance = float(noise_var)
        self.optimization_runs = []
        self._fail_count = 0
        self._last_grads = None
        self._lml_cache = None
        self._evaluator = evaluator if evaluator is not None else HipEvaluator(device)

    # -- parameter vector in GPy's order ---------------------------------------
    def _positive_params(self):
        vals = []
        for p in self.kern.parts:
            vals.append(p.variance)
            vals.extend(p.lengthscale.tolist())
        vals.append(self.noise_variance)
        return np.array(vals, dtype=np.float64)

    def _set_positive_params(self, vals):
        i = 0
        for p in self.kern.parts:
            p.variance = float(vals[i]); i += 1
            n = p.lengthscale.size
            p.lengthscale = np.array(vals[i:i + n], dtype=np.float64); i += n
        self.noise_variance = float(vals[i])
        self._lml_cache = None

    def _free_block(self):          # unconstrained leading block (Z in the sparse model)
        return np.zeros(0)

    def _set_free_block(self, vals):
        pass

    @property
    def optimizer_array(self):
        return np.concatenate([self._free_block(), logexp_finv(self._positive_params())])

    @optimizer_array.setter
    def optimizer_array(self, x):
        x = np.asarray(x, dtype=np.float64)
        nf = self._free_block().size
        self._set_free_block(x[:nf])
        self._set_positive_params(logexp_f(x[nf:]))

    @property
    def param_array(self):
        return np.concatenate([self._free_block(), self._positive_params()])

    def parameter_names(self):
        names = []
        multi = len(self.kern.parts) > 1
        for p in self.kern.parts:
            prefix = ("sum." + p.name) if multi else p.name
            names.append(prefix + ".variance")
            names.extend([prefix + ".lengthscale"] * p.lengthscale.size)
        names.append("Gaussian_noise.variance")
        return names

    # -- objective ----------------------------------------------------------------
    def _pack_gradients(self, dvar, dlen, dnoise, dZ):
        g = []
        for i, p in enumerate(self.kern.parts):
            g.append(dvar[i])
            if p.ARD:
                g.extend(np.asarray(dlen[i]).tolist())
            else:
                g.append(float(np.sum(dlen[i])))
        g.append(dnoise)
        return np.array(g, dtype=np.float64)

    def _log_likelihood_gradients(self):
        lml, dvar, dlen, dnoise, dZ = self._evaluator.lml_grad(self)
        self._lml_cache = lml
        gpos = self._pack_gradients(dvar, dlen, dnoise, dZ)
        gfree = np.zeros(0) if dZ is None else np.asarray(dZ).ravel()
        return lml, gfree, gpos

    def _objective_grads(self, x):
        """paramz Model._objective_grads: -(LML, transformed gradient); a linear-algebra
        failure counts against a budget and is reported as an infinite objective."""
        try:
            self.optimizer_array = x
            lml, gfree, gpos = self._log_likelihood_gradients()
            grads = -np.concatenate([gfree, logexp_gradfactor(self._positive_params(), gpos)])
            if not np.isfinite(lml):
                raise ValueError("non-finite log likelihood")
            self._fail_count = 0
            self._last_grads = grads
            return -lml, grads
        except (np.linalg.LinAlgError, ZeroDivisionError, ValueError, NotPositiveDefiniteError):
            if self._fail_count >= self._allowed_failures:
                raise
            self._fail_count += 1
            g = self._last_grads if self._last_grads is not None else np.zeros_like(x)
            return np.inf, np.clip(g, -1e10, 1e10)

    def objective_function(self):
        return -float(np.sum(self.log_likelihood()))

    # -- GPy surface -------------------------------------------------------------
    def _lml(self):
        if self._lml_cache is None:
            self._lml_cache = self._evaluator.lml(self)
        return self._lml_cache

    def optimize(self, optimizer=None, start=None, messages=False, max_iters=1000, **kwargs):
        """paramz Model.optimize with its default optimiser: L-BFGS-B through
        scipy.optimize.fmin_l_bfgs_b(f_fp, x0, maxfun=max_iters, maxiter=max_iters)."""
        if optimizer not in (None, "lbfgsb", "lbfgs", "bfgs", "scg"):
            raise NotImplementedError("optimizer %r: only L-BFGS-B is available" % (optimizer,))
        x0 = self.optimizer_array if start is None else np.asarray(start, dtype=np.float64)
        opt_dict = {}
        if kwargs.get("gtol") is not None:
            opt_dict["pgtol"] = kwargs["gtol"]
        if kwargs.get("bfgs_factor") is not None:
            opt_dict["factr"] = kwargs["bfgs_factor"]
        self._fail_count = 0
        from .utils import host_threads
        with host_threads():         # keep the host BLAS pool small while the device does the work
            x_opt, f_opt, info = _sciopt.fmin_l_bfgs_b(self._objective_grads, x0, maxfun=int(max_iters),
                                                       maxiter=int(max_iters), **opt_dict)
        self.optimizer_array = x_opt
        status = {0: "Converged", 1: "Maximum number of f evaluations reached"}.get(info["warnflag"], info.get("task", "Error"))
        run = OptimizationResult(np.array(x_opt), float(f_opt), info["funcalls"], status)
        self.optimization_runs.append(run)
        if messages:
            print("L-BFGS-B: f = %.6g after %d evaluations (%s)" % (run.f_opt, run.funct_eval, status))
        return run

    def randomize(self, rng=None):
        """paramz: the optimiser vector is redrawn from N(0, 1)."""
        draw = (np.random.normal if rng is None else rng.normal)
        self.optimizer_array = draw(loc=0.0, scale=1.0, size=self.optimizer_array.size)

    def optimize_restarts(self, num_restarts=10, robust=False, verbose=True, parallel=False, num_processes=None,
                          **kwargs):
        """paramz Model.optimize_restarts (sequential; ``parallel`` is accepted and ignored:
        one device context, evaluations are already device-wide)."""
        initial_length = len(self.optimization_runs)
        initial_parameters = self.optimizer_array.copy()
        for i in range(num_restarts):
            try:
                if i > 0:
                    self.randomize()
                self.optimize(**kwargs)
                if verbose:
                    print("Optimization restart %d/%d, f = %s" % (i + 1, num_restarts, self.optimization_runs[-1].f_opt))
            except Exception as e:       # noqa: BLE001  (paramz does the same)
                if robust:
                    print("Warning - optimization restart %d/%d failed: %s" % (i + 1, num_restarts, e))
                else:
                    raise
        if len(self.optimization_runs) > initial_length:
            best = int(np.argmin([o.f_opt for o in self.optimization_runs[initial_length:]]))
            self.optimizer_array = self.optimization_runs[initial_length + best].x_opt
        else:
            self.optimizer_array = initial_parameters
        return self.optimization_runs

    def _raw_predict(self, Xnew, want_var=True):
        Xnew = np.ascontiguousarray(np.asarray(Xnew, dtype=np.float64))
        mu, var = self._evaluator.predict(self, Xnew, want_var)
        return mu[:, None], (None if var is None else var[:, None])

    def predict(self, Xnew, full_cov=False, Y_metadata=None, kern=None, likelihood=None, include_likelihood=True):
        """GPy GP.predict: latent moments, + Gaussian noise, un-normalised.  -> (mean (n,1), var (n,1))"""
        if full_cov:
            raise NotImplementedError("full_cov=True is not on the MI355X path")
        mu, var = self._raw_predict(Xnew)
        if include_likelihood:
            var = var + self.noise_variance
        if self.normalizer is not None:
            mu = self.normalizer.inverse_mean(mu)
            var = self.normalizer.inverse_variance(var)
        return mu, var

    def predictive_gradients(self, Xnew, kern=None):
        """GPy GP.predictive_gradients: (d mean / d x* as (n, D, 1), None).  The
        variance Jacobian is computed upstream and discarded by the reference
        (``[0]`` at gp_model/base.py:222); it is not computed here.  The mean
        Jacobian is that of the un-normalised mean (scaled by the normaliser's std)."""
        Xnew = np.ascontiguousarray(np.asarray(Xnew, dtype=np.float64))
        if Xnew.shape == self.X.shape and np.array_equal(Xnew, self.X):     # training inputs are resident
            _, jac = self._evaluator.predict_train(self, want_mean=False, want_grad=True)
        else:
            jac = self._evaluator.predict_grad(self, Xnew)
        if self.normalizer is not None:
            jac = jac * self.normalizer.std
        return jac[:, :, None], None

    def predictive_gradient_gram(self, Xnew):
        """G^T G (D x D) and the column sums of G = d mean / d x* (un-normalised mean) at the TRAINING inputs, without
        moving the N x D matrix G to the host: what the EDR step consumes (edrgp/base.py:459-463 -> a linear
        decomposition of G).  Returns None when Xnew is not the training set (the caller then takes
        ``predictive_gradients``)."""
        Xnew = np.asarray(Xnew, dtype=np.float64)
        if Xnew.shape != self.X.shape or not np.array_equal(Xnew, self.X):
            return None
        if hasattr(self._evaluator, "predict_train_gram"):
            gram, gsum = self._evaluator.predict_train_gram(self)
        else:                                        # evaluators without the device-side reduction (tests)
            _, jac = self._evaluator.predict_train(self, want_mean=False, want_grad=True)
            gram, gsum = jac.T @ jac, jac.sum(0)
        if self.normalizer is not None:
            std = float(np.ravel(self.normalizer.std)[0])
            gram, gsum = gram * std * std, gsum * std
        return gram, gsum, self.num_data

    # -- persistence (replaces GPy's pickle: data, parameters, no code) -----------
    def to_dict(self):
        parts = self.kern.parts
        d = dict(model=type(self).__name__, X=self.X, Y=self.Y, noise_variance=self.noise_variance,
                 kinds=np.array([p.kind for p in parts]), variances=np.array([p.variance for p in parts]),
                 ARD=np.array([p.ARD for p in parts]),
                 normalize=np.array(isinstance(self.normalizer, Standardize)))
        for i, p in enumerate(parts):
            d["lengthscale_%d" % i] = p.lengthscale
        return d

    def pickle(self, path):
        with open(path, "wb") as f:
            np.savez(f, **self.to_dict())


class GPRegressionModel(_GPModel):
    """Stands where ``GPy.models.GPRegression`` stands (regression.py:76-77)."""

    def __init__(self, X, Y, kernel=None, Y_metadata=None, normalizer=None, noise_var=1.0, mean_function=None,
                 evaluator=None, device=None):
        if Y_metadata is not None or mean_function is not None:
            raise NotImplementedError("Y_metadata / mean_function are GPy-specific and not supported")
        super().__init__(X, Y, kernel, noise_var, normalizer, evaluator, device)

    def log_likelihood(self):
        return float(self._lml())


class SparseGPRegressionModel(_GPModel):
    """Stands where ``GPy.models.SparseGPRegression`` stands (regression.py:153-157):
    Z defaults to ``X[np.random.permutation(N)[:min(num_inducing, N)]]`` (global numpy
    RNG, as upstream), is an optimised parameter, noise starts at 1."""
    is_sparse = True

    def __init__(self, X, Y, kernel=None, Z=None, num_inducing=10, X_variance=None, mean_function=None,
                 normalizer=None, evaluator=None, device=None, precision=64):
        if X_variance is not None or mean_function is not None:
            raise NotImplementedError("X_variance / mean_function are GPy-specific and not supported")
        if precision not in (32, 64):
            raise ValueError("precision must be 32 or 64")
        super().__init__(X, Y, kernel, 1.0, normalizer, evaluator, device)
        # 32: the row pass of the native path in fp32 where that is safe (include/edrgp_hip.h, "precision")
        self.precision = precision
        if Z is None:
            i = np.random.permutation(self.num_data)[:min(num_inducing, self.num_data)]
            Z = self.X[i].copy()
        else:
            Z = np.array(Z, dtype=np.float64)
            assert Z.ndim == 2 and Z.shape[1] == self.input_dim
        self.Z = np.ascontiguousarray(Z)
        self.num_inducing = self.Z.shape[0]

    def _free_block(self):
        return self.Z.ravel()

    def _set_free_block(self, vals):
        self.Z = np.ascontiguousarray(np.asarray(vals, dtype=np.float64).reshape(self.Z.shape))
        self._lml_cache = None

    def parameter_names(self):
        return ["inducing_inputs"] * self.Z.size + super().parameter_names()

    def log_likelihood(self):
        return np.array([[self._lml()]])        # (1, 1), as VarDTC returns it

    def to_dict(self):
        d = super().to_dict()
        d["Z"] = self.Z
        d["precision"] = np.array(self.precision)
        return d


class GPClassificationModel(_GPModel):
    """Stands where ``GPy.models.GPClassification`` stands (edrgp/gp_model/classification.py:63-64): Bernoulli
    likelihood with the probit link, expectation propagation (GPy ``EP``: epsilon 1e-6, eta = delta = 1, sequential
    site updates in a random order, mode "alt" -- the sites are fitted once at the start of every ``optimize()`` call
    and held fixed while the kernel hyper-parameters are optimised).  Parameters: the kernel's only (no noise)."""
    is_classifier = True

    def __init__(self, X, Y, kernel=None, Y_metadata=None, mean_function=None, evaluator=None, device=None):
        if Y_metadata is not None or mean_function is not None:
            raise NotImplementedError("Y_metadata / mean_function are GPy-specific and not supported")
        Y = np.asarray(Y, dtype=np.float64)
        if not np.all(np.isin(np.unique(Y), (0.0, 1.0))):
            raise ValueError("GP classification expects labels 0 / 1")
        super().__init__(X, Y, kernel, 1.0, None, evaluator, device)
        self.ep_epsilon, self.ep_eta, self.ep_delta, self.ep_max_iters = 1e-6, 1.0, 1.0, 10 ** 6
        self._sites = None                 # (tau_tilde, v_tilde, log_Z_tilde) of the last EP run
        self._sites_version = None
        self.ep_sweeps = 0

    def _set_sites(self, tau, v, log_Z_tilde, sweeps=0):
        self._sites = (np.array(tau, dtype=np.float64), np.array(v, dtype=np.float64), float(log_Z_tilde))
        self._sites_version = object()
        self.ep_sweeps = int(sweeps)
        self._lml_cache = None

    # -- parameters: kernel only ---------------------------------------------------
    def _positive_params(self):
        vals = []
        for p in self.kern.parts:
            vals.append(p.variance)
            vals.extend(p.lengthscale.tolist())
        return np.array(vals, dtype=np.float64)

    def _set_positive_params(self, vals):
        i = 0
        for p in self.kern.parts:
            p.variance = float(vals[i]); i += 1
            n = p.lengthscale.size
            p.lengthscale = np.array(vals[i:i + n], dtype=np.float64); i += n
        self._lml_cache = None

    def parameter_names(self):
        return super().parameter_names()[:-1]

    def _log_likelihood_gradients(self):
        lml, dvar, dlen, dZ = self._evaluator.ep_lml_grad(self)
        self._lml_cache = lml
        g = []
        for i, p in enumerate(self.kern.parts):
            g.append(dvar[i])
            if p.ARD:
                g.extend(np.asarray(dlen[i]).tolist())
            else:
                g.append(float(np.sum(dlen[i])))
        return lml, (np.zeros(0) if dZ is None else np.asarray(dZ).ravel()), np.array(g, dtype=np.float64)

    def _lml(self):
        if self._lml_cache is None:
            self._lml_cache = self._evaluator.ep_lml(self)
        return self._lml_cache

    def log_likelihood(self):
        return float(self._lml())

    def optimize(self, *args, **kwargs):
        self._sites = None                 # EP.on_optimization_start: a fresh EP run for this call
        self._lml_cache = None
        return super().optimize(*args, **kwargs)

    # -- prediction ------------------------------------------------------------------
    def predict(self, Xnew, full_cov=False, Y_metadata=None, kern=None, likelihood=None, include_likelihood=True):
        """GPy GP.predict through the Bernoulli likelihood: (P(y = 1 | x*) = Phi(mu / sqrt(1 + var)) as (n, 1), NaN):
        the probit link has no closed-form predictive variance upstream either."""
        from scipy.special import ndtr
        if full_cov:
            raise NotImplementedError("full_cov=True is not on the MI355X path")
        mu, var = self._raw_predict(Xnew)
        if not include_likelihood:
            return mu, var
        return ndtr(mu / np.sqrt(1.0 + var)), np.full_like(mu, np.nan)

    def to_dict(self):
        d = super().to_dict()
        d.pop("noise_variance", None)
        d["noise_variance"] = 0.0
        if self._sites is not None:
            d["ep_tau"], d["ep_v"], d["ep_log_Z_tilde"] = self._sites[0], self._sites[1], np.array(self._sites[2])
        return d


class SparseGPClassificationModel(GPClassificationModel):
    """Stands where ``GPy.models.SparseGPClassification`` stands (edrgp/gp_model/classification.py:147-149):
    SparseGP with the Bernoulli / probit likelihood and EPDTC -- EP's site updates on the inducing-point (DTC)
    posterior, then VarDTC with the pseudo-targets v / tau and the heteroscedastic precision tau.  Z defaults to
    ``X[np.random.permutation(N)[:num_inducing]]`` (global numpy RNG, as upstream) and is an optimised parameter
    beside the kernel's; the sites are fitted once per ``optimize()`` call (EPDTC's default ep_mode "alternated")."""
    is_sparse = True

    def __init__(self, X, Y, likelihood=None, kernel=None, Z=None, num_inducing=10, Y_metadata=None, evaluator=None,
                 device=None):
        if likelihood is not None:
            raise NotImplementedError("only the default likelihood (Bernoulli, probit link) is on the MI355X path")
        super().__init__(X, Y, kernel, Y_metadata, None, evaluator, device)
        if Z is None:
            i = np.random.permutation(self.num_data)[:min(num_inducing, self.num_data)]
            Z = self.X[i].copy()
        else:
            Z = np.array(Z, dtype=np.float64)
            assert Z.ndim == 2 and Z.shape[1] == self.input_dim
        self.Z = np.ascontiguousarray(Z)
        self.num_inducing = self.Z.shape[0]

    def _free_block(self):
        return self.Z.ravel()

    def _set_free_block(self, vals):
        self.Z = np.ascontiguousarray(np.asarray(vals, dtype=np.float64).reshape(self.Z.shape))
        self._lml_cache = None

    def parameter_names(self):
        return ["inducing_inputs"] * self.Z.size + super().parameter_names()

    def to_dict(self):
        d = super().to_dict()
        d["Z"] = self.Z
        return d


def load_model(path, evaluator=None, device=None):
    """Rebuilds a model written by ``pickle`` (an .npz archive: arrays and strings only)."""
    with np.load(path, allow_pickle=False) as f:
        kinds = [str(k) for k in f["kinds"]]
        parts = [getattr(_kern, kinds[i])(f["X"].shape[1], float(f["variances"][i]), f["lengthscale_%d" % i],
                                          bool(f["ARD"][i])) for i in range(len(kinds))]
        kernel = parts[0] if len(parts) == 1 else _kern.Add(parts)
        normalizer = bool(f["normalize"])
        if str(f["model"]) in ("GPClassificationModel", "SparseGPClassificationModel"):
            if str(f["model"]) == "SparseGPClassificationModel":
                m = SparseGPClassificationModel(f["X"], f["Y"], kernel=kernel, Z=f["Z"], evaluator=evaluator, device=device)
            else:
                m = GPClassificationModel(f["X"], f["Y"], kernel, evaluator=evaluator, device=device)
            if "ep_tau" in f.files:
                m._set_sites(f["ep_tau"], f["ep_v"], float(f["ep_log_Z_tilde"]))
            return m
        if str(f["model"]) == "SparseGPRegressionModel":
            m = SparseGPRegressionModel(f["X"], f["Y"], kernel, Z=f["Z"], normalizer=normalizer,
                                        evaluator=evaluator, device=device,
                                        precision=int(f["precision"]) if "precision" in f.files else 64)
        else:
            m = GPRegressionModel(f["X"], f["Y"], kernel, normalizer=normalizer, evaluator=evaluator, device=device)
        m.noise_variance = float(f["noise_variance"])
    return m
