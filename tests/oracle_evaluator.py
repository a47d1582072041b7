"""Test-only evaluator: answers the model layer's device calls with the CPU oracle,
so the host logic (parameter packing, Logexp, L-BFGS-B driver, normaliser, EDR loop)
can be exercised without a GPU.  Never imported by the product."""
import numpy as np

from oracle import gp_oracle as go


class OracleEvaluator:
    def _kern(self, model):
        D = model.input_dim
        return go.Add([go.Stationary(p.kind, D, p.variance, p.lengthscale.copy(), p.ARD) for p in model.kern.parts])

    def _expand(self, kern, kg):
        D = kern.input_dim
        dvar = np.array([g[0] for g in kg])
        dlen = np.zeros((len(kg), D))
        for i, (g, part) in enumerate(zip(kg, kern.parts)):
            if part.ARD:
                dlen[i] = g[1]
            else:
                dlen[i, 0] = g[1][0]       # the model sums the row for an isotropic part
        return dvar, dlen

    def lml(self, model):
        k = self._kern(model)
        if model.is_sparse:
            return go.vardtc_inference(k, model.X, model.Z, model.noise_variance, model.Y_normalized)["log_marginal"][0, 0]
        return go.exact_inference(k, model.X, model.noise_variance, model.Y_normalized)["log_marginal"]

    def lml_grad(self, model):
        k = self._kern(model)
        if model.is_sparse:
            inf = go.vardtc_inference(k, model.X, model.Z, model.noise_variance, model.Y_normalized)
            kg, dn, dZ = go.sparse_gradients(k, model.X, model.Z, inf)
            dvar, dlen = self._expand(k, kg)
            return inf["log_marginal"][0, 0], dvar, dlen, dn, dZ
        inf = go.exact_inference(k, model.X, model.noise_variance, model.Y_normalized)
        kg, dn = go.dense_gradients(k, model.X, inf)
        dvar, dlen = self._expand(k, kg)
        return inf["log_marginal"], dvar, dlen, dn, None

    def _post(self, model):
        k = self._kern(model)
        if getattr(model, "is_classifier", False):
            if model.is_sparse:
                inf = go.epdtc_inference(k, model.X, model.Z, self._ensure_sites(model))
                return k, model.Z, inf["woodbury_vector"], inf["woodbury_inv"]
            inf = go.ep_inference(k, model.X, self._ensure_sites(model))
            return k, model.X, inf["woodbury_vector"], inf["woodbury_inv"]
        if model.is_sparse:
            inf = go.vardtc_inference(k, model.X, model.Z, model.noise_variance, model.Y_normalized)
            return k, model.Z, inf["woodbury_vector"], inf["woodbury_inv"]
        inf = go.exact_inference(k, model.X, model.noise_variance, model.Y_normalized)
        return k, model.X, inf["woodbury_vector"], inf["woodbury_inv"]

    def predict(self, model, Xs, want_var=True):
        k, P, wv, winv = self._post(model)
        mu, var = go.raw_predict(k, Xs, P, wv, winv)
        return mu[:, 0], (var[:, 0] if want_var else None)

    def predict_grad(self, model, Xs):
        k, P, wv, _ = self._post(model)
        return go.predictive_gradients_mean(k, Xs, P, wv)[:, :, 0]

    def predict_train(self, model, want_mean=True, want_grad=True):
        mu = self.predict(model, model.X, False)[0] if want_mean else None
        return mu, (self.predict_grad(model, model.X) if want_grad else None)

    # -- GP classification (EP) ----------------------------------------------------------
    def _ensure_sites(self, model):
        if model._sites is None:
            k = self._kern(model)
            if model.is_sparse:
                sites = go.expectation_propagation_dtc(k.K(model.Z), k.K(model.Z, model.X), model.Y[:, 0],
                                                       epsilon=model.ep_epsilon, eta=model.ep_eta, delta=model.ep_delta)
            else:
                sites = go.expectation_propagation(k.K(model.X), model.Y[:, 0], epsilon=model.ep_epsilon,
                                                   eta=model.ep_eta, delta=model.ep_delta)
            model._set_sites(sites["tau"], sites["v"], sites["log_Z_tilde"], sites["sweeps"])
        return dict(tau=model._sites[0], v=model._sites[1], log_Z_tilde=model._sites[2])

    def ep_lml(self, model):
        if model.is_sparse:
            return go.epdtc_inference(self._kern(model), model.X, model.Z, self._ensure_sites(model))["log_marginal"]
        return go.ep_inference(self._kern(model), model.X, self._ensure_sites(model))["log_marginal"]

    def ep_lml_grad(self, model):
        k = self._kern(model)
        if model.is_sparse:
            inf = go.epdtc_inference(k, model.X, model.Z, self._ensure_sites(model))
            kg, _, dZ = go.sparse_gradients(k, model.X, model.Z, inf)
            dvar, dlen = self._expand(k, kg)
            return inf["log_marginal"], dvar, dlen, dZ
        inf = go.ep_inference(k, model.X, self._ensure_sites(model))
        dvar, dlen = self._expand(k, k.param_gradients_full(inf["dL_dK"], model.X, None))
        return inf["log_marginal"], dvar, dlen, None
