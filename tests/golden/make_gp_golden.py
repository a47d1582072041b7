"""Generates tests/golden/gp_golden.npz: known-answer vectors for the GP path.

Source of the numbers: the CPU oracle (oracle/gp_oracle.py), cross-checked at
generation time against scikit-learn's exact GP (log marginal likelihood and
posterior mean, |d| < 1e-9) -- the reference holds no golden vectors for this
path and GPy cannot be run here (see the oracle's header).  Inputs and expected
outputs only; run from the repo root:  python tests/golden/make_gp_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as go  # noqa: E402
from sklearn.gaussian_process import GaussianProcessRegressor as SK  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel  # noqa: E402

SK_KERN = {"RBF": lambda l: RBF(l), "Matern52": lambda l: Matern(l, nu=2.5), "Matern32": lambda l: Matern(l, nu=1.5)}
out = {}
case_id = 0
for seed in (0, 1, 2):
    for kind in ("RBF", "Matern52", "Matern32"):
        for (N, M, D) in ((256, 32, 4), (512, 64, 8)):
            X, y, Z = go.synthetic_problem(N, D, M, seed=seed)
            var, ls, noise = 0.8 + 0.1 * seed, 0.9 * np.sqrt(D), 0.05
            kern = go.make_kernel([kind], D, [dict(variance=var, lengthscale=ls)])
            Y = y[:, None]
            inf = go.vardtc_inference(kern, X, Z, noise, Y)
            kg, dn, dZ = go.sparse_gradients(kern, X, Z, inf)
            rng = np.random.default_rng(100 + seed)
            Xs = rng.standard_normal((16, D))
            mu, v = go.raw_predict(kern, Xs, Z, inf["woodbury_vector"], inf["woodbury_inv"])
            jac = go.predictive_gradients_mean(kern, Xs, Z, inf["woodbury_vector"])[:, :, 0]
            dense = go.exact_inference(kern, X, noise, Y)
            dkg, ddn = go.dense_gradients(kern, X, dense)
            dmu, dv = go.raw_predict(kern, Xs, X, dense["woodbury_vector"], dense["woodbury_inv"])
            # independent known answers (scikit-learn)
            sk = SK(ConstantKernel(var) * SK_KERN[kind](ls) + WhiteKernel(noise), alpha=1e-8, optimizer=None).fit(X, y)
            sk_lml = sk.log_marginal_likelihood(sk.kernel_.theta)
            assert abs(sk_lml - dense["log_marginal"]) < 1e-9 * abs(sk_lml), (sk_lml, dense["log_marginal"])
            assert np.abs(sk.predict(Xs) - dmu[:, 0]).max() < 1e-9
            assert inf["log_marginal"][0, 0] <= dense["log_marginal"] + 1e-6          # variational bound
            st = go.gram_stats(kern, X, Z, Y)
            p = "c%02d_" % case_id
            out[p + "meta"] = np.array([seed, N, M, D])
            out[p + "kind"] = np.array(kind)
            out[p + "theta"] = np.array([var, ls, noise])
            out[p + "X"], out[p + "y"], out[p + "Z"], out[p + "Xs"] = X, y, Z, Xs
            out[p + "bound"] = inf["log_marginal"]
            out[p + "phi_sum_trace"] = np.array([st["Phi"].sum(), np.trace(st["Phi"]), st["b"].sum()])
            out[p + "wv"] = inf["woodbury_vector"][:, 0]
            # fingerprints of the M x M intermediates: [sum, trace, Frobenius norm] and the diagonals
            fp = lambda A: np.array([A.sum(), np.trace(A), np.linalg.norm(A)])   # noqa: E731
            out[p + "kmm_fp"], out[p + "lm_fp"], out[p + "lb_fp"] = fp(inf["Kmm"]), fp(inf["Lm"]), fp(inf["LB"])
            out[p + "knm_fp"] = np.array([inf["psi1"].sum(), np.linalg.norm(inf["psi1"])])
            out[p + "lm_diag"], out[p + "lb_diag"] = np.diag(inf["Lm"]).copy(), np.diag(inf["LB"]).copy()
            out[p + "pred_mean"], out[p + "pred_var"], out[p + "pred_jac"] = mu[:, 0], v[:, 0], jac
            out[p + "grad"] = np.array([kg[0][0], kg[0][1][0], dn])
            out[p + "dZ"] = dZ
            out[p + "dense_lml"] = np.array([dense["log_marginal"], sk_lml])
            out[p + "dense_grad"] = np.array([dkg[0][0], dkg[0][1][0], ddn])
            out[p + "dense_pred_mean"], out[p + "dense_pred_var"] = dmu[:, 0], dv[:, 0]
            case_id += 1
out["n_cases"] = np.array(case_id)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gp_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB,", case_id, "cases")
