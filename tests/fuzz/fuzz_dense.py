"""Random-shape parity fuzz of the dense exact GP and of the device-side EDR step against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as go
from edrgp_amd.backend import HipGP
kinds_all = ["RBF", "Matern32", "Matern52"]


def run(h, seed=0, n_cases=40, verbose=True):
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(n_cases):
        D = int(rng.integers(1, 12))
        N = int(rng.integers(2, 700))
        nparts = 1 if rng.random() < 0.7 else 2
        kinds = [kinds_all[int(rng.integers(0, 3))] for _ in range(nparts)]
        opts = [dict(variance=float(rng.uniform(0.3, 2.0)), lengthscale=float(rng.uniform(0.8, 3.0) * np.sqrt(D))) for _ in range(nparts)]
        noise = float(10 ** rng.uniform(-2, 0))
        X, y, _ = go.synthetic_problem(N, D, 1, seed=2000 + case)
        kern = go.make_kernel(kinds, D, opts)
        try:
            inf = go.exact_inference(kern, X, noise, y[:, None])
            kg, dn = go.dense_gradients(kern, X, inf)
            h.set_data(X, y)
            h.set_kernel(kinds, [o["variance"] for o in opts], np.array([np.full(D, o["lengthscale"]) for o in opts]), noise)
            lml, dvar, dlen, dnoise = h.dense_lml_grad()
            h.dense_posterior()
            Xs = np.vstack([X[:20], np.random.default_rng(case).standard_normal((20, D))])
            mu, var = h.predict(Xs)
            jac = h.predict_grad(Xs)
            mu_ref, var_ref = go.raw_predict(kern, Xs, X, inf["woodbury_vector"], inf["woodbury_inv"])
            jac_ref = go.predictive_gradients_mean(kern, Xs, X, inf["woodbury_vector"])[:, :, 0]
            _, G, gs = h.predict_train_gram()
            _, J = h.predict_train(False, True)
            gscale = max(abs(dn), max(abs(g[0]) for g in kg), max(np.abs(g[1]).max() for g in kg))
            ref = float(inf["log_marginal"])
            cond = np.linalg.cond(kern.K(X) + (noise + 1e-8) * np.eye(N))
            amp = max(1.0, cond / 1e6)
            errs = dict(lml=abs(lml - ref) / abs(ref),
                        dvar=max(abs(dvar[p] - kg[p][0]) for p in range(nparts)) / gscale,
                        dlen=max(abs(dlen[p].sum() - np.sum(kg[p][1])) for p in range(nparts)) / gscale,
                        dnoise=abs(dnoise - dn) / gscale,
                        mean=np.abs(mu - mu_ref[:, 0]).max() / max(np.abs(mu_ref).max(), 1e-12),
                        var=np.abs(var - var_ref[:, 0]).max() / max(1.0, np.abs(var_ref).max()),
                        jac=np.abs(jac - jac_ref).max() / max(np.abs(jac_ref).max(), 1e-12),
                        gram=np.abs(G - J.T @ J).max() / max(np.abs(J.T @ J).max(), 1e-300),
                        gsum=np.abs(gs - J.sum(0)).max() / max(np.abs(J).sum(0).max(), 1e-300))
            tol = {k: v * amp for k, v in dict(lml=1e-9, dvar=1e-6, dlen=1e-6, dnoise=1e-6, mean=1e-7, var=1e-6, jac=1e-7).items()}
            tol.update(gram=1e-12, gsum=1e-12)
            fails = {k: float(v) for k, v in errs.items() if not (v <= tol[k])}
            if fails:
                bad.append(case)
                if verbose: print("CASE", case, dict(N=N, D=D, kinds=kinds, noise=noise), "cond %.1e" % cond, "FAIL", fails, flush=True)
        except Exception as e:
            bad.append(case)
            if verbose: print("CASE", case, dict(N=N, D=D, kinds=kinds, noise=noise), "EXC", type(e).__name__, e, flush=True)
    return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    failing = run(HipGP(0), seed, n)
    print("dense fuzz: %d cases, %d outside tolerance" % (n, len(failing)))
