"""Random-shape parity of both EP classifiers (dense EP and EPDTC) against the oracle, same site orders.
    python tests/fuzz/fuzz_ep.py [seed] [cases]
Test infrastructure (imports the oracle); run on a GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as go

KINDS = ["RBF", "Matern32", "Matern52"]


class Orders:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.used = []

    def __call__(self, n):
        o = self.rng.permutation(n)
        self.used.append(o)
        return o


def one_case(h, rng, verbose=False):
    sparse = bool(rng.integers(0, 3))                       # 2 of 3 cases: EPDTC
    N = int(rng.integers(20, 260))
    D = int(rng.integers(1, 7))
    nparts = 1 if rng.random() < 0.75 else 2
    kinds = [KINDS[int(rng.integers(0, 3))] for _ in range(nparts)]
    ard = rng.random() < 0.4
    opts, ls = [], []
    for _ in range(nparts):
        l = (0.5 + 2.0 * rng.random(D if ard else 1)) * np.sqrt(D)
        opts.append(dict(variance=float(0.3 + 2.0 * rng.random()), lengthscale=l if ard else float(l[0]), ARD=ard))
        ls.append(np.broadcast_to(l, (D,)).copy())
    kern = go.make_kernel(kinds, D, opts)
    X = rng.standard_normal((N, D))
    f = np.sin(1.5 * X[:, 0]) + (0.6 * X[:, 1 % D] if D > 1 else 0.0)
    y = (f + 0.4 * rng.standard_normal(N) > 0.1 * rng.standard_normal()).astype(float)
    if y.min() == y.max():
        y[0] = 1.0 - y[0]
    desc = dict(sparse=sparse, N=N, D=D, kinds=kinds, ard=ard)
    h.set_data(X, y)
    h.set_kernel(kinds, [o["variance"] for o in opts], np.array(ls), 1.0)
    orders = Orders(int(rng.integers(0, 1 << 30)))
    if sparse:
        M = int(rng.integers(1, min(N, 70) + 1))
        Z = X[rng.permutation(N)[:M]].copy()
        if rng.random() < 0.3:
            Z = Z + 0.1 * rng.standard_normal(Z.shape)
        desc["M"] = M
        desc["cond_Kmm"] = float(np.linalg.cond(kern.K(Z)))
        sites = go.expectation_propagation_dtc(kern.K(Z), kern.K(Z, X), y, order_fn=orders, max_iters=60)
        h.set_inducing(Z)
        h.epdtc_begin(y)
        sweep = h.epdtc_sweep
    else:
        sites = go.expectation_propagation(kern.K(X), y, order_fn=orders, max_iters=60)
        h.ep_begin(y)
        sweep = h.ep_sweep
    for k in range(sites["sweeps"]):
        d_tau, d_v = sweep(orders.used[k])
    logZt = h.ep_finish()
    tau, v = h.ep_get_sites()
    errs = {"tau": np.abs(tau - sites["tau"]).max() / np.abs(sites["tau"]).max(),
            "v": np.abs(v - sites["v"]).max() / max(np.abs(sites["v"]).max(), 1e-300),
            "logZt": abs(logZt - sites["log_Z_tilde"]) / max(abs(sites["log_Z_tilde"]), float(N))}
    dev_sites = dict(tau=tau, v=v, log_Z_tilde=logZt)
    if sparse:
        inf = go.epdtc_inference(kern, X, Z, dev_sites)
        lml, dvar, dlen, dZ = h.epdtc_lml_grad()
        kg, _, dZ_ref = go.sparse_gradients(kern, X, Z, inf)
        errs["dZ"] = np.abs(dZ - dZ_ref).max() / max(1.0, np.abs(dZ_ref).max())
        h.epdtc_posterior()
        P = Z
    else:
        inf = go.ep_inference(kern, X, dev_sites)
        lml, dvar, dlen = h.ep_lml_grad()
        kg = kern.param_gradients_full(inf["dL_dK"], X, None)
        h.ep_posterior()
        P = X
    errs["lml"] = abs(lml - inf["log_marginal"]) / max(abs(inf["log_marginal"]), float(N))
    gs = max(max(abs(g[0]), np.abs(g[1]).max()) for g in kg) + 1e-300
    errs["dvar"] = max(abs(dvar[p] - kg[p][0]) for p in range(nparts)) / gs
    errs["dlen"] = max(np.abs((dlen[p] if ard else dlen[p].sum()) - kg[p][1]).max() for p in range(nparts)) / gs
    Xs = np.vstack([rng.standard_normal((17, D)), X[:5]])
    mu_ref, var_ref = go.raw_predict(kern, Xs, P, inf["woodbury_vector"], inf["woodbury_inv"])
    mu, vr = h.predict(Xs)
    errs["mean"] = np.abs(mu - mu_ref[:, 0]).max() / max(1.0, np.abs(mu_ref).max())
    errs["var"] = np.abs(vr - var_ref[:, 0]).max() / max(1.0, np.abs(var_ref).max())
    jac_ref = go.predictive_gradients_mean(kern, Xs, P, inf["woodbury_vector"])[:, :, 0]
    errs["jac"] = np.abs(h.predict_grad(Xs) - jac_ref).max() / max(1.0, np.abs(jac_ref).max())
    tol = dict(tau=1e-6, v=1e-6, logZt=1e-7, lml=1e-8, dvar=1e-6, dlen=1e-6, dZ=1e-6, mean=1e-7, var=1e-7, jac=1e-7)
    # EPDTC factors kern.K(Z) WITHOUT jitter (as upstream): with a numerically singular K(Z, Z) (cond > 1e12: Z = all of
    # X, or 48 points of a smooth 2-D kernel) Lm^-1 has entries of 1e6 and more and every implementation's sites carry
    # that noise -- those cases are only held to 1e-2, beyond cond 1e15 to finiteness and the order of magnitude
    cond = desc.get("cond_Kmm", 1.0)
    lim = (lambda k: tol[k]) if cond <= 1e12 else ((lambda k: 1e-2) if cond <= 1e15 else (lambda k: 0.5))
    bad = {k: float(e) for k, e in errs.items() if not (e <= lim(k))}
    if not (np.all(np.isfinite(tau)) and np.all(tau > 0) and np.isfinite(lml)):
        bad["finite"] = 1.0
    if verbose and bad:
        print("FAIL", desc, bad, flush=True)
    return desc, bad


def run(h, seed=0, n_cases=40, verbose=False):
    rng = np.random.default_rng(seed)
    failing = []
    for i in range(n_cases):
        desc, bad = one_case(h, rng, verbose)
        if bad:
            failing.append((i, desc, bad))
    return failing


if __name__ == "__main__":
    from edrgp_amd.backend import HipGP
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad = run(HipGP(0), seed, cases, verbose=True)
    print("EP fuzz: %d cases, failing: %s" % (cases, bad))
