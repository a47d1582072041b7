"""Random-shape fuzz of precision = 32: no failures, bound = fp64 bound unless the fp32 Gram matrix was accepted."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as go
from edrgp_amd.backend import HipGP
kinds_all = ["RBF", "Matern32", "Matern52"]


def run(h, seed=0, n_cases=40, verbose=True, f32_phi=0):
    rng = np.random.default_rng(seed)
    bad = []
    stats = {1: 0, 2: 0, 3: 0}
    for case in range(n_cases):
        D = int(rng.choice([1, 3, 8, 20, 33, 64, 70]))
        M = int(rng.integers(1, 300))
        N = int(rng.integers(max(M, 2), 1500))
        kind = kinds_all[int(rng.integers(0, 3))]
        var, ls, noise = float(rng.uniform(0.3, 2.0)), float(rng.uniform(0.8, 3.0) * np.sqrt(D)), float(10 ** rng.uniform(-2, 0))
        chunk = int(rng.choice([128, 300, 131072]))
        X, y, Z = go.synthetic_problem(N, D, M, seed=3000 + case)
        try:
            out = {}
            for prec in (64, 32):
                h.set_option("precision", prec)
                h.set_option("f32_phi", f32_phi)
                h.set_option("chunk_rows", chunk)
                h.set_data(X, y); h.set_inducing(Z)
                h.set_kernel([kind], [var], np.full((1, D), ls), noise)
                g = h.sparse_lml_grad()
                md = h.get("MODE", (3,))
                mode = int(md[0]); rho = md[1]; pr2 = md[2]
                h.sparse_posterior()
                mu, _ = h.predict(X[:30])
                out[prec] = (g, mode, mu)
            (g64, _, mu64), (g32, mode, mu32) = out[64], out[32]
            stats[mode] += 1
            gscale = max(abs(g64[3]), np.abs(g64[1]).max(), np.abs(g64[2]).max())
            errs = dict(lml=abs(g32[0] - g64[0]) / abs(g64[0]),
                        dvar=np.abs(g32[1] - g64[1]).max() / gscale, dlen=np.abs(g32[2] - g64[2]).max() / gscale,
                        dnoise=abs(g32[3] - g64[3]) / gscale,
                        dZ=np.abs(g32[4] - g64[4]).max() / max(np.abs(g64[4]).max(), 1e-3 * gscale),
                        mean=np.abs(mu32 - mu64).max() / max(np.abs(mu64).max(), 1e-12))
            if mode == 3:
                tol = dict(lml=1e-2, dvar=5e-2, dlen=5e-2, dnoise=1e-1, dZ=0.5, mean=1e-1)
            else:
                tol = dict(lml=1e-10, dvar=2e-2, dlen=2e-2, dnoise=1e-9, dZ=0.3, mean=1e-9)
            fails = {k: float(v) for k, v in errs.items() if not (v <= tol[k])}
            if verbose and mode == 3: print("  mode3 case", case, "rho %.2e pr2 %.1f lml err %.2e" % (rho, pr2, errs["lml"]), flush=True)
            if fails:
                bad.append(case)
                if verbose: print("CASE", case, dict(N=N, M=M, D=D, kind=kind, noise=noise, chunk=chunk, mode=mode), "rho %.2e pr2 %.1f" % (rho, pr2), "FAIL", fails, flush=True)
        except Exception as e:
            bad.append(case)
            if verbose: print("CASE", case, dict(N=N, M=M, D=D, kind=kind, noise=noise, chunk=chunk), "EXC", type(e).__name__, e, flush=True)
        finally:
            h.set_option("f32_phi", 0)
            h.set_option("precision", 64)
            h.set_option("chunk_rows", 131072)
    if verbose: print("modes that ran with precision 32:", stats)
    return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    failing = run(HipGP(0), seed, n, f32_phi=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    print("fp32 fuzz: %d cases, %d outside tolerance" % (n, len(failing)))
