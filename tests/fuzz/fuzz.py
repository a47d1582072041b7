"""Random-shape parity fuzz against the CPU oracle (bound, gradients, posterior mean / Jacobian)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gp_oracle as go
from edrgp_amd.backend import HipGP
kinds_all = ["RBF", "Matern32", "Matern52"]


def run(h, seed=0, n_cases=60, modes=(0, 0, 1, 2), verbose=True, large=False):
  """Returns the list of failing cases (empty = all within tolerance).  ``large``: shapes that reach the persistent
  grids, the k-segmented / length-ranked M x M x M products and the chunked row pass (N in the tens of thousands,
  M beyond 1024, not multiples of the tile sizes); the oracle needs seconds per case there."""
  rng = np.random.default_rng(seed)
  bad = []
  for case in range(n_cases):
      D = int(rng.integers(1, 10)) if case % 5 else int(rng.integers(33, 45))
      M = int(rng.integers(1, 200))
      N = int(rng.integers(max(M, 2), 900))
      if case % 7 == 3:        # inducing sets beyond 256: phantom padding to a multiple of 128, the (M + 1)-row bound-only path
          M = int(rng.integers(257, 520))
          N = int(rng.integers(M, 1600))
      if large:
          M = int(rng.integers(1024, 2300))
          N = int(rng.integers(max(20000, 28_000_000 // M), 52000))
          D = int(rng.integers(2, 33)) if case % 3 else int(rng.integers(33, 65))
      nparts = 1 if rng.random() < 0.7 else 2
      if large and case % 2 == 0:
          nparts = 1                      # the fused gradient epilogue is the single-part path
      kinds = [kinds_all[int(rng.integers(0, 3))] for _ in range(nparts)]
      opts = [dict(variance=float(rng.uniform(0.3, 2.0)), lengthscale=float(rng.uniform(0.8, 3.0) * np.sqrt(D))) for _ in range(nparts)]
      noise = float(10 ** rng.uniform(-2, 0))
      chunk = int(rng.choice([64, 100, 257, 131072])) if not large else int(rng.choice([16384, 30000, 131072, 131072]))
      mode = int(rng.choice(list(modes)))
      X, y, Z = go.synthetic_problem(N, D, M, seed=1000 + case)
      kern = go.make_kernel(kinds, D, opts)
      inf = go.vardtc_inference(kern, X, Z, noise, y[:, None])
      kg, dn, dZ = go.sparse_gradients(kern, X, Z, inf)
      ref = inf["log_marginal"][0, 0]
      h.set_option("chunk_rows", chunk); h.set_option("stats_mode", mode)
      try:
          h.set_data(X, y); h.set_inducing(Z)
          h.set_kernel(kinds, [o["variance"] for o in opts], np.array([np.full(D, o["lengthscale"]) for o in opts]), noise)
          lml_only = h.sparse_lml()          # bound-only evaluations take their own route through the second factorisation
          lml, dvar, dlen, dnoise, dZd = h.sparse_lml_grad()
          h.sparse_posterior()
          mu, var = h.predict(X[:50])
          jac = h.predict_grad(X[:50])
          mu_ref, var_ref = go.raw_predict(kern, X[:50], Z, inf["woodbury_vector"], inf["woodbury_inv"])
          jac_ref = go.predictive_gradients_mean(kern, X[:50], Z, inf["woodbury_vector"])[:, :, 0]
          gscale = max(abs(dn), max(abs(g[0]) for g in kg), max(np.abs(g[1]).max() for g in kg))
          # the bound is a sum of N terms of order one that can cancel to a small number: its error is measured
          # against max(|bound|, N)
          errs = dict(lml=max(abs(lml - ref), abs(lml_only - ref)) / max(abs(ref), float(N)),
                      dvar=max(abs(dvar[p] - kg[p][0]) for p in range(nparts)) / gscale,
                      dlen=max(abs(dlen[p].sum() - np.sum(kg[p][1])) for p in range(nparts)) / gscale,
                      dnoise=abs(dnoise - dn) / gscale,
                      dZ=np.abs(dZd - dZ).max() / max(np.abs(dZ).max(), 1e-12),
                      mean=np.abs(mu - mu_ref[:, 0]).max() / max(np.abs(mu_ref).max(), 1e-12),
                      var=np.abs(var - var_ref[:, 0]).max() / max(1.0, np.abs(var_ref).max()),
                      jac=np.abs(jac - jac_ref).max() / max(np.abs(jac_ref).max(), 1e-12))
          # both sides lose digits in proportion to cond(Kmm) (explicit inverses on the device, LAPACK solves in the
          # oracle): scale the tolerances by it; dZ is measured against the overall gradient scale as well, it is
          # ~0 when the inducing inputs saturate the input space
          cond = np.linalg.cond(kern.K(Z) + 1e-8 * np.eye(M))
          amp = max(1.0, cond / 1e6)
          errs["dZ"] = np.abs(dZd - dZ).max() / max(np.abs(dZ).max(), 1e-3 * gscale)
          tol = {k: v * amp for k, v in dict(lml=1e-9, dvar=1e-6, dlen=1e-6, dnoise=1e-6, dZ=1e-5, mean=1e-7, var=1e-6, jac=1e-7).items()}
          fails = {k: float(v) for k, v in errs.items() if not (v <= tol[k])}
          if fails:
              bad.append(case)
              if verbose: print("CASE", case, dict(N=N, M=M, D=D, kinds=kinds, noise=noise, chunk=chunk, mode=mode), "cond %.1e" % cond, "FAIL", fails, flush=True)
      except Exception as e:
          bad.append(case)
          if verbose: print("CASE", case, dict(N=N, M=M, D=D, kinds=kinds, noise=noise, chunk=chunk, mode=mode), "EXC", type(e).__name__, e, flush=True)
  h.set_option("chunk_rows", 131072)
  h.set_option("stats_mode", 0)
  return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    # forced Gram form (mode 1) is left out of the large shapes: their Kmm reaches cond 1e5 .. 1e11, where that form
    # loses the digits the automatic rule avoids by switching
    failing = run(HipGP(0), seed, n, modes=(0, 0, 2) if large else (0, 0, 1, 2), large=large)
    print("fuzz: %d cases, %d outside tolerance" % (n, len(failing)))
