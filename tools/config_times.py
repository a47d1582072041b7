"""Per-evaluation times at the BASELINE.json config shapes (one MI355X)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
from edrgp_amd.sweep import projection_sweep

h = HipGP(0)

def timeit(f, n=9):
    """Median of n calls after a warm-up (the one-time ~70 ms queue / allocation cost of a fresh process, clock ramp):
    a mean over five calls once carried a single 60 ms hiccup of the box into the log as "C2 bound 17 ms"."""
    for _ in range(6):
        f()
    ts = []
    for _ in range(n):
        h.sync()
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3

for name, N, M, D, kind in [("C2", 100000, 1024, 16, "RBF"), ("Cm", 100000, 2048, 32, "RBF"),
                            ("C3 shard (1/8 of 1M rows)", 125000, 2048, 32, "Matern52"),
                            ("C3 whole on one GPU", 1000000, 2048, 32, "Matern52")]:
    X, y, Z = bench.synthetic(N, D, M, seed=7)
    h.set_data(X, y); h.set_inducing(Z)
    h.set_kernel([kind], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
    t_l = timeit(h.sparse_lml)
    t_g = timeit(h.sparse_lml_grad)
    h.sparse_posterior()
    t_p = timeit(lambda: h.predict_train(True, True), 3)
    t_pg = timeit(lambda: h.predict_train_gram(True), 3)
    print("%-28s N=%d M=%d D=%d %s: bound %.2f ms (%.1f/s), bound+grad %.2f ms (%.1f/s), predict+predict_gradient on N rows %.1f ms "
          "(N x D Jacobian copied to the host; %.1f ms with its D x D Gram matrix reduced on the device instead, as the EDR loop takes it)"
          % (name, N, M, D, kind, t_l, 1e3 / t_l, t_g, 1e3 / t_g, t_p, t_pg), flush=True)

# C4: one candidate projection x 16 length-scales at full N (unit of work: bound + N-point mean gradient)
N, M, D = 200000, 1024, 64
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)); y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
Zr = rng.permutation(N)[:M]
projection_sweep(X, y, Zr, [np.linalg.qr(rng.standard_normal((D, D)))[0]], n_scales=16, want_gram=True, hip=h)   # workspaces (16 K_g kept: 26 GB) allocated once
for c in (0, 32, 63):
    W = np.linalg.qr(rng.standard_normal((D, D - c)))[0]
    t0 = time.perf_counter()
    out = projection_sweep(X, y, Zr, [W], n_scales=16, want_gram=True, hip=h)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    out2 = projection_sweep(X, y, Zr, [W], n_scales=16, want_gradients=True, hip=h)
    dt2 = time.perf_counter() - t0
    g = out2[3]["grad"]
    err = np.abs(out[3]["gram"] - g.T @ g).max() / np.abs(g.T @ g).max()
    print("C4 candidate d_c=%d: 16 (bound + N-point gradient) evaluations: %.1f ms each with the gradient Gram matrix reduced on the device, "
          "%.1f ms each with the N x d_c gradients copied to the host and back-projected there (Gram rel. diff %.1e); best bound %.1f"
          % (D - c, dt * 1e3 / 16, dt2 * 1e3 / 16, err, max(o["bound"] for o in out)), flush=True)
