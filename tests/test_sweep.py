"""Config 4 (candidate-direction x length-scale sweep): every sweep entry equals the
oracle's bound / mean-gradient on the projected data; rank sharding partitions the
entries without overlap."""
import numpy as np
import pytest

from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def test_projection_sweep_matches_oracle_and_shards(hip):
    from edrgp_amd.sweep import length_scale_grid, projection_sweep
    rng = np.random.default_rng(3)
    N, D, M = 600, 6, 40
    X = rng.standard_normal((N, D))
    y = np.tanh(X[:, 0] - 0.5 * X[:, 1]) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    projections = [np.linalg.qr(rng.standard_normal((D, D - c)))[0] for c in range(3)]
    Zrows = rng.permutation(N)[:M]
    full = projection_sweep(X, y, Zrows, projections, n_scales=4, want_gradients=True, hip=hip)
    assert len(full) == 12
    for rec in full:
        W = projections[rec["candidate"]]
        Xp = X @ W
        kern = go.make_kernel(["RBF"], Xp.shape[1], [dict(variance=1.0, lengthscale=rec["lengthscale"])])
        inf = go.vardtc_inference(kern, Xp, Xp[Zrows], 0.01, y[:, None])
        ref = inf["log_marginal"][0, 0]
        assert abs(rec["bound"] - ref) <= 1e-9 * abs(ref)
        jac = go.predictive_gradients_mean(kern, Xp, Xp[Zrows], inf["woodbury_vector"])[:, :, 0]
        assert np.abs(rec["grad"] - jac).max() <= 1e-7 * max(1.0, np.abs(jac).max())
        assert rec["grad_full"].shape == (N, D)
        assert np.isclose(rec["lengthscale"], length_scale_grid(Xp.shape[1], 4)[rec["scale"]])
    parts = [projection_sweep(X, y, Zrows, projections, n_scales=4, rank=r, world_size=2, hip=hip) for r in range(2)]
    keys = [sorted((p["candidate"], p["scale"]) for p in part) for part in parts]
    assert not set(keys[0]) & set(keys[1]) and len(keys[0]) + len(keys[1]) == 12
    lookup = {(p["candidate"], p["scale"]): p["bound"] for p in full}
    for part in parts:
        for p in part:
            assert p["bound"] == lookup[(p["candidate"], p["scale"])]


def test_sweep_gram_and_device_projection(hip):
    """The gradient outer-product matrix reduced on the device equals grad^T grad of the copied
    gradients, its back-projection W G W^T equals grad_full^T grad_full, and data projected on the
    device (edrgp_set_data_projected) gives the bound of the host-projected data."""
    from edrgp_amd.sweep import projection_sweep
    rng = np.random.default_rng(5)
    N, D, M = 3000, 12, 64
    X = rng.standard_normal((N, D))
    y = np.tanh(X[:, 0] + 0.3 * X[:, 2]) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    projections = [np.linalg.qr(rng.standard_normal((D, D - c)))[0] for c in (0, 5, 11)]
    Zrows = rng.permutation(N)[:M]
    recs = projection_sweep(X, y, Zrows, projections, n_scales=3, want_gradients=True, want_gram=True, hip=hip)
    assert len(recs) == 9
    for rec in recs:
        g = rec["grad"]
        ref = g.T @ g
        assert np.abs(rec["gram"] - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300)
        assert np.abs(rec["gsum"] - g.sum(0)).max() <= 1e-11 * max(np.abs(g).sum(0).max(), 1e-300)
        full = rec["grad_full"].T @ rec["grad_full"]
        assert np.abs(rec["gram_full"] - full).max() <= 1e-11 * max(np.abs(full).max(), 1e-300)
    # device projection vs host projection of the same candidate
    W = projections[1]
    Xp = np.ascontiguousarray(X @ W)
    hip.set_data(Xp, y)
    hip.set_inducing(np.ascontiguousarray(Xp[Zrows]))
    hip.set_kernel(["RBF"], [1.0], np.full((1, Xp.shape[1]), recs[4]["lengthscale"]), 0.01)
    ref = hip.sparse_lml()
    assert abs(recs[4]["bound"] - ref) <= 1e-10 * abs(ref)
    with pytest.raises(ValueError):
        hip.set_data_projected(np.zeros((D + 1, 2)))          # W does not match the resident base rows


def test_sweep_concurrent_contexts_match_sequential(hip):
    """Two native contexts on their own host threads share the (candidate, scale) pairs: same records, same
    bounds bit for bit (the C ABI is thread-safe across contexts)."""
    from edrgp_amd.sweep import projection_sweep
    rng = np.random.default_rng(9)
    N, D, M = 4000, 10, 96
    X = rng.standard_normal((N, D))
    y = np.tanh(X[:, 1]) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    projections = [np.linalg.qr(rng.standard_normal((D, D - c)))[0] for c in (0, 4)]
    Zrows = rng.permutation(N)[:M]
    seq = projection_sweep(X, y, Zrows, projections, n_scales=5, want_gram=True, hip=hip)
    par = projection_sweep(X, y, Zrows, projections, n_scales=5, want_gram=True, concurrency=2, device=hip.device)
    assert [(r["candidate"], r["scale"]) for r in seq] == [(r["candidate"], r["scale"]) for r in par]
    for a, b in zip(seq, par):
        assert a["bound"] == b["bound"]
        assert np.array_equal(a["gram"], b["gram"])


@pytest.mark.parametrize("N,D,M,dcs,kind", [(3000, 12, 96, (12, 7), "RBF"), (5000, 10, 300, (10,), "RBF"),
                                             (3000, 12, 96, (12, 7), "Matern52"), (5000, 10, 300, (10,), "Matern32")])
def test_sweep_batched_scales_equal_per_unit_path(hip, N, D, M, dcs, kind):
    """edrgp_sweep_scales (all length-scales of a candidate as one batch: distances formed once, batched
    factorisations) gives the numbers of the per-unit path -- bound 1e-10, gradient Gram matrix 1e-8 -- and of the
    oracle; the grid spans both statistics forms (wide length-scales take the stable one)."""
    from edrgp_amd.sweep import projection_sweep
    rng = np.random.default_rng(21)
    X = rng.standard_normal((N, D))
    y = np.tanh(X[:, 0] + 0.3 * X[:, 2]) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    projections = [np.linalg.qr(rng.standard_normal((D, dc)))[0] for dc in dcs]
    Zrows = rng.permutation(N)[:M]
    bat = projection_sweep(X, y, Zrows, projections, n_scales=8, want_gram=True, hip=hip, kind=kind)
    seq = projection_sweep(X, y, Zrows, projections, n_scales=8, want_gram=True, hip=hip, batch=False, kind=kind)
    assert [(r["candidate"], r["scale"]) for r in bat] == [(r["candidate"], r["scale"]) for r in seq]
    forms = set()
    for a, b in zip(bat, seq):
        forms.add(a.get("form"))
        assert abs(a["bound"] - b["bound"]) <= 1e-10 * max(abs(b["bound"]), float(N)), (a["candidate"], a["scale"])
        assert np.abs(a["gram"] - b["gram"]).max() <= 1e-8 * np.abs(b["gram"]).max()
        assert np.abs(a["gsum"] - b["gsum"]).max() <= 1e-8 * max(np.abs(b["gsum"]).max(), np.sqrt(np.abs(b["gram"]).max()))
    assert 1 in forms and all(r.get("form") in (1, 2) for r in bat)       # every record came out of the batch
    if kind == "RBF":
        assert forms >= {1, 2}             # both statistics forms occurred in the batch
    # and against the oracle for one record per form
    for form in sorted(forms):
        rec = next(r for r in bat if r.get("form") == form)
        W = projections[rec["candidate"]]
        Xp = X @ W
        kern = go.make_kernel([kind], Xp.shape[1], [dict(variance=1.0, lengthscale=rec["lengthscale"])])
        inf = go.vardtc_inference(kern, Xp, Xp[Zrows], 0.01, y[:, None])
        ref = inf["log_marginal"][0, 0]
        assert abs(rec["bound"] - ref) <= 1e-9 * max(abs(ref), float(N))
        jac = go.predictive_gradients_mean(kern, Xp, Xp[Zrows], inf["woodbury_vector"])[:, :, 0]
        G = jac.T @ jac
        assert np.abs(rec["gram"] - G).max() <= 1e-6 * np.abs(G).max()
