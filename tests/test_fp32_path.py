"""fp32 row pass (BASELINE config 5; include/edrgp_hip.h option "precision" = 32).

Tolerances are the ones the path achieves against the fp64 CPU oracle and are written here:
the fp32 MFMA is an exact k-ordered fmaf chain, the slabs are summed in fp64, every M x M
object is fp64.  Default: only the gradient contraction is fp32 -- the bound is the fp64 one
(1e-10), hyper-parameter gradients 5e-3, dZ 5e-2 (max-norm relative).  With "f32_phi" = 1 and the
fp32 Gram matrix accepted by the guard (mode 3): bound 1e-4 rel, hyper-parameter gradients 2e-3,
dZ 3e-2, posterior mean / gradient 1e-2.
"""
import numpy as np
import pytest

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


@pytest.mark.parametrize("m,n,k,ta,tb", [(200, 150, 333, False, False), (129, 257, 64, False, True),
                                         (300, 70, 1000, True, False), (64, 64, 40, True, True),
                                         (1, 5, 3, False, False), (130, 130, 31, True, False)])
def test_gemm32_layouts(hip, m, n, k, ta, tb):
    rng = np.random.default_rng(m + n + k)
    A = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
    B = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
    C = hip.dbg_gemm32(A, B, ta, tb)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    assert np.abs(C - ref).max() <= 2e-5 * np.sqrt(k)


@pytest.mark.parametrize("rows,M", [(5000, 384), (777, 130), (40000, 256)])
def test_syrk32_fp64_slab_sums(hip, rows, M):
    rng = np.random.default_rng(rows)
    K = rng.random((rows, M)).astype(np.float32)
    P = hip.dbg_syrk32(K)
    ref = K.astype(np.float64).T @ K.astype(np.float64)
    assert np.array_equal(P, P.T)
    assert _rel(P, ref) <= 2e-6


def _problem(N, M, D, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    y = np.tanh(X @ (np.ones(D) / np.sqrt(D))) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    Z = X[rng.permutation(N)[:M]].copy()
    return X, y, Z


def _oracle(kind, X, y, Z, var, ls, noise):
    D = X.shape[1]
    kern = orc.make_kernel([kind], D, [dict(variance=var, lengthscale=list(ls), ARD=True)])
    inf = orc.vardtc_inference(kern, X, Z, noise, y[:, None])
    kg, dn, dZ = orc.sparse_gradients(kern, X, Z, inf)
    mu, _ = orc.raw_predict(kern, X[:64], Z, inf["woodbury_vector"], inf["woodbury_inv"])
    jac = orc.predictive_gradients_mean(kern, X[:64], Z, inf["woodbury_vector"])[:, :, 0]
    return dict(lml=inf["log_marginal"][0, 0], dvar=kg[0][0], dlen=kg[0][1], dnoise=dn, dZ=dZ), mu[:, 0], jac


@pytest.mark.parametrize("kind", ["RBF", "Matern52"])
def test_fp32_gram_accepted(hip, kind):
    # D = 64 with l ~ sqrt(D): Kmm is well conditioned, the guard accepts the fp32 Gram matrix
    N, M, D = 6000, 256, 64
    X, y, Z = _problem(N, M, D, 11)
    ls = np.full(D, 0.9 * np.sqrt(D))
    hip.set_option("precision", 32)
    hip.set_option("f32_phi", 1)
    try:
        hip.set_data(X, y); hip.set_inducing(Z)
        hip.set_kernel([kind], [1.3], ls[None, :], 0.05)
        lml, dvar, dlen, dnoise, dZ = hip.sparse_lml_grad()
        mode = hip.get("MODE", (2,))
        hip.sparse_posterior()
        mean, _ = hip.predict(X[:64], want_var=False)
        jac = hip.predict_grad(X[:64])
    finally:
        hip.set_option("f32_phi", 0)
        hip.set_option("precision", 64)
    assert int(mode[0]) == 3 and 0.0 < mode[1] <= 1.0
    g, mu, J = _oracle(kind, X, y, Z, 1.3, ls, 0.05)
    assert abs(lml - g["lml"]) <= 1e-4 * abs(g["lml"])
    assert abs(dvar[0] - g["dvar"]) <= 2e-3 * abs(g["dvar"])
    assert _rel(dlen[0], g["dlen"]) <= 2e-3
    assert abs(dnoise - g["dnoise"]) <= 2e-3 * abs(g["dnoise"])
    assert _rel(dZ, g["dZ"]) <= 3e-2
    assert _rel(mean, mu) <= 1e-2
    assert _rel(jac, J) <= 1e-2


def test_fp32_gram_rejected_falls_back_to_fp64_bound(hip):
    # D = 8: Kmm is ill conditioned, the fp32 Gram matrix must be rejected (Cholesky failure or guard)
    N, M, D = 3000, 256, 8
    X, y, Z = _problem(N, M, D, 5)
    ls = np.full(D, 0.9 * np.sqrt(D))
    hip.set_option("precision", 64)
    hip.set_data(X, y); hip.set_inducing(Z)
    hip.set_kernel(["RBF"], [1.3], ls[None, :], 0.05)
    ref = hip.sparse_lml_grad()
    hip.set_option("precision", 32)
    hip.set_option("f32_phi", 1)
    try:
        hip.set_data(X, y); hip.set_inducing(Z)
        hip.set_kernel(["RBF"], [1.3], ls[None, :], 0.05)
        first = hip.sparse_lml_grad()
        mode1 = int(hip.get("MODE", (2,))[0])
        second = hip.sparse_lml()
        mode2 = int(hip.get("MODE", (2,))[0])
    finally:
        hip.set_option("f32_phi", 0)
        hip.set_option("precision", 64)
    assert mode1 in (1, 2) and mode2 in (1, 2)
    assert abs(first[0] - ref[0]) <= 1e-10 * abs(ref[0])          # the bound is the fp64 one
    assert abs(second - ref[0]) <= 1e-10 * abs(ref[0])
    assert abs(first[1][0] - ref[1][0]) <= 5e-3 * abs(ref[1][0])  # fp32 gradient contraction
    assert _rel(first[2], ref[2]) <= 5e-3
    assert abs(first[3] - ref[3]) <= 1e-10 * abs(ref[3])          # d/dnoise comes from the M x M scalars
    assert _rel(first[4], ref[4]) <= 5e-2


def test_fp32_chunked_rows_match_resident(hip):
    N, M, D = 5000, 256, 64
    X, y, Z = _problem(N, M, D, 3)
    ls = np.full(D, np.sqrt(D))
    out = []
    hip.set_option("precision", 32)
    hip.set_option("f32_phi", 1)
    try:
        for chunk in (131072, 1024):
            hip.set_option("chunk_rows", chunk)
            hip.set_data(X, y); hip.set_inducing(Z)
            hip.set_kernel(["RBF"], [1.0], ls[None, :], 0.05)
            out.append(hip.sparse_lml_grad())
    finally:
        hip.set_option("chunk_rows", 0)
        hip.set_option("f32_phi", 0)
        hip.set_option("precision", 64)
    a, b = out
    # different chunking = different fp32 accumulation chains of Phi: both are within the 1e-4 class of
    # the fp32 Gram form (1.2e-4 observed between them)
    assert abs(a[0] - b[0]) <= 5e-4 * abs(b[0])
    assert _rel(a[2], b[2]) <= 2e-3 and _rel(a[4], b[4]) <= 3e-2


def test_fp32_estimator_api(hip):
    from edrgp_amd.gp_model import SparseGaussianProcessRegressor
    rng = np.random.default_rng(0)
    X = rng.standard_normal((2000, 16))
    y = np.tanh(X[:, 0]) + 0.1 * rng.standard_normal(2000)
    np.random.seed(0)
    a = SparseGaussianProcessRegressor(num_inducing=128, precision="fp32").fit(X, y, max_iters=15)
    np.random.seed(0)
    b = SparseGaussianProcessRegressor(num_inducing=128).fit(X, y, max_iters=15)
    pa, pb = a.predict(X[:200]), b.predict(X[:200])
    assert np.corrcoef(pa, pb)[0, 1] > 0.999
    ga, gb = a.predict_gradient(X[:200]), b.predict_gradient(X[:200])
    # two 15-iteration optimiser paths at different precisions: the fitted models are close, not identical
    assert np.corrcoef(ga[:, 0], gb[:, 0])[0, 1] > 0.9
    with pytest.raises(ValueError):
        SparseGaussianProcessRegressor(precision="fp16").fit(X, y)
