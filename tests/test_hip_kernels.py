"""GPU parity tests for the HIP building blocks (through the C ABI debug hooks).

Every result is compared with numpy / the CPU oracle on the same seeded inputs.
Tolerances are fp64 rounding-level bounds, stated per test.
"""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_mfma_f64_fragment_layout(hip):
    # exact small-integer data, asymmetric A and B: any lane-map mistake shows
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, size=(16, 4)).astype(np.float64)
    B = rng.integers(-8, 9, size=(4, 16)).astype(np.float64)
    C = hip.dbg_mfma_layout(A, B)
    assert np.array_equal(C, A @ B)


@pytest.mark.parametrize("transA", [False, True])
@pytest.mark.parametrize("transB", [False, True])
@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 128, 32), (200, 136, 72), (37, 53, 29),
                                   (12, 12, 12), (1, 7, 3), (384, 256, 160), (513, 130, 257)])
def test_gemm_variants(hip, transA, transB, shape):
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((K, M) if transA else (M, K))
    B = rng.standard_normal((N, K) if transB else (K, N))
    C0 = rng.standard_normal((M, N))
    ref = 0.75 * (A.T if transA else A) @ (B.T if transB else B) - 0.5 * C0
    out = hip.dbg_gemm(A, B, transA, transB, alpha=0.75, beta=-0.5, C=C0)
    assert rel(out, ref) < 1e-13


def test_gemm_large_square(hip):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((1024, 768))
    B = rng.standard_normal((768, 1152))
    assert rel(hip.dbg_gemm(A, B), A @ B) < 1e-13


def test_gemm_splitk_long_inner(hip):
    # the T^T [X|1|X^2] shape: small output, long inner dimension -> split-K slabs
    rng = np.random.default_rng(6)
    A = rng.standard_normal((9000, 96))
    B = rng.standard_normal((9000, 17))
    assert rel(hip.dbg_gemm(A, B, transA=True), A.T @ B) < 1e-12


@pytest.mark.parametrize("N,M", [(64, 16), (1000, 12), (5000, 256), (4096, 384), (3001, 130), (20000, 512)])
def test_syrk_phi(hip, N, M):
    rng = np.random.default_rng(N + M)
    K = rng.standard_normal((N, M))
    Phi = hip.dbg_syrk(K)
    ref = K.T @ K
    assert rel(Phi, ref) < 1e-12
    assert np.array_equal(Phi, Phi.T)          # mirrored, bitwise symmetric


def _spd(n, seed, cond=1e4):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.geomspace(1.0, cond, n)
    return (Q * ev) @ Q.T


@pytest.mark.parametrize("n", [1, 5, 64, 65, 100, 128, 200, 257, 512, 1000])
def test_potrf(hip, n):
    A = _spd(n, n)
    L, info = hip.dbg_potrf(A)
    assert info == 0
    ref = np.linalg.cholesky(A)
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    assert rel(L, ref) < 1e-11
    assert rel(L @ L.T, A) < 1e-13


@pytest.mark.parametrize("n", [3100, 3584, 4100])
def test_potrf_large(hip, n):
    """n > 3072: the two-launch-per-step path (the fused step kernel is used up to 3072)"""
    A = _spd(n, n, cond=1e3)
    L, info = hip.dbg_potrf(A)
    assert info == 0
    ref = np.linalg.cholesky(A)
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    assert rel(L, ref) < 1e-11
    B = A.copy()
    B[3090, 3090] = -1.0
    _, info = hip.dbg_potrf(B)
    assert info == 3091


@pytest.mark.parametrize("n", [1, 64, 65, 100, 128, 129, 200, 257, 512, 1000, 2048, 2049, 3072, 3100])
def test_potrf_with_inverse(hip, n):
    """Factor and inverse of the factor out of the same launches (n in (64, 3072]: the appended identity rows ride
    through the fused steps; outside: factorisation + recursive-doubling inverse).  The factor must be bit-identical
    to the factorisation alone; every entry of the inverse is written (the output buffer starts as NaN)."""
    A = _spd(n, n, cond=1e3)
    L, Li, info = hip.dbg_potrf_inv(A)
    assert info == 0
    L0, _ = hip.dbg_potrf(A)
    assert np.array_equal(L, L0)
    assert np.all(np.isfinite(Li))
    assert np.array_equal(np.triu(Li, 1), np.zeros_like(Li))
    ref = sla.solve_triangular(np.linalg.cholesky(A), np.eye(n), lower=True)
    assert rel(Li, ref) < 1e-10
    assert rel(Li @ L, np.eye(n)) < 1e-11
    Li2 = hip.dbg_trtri(L)                   # the recursive-doubling inverse of the same factor
    assert rel(Li, Li2) < 1e-11


def test_potrf_reports_failure(hip):
    A = _spd(130, 3)
    A[100, 100] = -1.0
    _, info = hip.dbg_potrf(A)
    assert info == 101                         # 1-based index of the first bad pivot


@pytest.mark.parametrize("n", [1, 7, 64, 65, 128, 129, 192, 200, 320, 512, 1000])
def test_trtri(hip, n):
    L = np.linalg.cholesky(_spd(n, 100 + n))
    Li = hip.dbg_trtri(L)
    assert np.array_equal(np.triu(Li, 1), np.zeros_like(Li))
    ref = sla.solve_triangular(L, np.eye(n), lower=True)
    assert rel(Li, ref) < 1e-11
    assert rel(Li @ L, np.eye(n)) < 1e-12


KERNEL_CASES = [
    (["RBF"], [dict(variance=0.7, lengthscale=1.3)]),
    (["Matern32"], [dict(variance=1.4, lengthscale=0.8)]),
    (["Matern52"], [dict(variance=0.9, lengthscale=2.1)]),
    (["RBF"], [dict(variance=0.7, lengthscale=[0.5, 1.0, 2.0, 4.0, 1.5], ARD=True)]),
    (["RBF", "Matern52"], [dict(variance=0.7, lengthscale=1.3), dict(variance=0.2, lengthscale=0.6)]),
]


def _set_kernel(hip, kern, noise):
    D = kern.input_dim
    kinds = [p.kind for p in kern.parts]
    var = [p.variance for p in kern.parts]
    ls = np.array([np.broadcast_to(p.lengthscale, (D,)) for p in kern.parts])
    hip.set_kernel(kinds, var, ls, noise)


@pytest.mark.parametrize("kinds,opts", KERNEL_CASES)
@pytest.mark.parametrize("N,M,D", [(50, 12, 5), (300, 70, 5), (1000, 129, 5)])
def test_kernel_matrix(hip, kinds, opts, N, M, D):
    rng = np.random.default_rng(N)
    X = rng.standard_normal((N, D))
    Z = X[rng.permutation(N)[:M]].copy()       # shared points -> exact zeros of r
    kern = go.make_kernel(kinds, D, opts)
    hip.set_data(X, np.zeros(N))
    _set_kernel(hip, kern, 0.1)
    Knm = hip.dbg_kernel_matrix(X, Z)
    assert rel(Knm, kern.K(X, Z)) < 1e-13
    Kmm = hip.dbg_kernel_matrix(Z)
    assert rel(Kmm, kern.K(Z)) < 1e-13
    assert np.array_equal(np.diag(Kmm), kern.Kdiag(Z))


@pytest.mark.parametrize("D", [1, 2, 33, 64, 70])
def test_kernel_matrix_feature_dims(hip, D):
    rng = np.random.default_rng(D)
    X = rng.standard_normal((257, D))
    Z = rng.standard_normal((65, D))
    kern = go.make_kernel(["Matern52"], D, [dict(variance=1.1, lengthscale=np.sqrt(D))])
    hip.set_data(X, np.zeros(257))
    _set_kernel(hip, kern, 0.1)
    assert rel(hip.dbg_kernel_matrix(X, Z), kern.K(X, Z)) < 1e-13


def test_calibration_runs(hip):
    tf, bw = hip.calib()
    print("fp64 MFMA issue rate: %.1f TFLOP/s; HBM copy: %.0f GB/s" % (tf, bw))
    assert tf > 10.0 and bw > 1000.0


def test_covariance_exp_against_libm(hip):
    """The covariance epilogues use their own exp for non-positive arguments (kfun.h): K(X, X2) for RBF /
    Matern must agree with the oracle's libm-based values over the useful range.  Both sides form r^2 as
    |x|^2 + |z|^2 - 2 x.z (GPy's formula), whose rounding error ~eps (|x|^2 + |z|^2) <= 1e-13 here becomes a
    relative error of the covariance; the exp itself contributes ~1e-16."""
    from oracle import gp_oracle as go
    rng = np.random.default_rng(11)
    X = rng.standard_normal((300, 3)) * np.array([0.01, 1.0, 6.0])      # r^2 from ~0 to a few hundred
    X2 = rng.standard_normal((200, 3)) * np.array([0.01, 1.0, 6.0])
    for kind in ("RBF", "Matern32", "Matern52"):
        kern = go.make_kernel([kind], 3, [dict(variance=1.7, lengthscale=0.9)])
        hip.set_data(X, np.zeros(300))
        hip.set_kernel([kind], [1.7], np.full((1, 3), 0.9), 0.1)
        K = hip.dbg_kernel_matrix(X, X2)
        ref = kern.K(X, X2)
        assert ref.min() < 1e-20 and ref.max() > 0.1
        assert np.abs(K / ref - 1.0).max() < 3e-13
