"""The HIP path evaluates VarDTC in the lean Gram form:
Phi = K_mn K_nm, explicit triangular inverses, and gradients finished from
G~ = T^T [X | 1 | X.^2].  This CPU test restates exactly that operation
sequence in numpy (mirroring edr-gp_amd/csrc/api.hip::sparse_eval) and checks
it against the oracle's GPy-ordered VarDTC: it pins the algebra of the
formulation independently of any GPU.
"""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import gp_oracle as go


def h_factor(part, X, X2):
    """h = (dK/dr)/r with GPy's _inv_dist convention (0 where r == 0)."""
    r = part._scaled_dist(X, X2)
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.where(r != 0.0, part.dK_dr(r) / np.where(r != 0.0, r, 1.0), 0.0)
    return h


def tilde(X):
    return np.hstack([X, np.ones((X.shape[0], 1)), X * X])


def gram_sparse_eval(kern, X, y, Z, noise, allreduce=None):
    """`allreduce` (sum over ranks, in place semantics by return value) marks exactly
    the exchanges the multi-GPU path performs (api.hip: allreduce_sum calls):
    [N, y'y], [Phi | b], and the row-dependent gradient partials."""
    if allreduce is None:
        allreduce = lambda a: a      # noqa: E731
    N_local, D = X.shape
    M = Z.shape[0]
    beta = 1.0 / max(noise, 1e-8)
    K = kern.K(X, Z)
    N, yy = allreduce(np.array([float(N_local), float(y @ y)]))
    stats = allreduce(np.concatenate([(K.T @ K).ravel(), K.T @ y]))
    Phi, b = stats[:M * M].reshape(M, M), stats[M * M:]
    Kmm = kern.K(Z) + 1e-8 * np.eye(M)
    Lm = np.linalg.cholesky(Kmm)
    Li = sla.solve_triangular(Lm, np.eye(M), lower=True)
    A = beta * (Li @ Phi) @ Li.T
    trA = np.trace(A)
    LB = np.linalg.cholesky(A + np.eye(M))
    LBi = sla.solve_triangular(LB, np.eye(M), lower=True)
    v = Li @ b
    c = beta * (LBi @ v)
    u = LBi.T @ c
    wv = Li.T @ u
    data_fit = c @ c
    psi0sum = N * sum(p.variance for p in kern.parts)
    lml = (-0.5 * N * (np.log(2 * np.pi) - np.log(beta)) - 0.5 * beta * yy
           - 0.5 * (beta * psi0sum - trA) - np.sum(np.log(np.diag(LB))) + 0.5 * data_fit)
    R = LBi @ Li
    P = R.T @ R
    Kmmi = Li.T @ Li
    winv = Kmmi - P
    S2 = beta * (Kmmi - P - np.outer(wv, wv))
    dKmm = 0.5 * Kmmi - 0.5 * P - 0.5 * np.outer(wv, wv) - 0.5 * (Li.T @ A) @ Li
    W = K @ S2
    Q = W + beta * np.outer(y, wv)
    Xt, Zt = tilde(X), tilde(Z)
    dvar, dlen, dZ = [], [], np.zeros_like(Z)
    for part in kern.parts:
        l = np.broadcast_to(part.lengthscale, (D,))
        T = Q * h_factor(part, X, Z)
        Tm = dKmm * h_factor(part, Z, None)
        red = allreduce(np.concatenate([(T.T @ Xt).ravel(), [np.sum(Q * part.K(X, Z)) / part.variance]]))
        Gn, dv_nm = red[:-1].reshape(M, 2 * D + 1), red[-1]
        GmT, GmN = Tm.T @ Zt, Tm @ Zt                       # replicated (Kmm term)
        dv = dv_nm + np.sum(dKmm * part.K(Z)) / part.variance - 0.5 * beta * N
        dl = np.zeros(D)
        for q in range(D):
            z = Z[:, q]
            s_nm = np.sum(Gn[:, D + 1 + q] + Gn[:, D] * z * z - 2 * z * Gn[:, q])
            s_mm = np.sum(GmT[:, D + 1 + q] + GmT[:, D] * z * z - 2 * z * GmT[:, q])
            dl[q] = -(s_nm + s_mm) / l[q] ** 3
            dZ[:, q] += ((z * Gn[:, D] - Gn[:, q]) + (z * GmT[:, D] - GmT[:, q]) + (z * GmN[:, D] - GmN[:, q])) / l[q] ** 2
        dvar.append(dv)
        dlen.append(dl if part.ARD else np.array([dl.sum()]))
    sumA_DBi = M - np.sum(LBi * LBi) + data_fit - u @ u
    dnoise = (-0.5 * N * beta + 0.5 * yy * beta ** 2 + 0.5 * (psi0sum * beta ** 2 - trA * beta)
              + beta * (0.5 * sumA_DBi - data_fit))
    return dict(lml=lml, wv=wv, winv=winv, dvar=dvar, dlen=dlen, dnoise=dnoise, dZ=dZ, dKmm=dKmm)


CASES = [
    (["RBF"], [dict(variance=0.8, lengthscale=1.7)]),
    (["Matern32"], [dict(variance=1.3, lengthscale=2.0)]),
    (["Matern52"], [dict(variance=1.1, lengthscale=2.5)]),
    (["RBF"], [dict(variance=0.9, lengthscale=[1.5, 2.0, 2.5, 3.0], ARD=True)]),
    (["RBF", "Matern52"], [dict(variance=0.6, lengthscale=2.0), dict(variance=0.3, lengthscale=1.2)]),
]


@pytest.mark.parametrize("kinds,opts", CASES)
def test_gram_form_equals_vardtc(kinds, opts):
    N, M, D = 300, 40, 4
    X, y, Z = go.synthetic_problem(N, D, M, seed=11)
    kern = go.make_kernel(kinds, D, opts)
    inf = go.vardtc_inference(kern, X, Z, 0.05, y[:, None])
    kg, dn, dZ = go.sparse_gradients(kern, X, Z, inf)
    g = gram_sparse_eval(kern, X, y, Z, 0.05)
    # Algebraic identity; the Gram form's rounding error grows with cond(Kmm)
    # (the library switches to the stable form there), hence max-norm bounds.
    def mx(a, b):
        return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()

    ref = inf["log_marginal"][0, 0]
    assert abs(g["lml"] - ref) < 1e-9 * abs(ref)
    assert mx(g["wv"], inf["woodbury_vector"][:, 0]) < 1e-6
    assert mx(g["winv"], inf["woodbury_inv"]) < 1e-6
    assert mx(g["dKmm"], inf["dL_dKmm"]) < 1e-6
    for p, (gv, gl) in enumerate(kg):
        assert np.isclose(g["dvar"][p], gv, rtol=1e-7)
        assert mx(g["dlen"][p], gl) < 1e-7
    assert np.isclose(g["dnoise"], dn, rtol=1e-7)
    assert mx(g["dZ"], dZ) < 1e-6
