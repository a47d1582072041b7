"""Error behaviour at the C ABI (include/edrgp_hip.h): every misuse returns a negative status with a message
(surfaced as ``EdrgpError``), never a crash, a hang or a silent number.  Non-finite inputs end in the
not-positive-definite status after the jitter ladder (GPy: ``LinAlgError`` out of ``jitchol``)."""
import numpy as np
import pytest

from edrgp_amd import _lib
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fresh():
    from edrgp_amd.backend import HipGP
    h = HipGP(0)
    yield h
    h.close()


def test_calls_out_of_order_report_state_errors(fresh):
    h = fresh
    with pytest.raises(_lib.EdrgpError) as e:
        h.sparse_lml()
    assert e.value.status < 0 and "set_data" in str(e.value)
    X, y, Z = go.synthetic_problem(100, 3, 10, seed=0)
    h.D = 3                                   # bypass the wrapper's own shape check: the library must refuse
    with pytest.raises(_lib.EdrgpError):
        h.set_inducing(Z)                     # before set_data: D unknown
    h.set_data(X, y)
    with pytest.raises(_lib.EdrgpError):
        h.sparse_lml()                        # no Z, no kernel
    h.set_inducing(Z)
    with pytest.raises(_lib.EdrgpError):
        h.sparse_lml_grad()                   # no kernel
    h.set_kernel(["RBF"], [1.0], np.full((1, 3), 1.0), 0.1)
    with pytest.raises(_lib.EdrgpError):
        h.predict(X[:5])                      # no posterior yet
    assert np.isfinite(h.sparse_lml())        # and the context is still usable afterwards
    h.sparse_posterior()
    assert np.all(np.isfinite(h.predict(X[:5])[0]))


def test_bad_arguments_are_refused(fresh):
    h = fresh
    X, y, Z = go.synthetic_problem(64, 2, 8, seed=1)
    h.set_data(X, y)
    h.set_inducing(Z)
    with pytest.raises(_lib.EdrgpError):
        h.set_option("no_such_option", 1.0)
    lib = h._lib
    import ctypes
    kinds = (ctypes.c_int * 1)(7)             # unknown kernel kind
    one = (ctypes.c_double * 1)(1.0)
    ls = (ctypes.c_double * 2)(1.0, 1.0)
    assert lib.edrgp_set_kernel(h._ctx, 1, kinds, one, ls, 0.1) < 0
    assert lib.edrgp_set_kernel(h._ctx, 0, kinds, one, ls, 0.1) < 0          # nparts out of range
    assert lib.edrgp_set_inducing(h._ctx, None, 8) < 0
    assert lib.edrgp_set_inducing(h._ctx, _lib.dptr(Z), 0) < 0
    assert len(lib.edrgp_last_error().decode()) > 0
    h.set_kernel(["RBF"], [1.0], np.full((1, 2), 1.0), 0.1)
    h.sparse_lml()
    with pytest.raises(_lib.EdrgpError):
        h.get("PHI", 3)                       # buffer smaller than M x M


def test_non_finite_rows_end_in_not_positive_definite(fresh):
    h = fresh
    X, y, Z = go.synthetic_problem(300, 3, 80, seed=2)
    Zbad = Z.copy()
    Zbad[5, 1] = np.nan
    h.set_data(X, y)
    h.set_inducing(Zbad)
    h.set_kernel(["RBF"], [1.0], np.full((1, 3), 1.5), 0.1)
    with pytest.raises(_lib.NotPositiveDefiniteError):
        h.sparse_lml()
    with pytest.raises(np.linalg.LinAlgError):        # the same error is a LinAlgError, as GPy's jitchol raises
        h.sparse_lml_grad()
    h.set_inducing(Z)                          # recovers with clean inputs
    ref = go.vardtc_inference(go.make_kernel(["RBF"], 3, [dict(variance=1.0, lengthscale=1.5)]), X, Z, 0.1, y[:, None])
    assert abs(h.sparse_lml() - ref["log_marginal"][0, 0]) <= 1e-9 * abs(ref["log_marginal"][0, 0])
