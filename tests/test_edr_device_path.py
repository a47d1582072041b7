"""The EDR loop on the device (edrgp/base.py:459-517, edrgp/utils.py:140-155): with the MI355X
regressors and SVDTransformer the estimators consume the posterior-mean gradients at the training rows as their
D x D Gram matrix, reduced on the device (edrgp_predict_train_gram), and refit the GP on rows the device projects
itself (edrgp_set_data_projected).  Same result as the reference's dense path up to the sign of a singular vector;
no N x D matrix crosses PCIe inside the loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(N, D, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D)) * (1.0 + 0.3 * np.arange(D) / D) + 2.0
    B = np.zeros((2, D)); B[0, 0] = 1.0; B[1, 1:3] = [0.6, 0.8]
    y = np.tanh(X @ B[0] - 2.0) + 0.5 * (X @ B[1] - 2.8) ** 2 + 0.05 * rng.standard_normal(N)
    return X, y, B


@pytest.mark.parametrize("step,n_components", [(None, 2), (3, 2)])
def test_edr_loop_device_reduction_equals_dense_path(monkeypatch, step, n_components):
    from edrgp_amd import EffectiveDimensionalityReduction
    from edrgp_amd.backend import HipGP
    from edrgp_amd.device import shared_context
    from edrgp_amd.gp_model import SparseGaussianProcessRegressor
    from edrgp_amd.utils import GradientGram, SVDTransformer, discrepancy
    X, y, B = _data(12000, 8, seed=3)
    calls = {"train_rows": [], "gram": 0}
    orig_train, orig_gram = HipGP.predict_train, HipGP.predict_train_gram

    def counted_train(self, want_mean=True, want_grad=True):
        calls["train_rows"].append((self.N, self.D) if want_grad else (self.N, 0))
        return orig_train(self, want_mean, want_grad)

    def counted_gram(self, want_mean=False):
        calls["gram"] += 1
        return orig_gram(self, want_mean)
    monkeypatch.setattr(HipGP, "predict_train", counted_train)
    monkeypatch.setattr(HipGP, "predict_train_gram", counted_gram)

    def run(device_reduction):
        np.random.seed(0)
        edr = EffectiveDimensionalityReduction(SparseGaussianProcessRegressor(num_inducing=96), SVDTransformer(),
                                               n_components=n_components, step=step, normalize=True)
        edr.device_reduction = device_reduction
        edr.fit(X, y, max_iters=40)
        return edr

    sends0 = shared_context().projected_sends
    dev = run(True)
    assert isinstance(dev._first_gradients_, GradientGram)
    assert calls["gram"] == dev.num_iter                             # one device-side reduction per decomposition
    assert shared_context().projected_sends - sends0 == dev.num_iter  # every refit projected the resident rows
    # the only gradient matrix that crossed PCIe is the final N x n_components one (subspace_gradients_)
    assert [d for (_, d) in calls["train_rows"] if d > 0] == [n_components]
    assert dev.subspace_gradients_.shape == (X.shape[0], n_components)

    calls["train_rows"].clear(); calls["gram"] = 0
    host = run(False)
    assert calls["gram"] == 0 and any(d == X.shape[1] for (_, d) in calls["train_rows"])
    assert host.num_iter == dev.num_iter and host.components_.shape == dev.components_.shape

    def unit_rows(C):
        C = C / np.linalg.norm(C, axis=1, keepdims=True)
        return C * np.sign(C[np.arange(C.shape[0]), np.argmax(np.abs(C), axis=1)])[:, None]
    # same subspace, same directions up to sign (the refits see mirrored coordinates: optimiser paths differ slightly)
    assert np.abs(unit_rows(dev.components_) - unit_rows(host.components_)).max() < 2e-3
    assert np.allclose(dev.subspace_variance_ratio_, host.subspace_variance_ratio_, rtol=2e-3, atol=1e-6)
    Bn = B.T / np.linalg.norm(B.T, axis=0)
    Q = np.linalg.qr(dev.components_.T)[0]
    assert discrepancy(Bn, Q) < 0.1
    # the dense first-iteration gradients are still available on demand
    G = np.asarray(dev._first_gradients_)
    assert G.shape == X.shape
    assert np.abs(G.T @ G - dev._first_gradients_.gram).max() <= 1e-9 * np.abs(dev._first_gradients_.gram).max()
