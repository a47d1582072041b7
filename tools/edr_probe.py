"""EDR end to end on the device estimators: N=50k, D=16, two relevant directions."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd import EffectiveDimensionalityReduction
from edrgp_amd.gp_model import SparseGaussianProcessRegressor
from edrgp_amd.utils import SVDTransformer, discrepancy
rng = np.random.default_rng(0)
N, D = [int(v) for v in sys.argv[1:3]] if len(sys.argv) > 2 else (50000, 16)
MIND = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
X = rng.standard_normal((N, D))
B = np.zeros((2, D)); B[0, 0] = 1.0; B[1, 1:3] = [0.6, 0.8]
y = np.tanh(X @ B[0]) + 0.5 * (X @ B[1]) ** 2 + 0.05 * rng.standard_normal(N)
np.random.seed(0)
edr = EffectiveDimensionalityReduction(SparseGaussianProcessRegressor(num_inducing=MIND, kernel_options=None),
                                       SVDTransformer(), n_components=2, normalize=True)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
edr.fit(X, y, max_iters=30)
dt = time.perf_counter() - t0
pr.disable()
print("EDR fit: %.2f s; subspace discrepancy vs the true directions: %.4f; variance ratios %s"
      % (dt, discrepancy(B, edr.components_), np.round(edr.subspace_variance_ratio_, 4)))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
