import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.gp_model import SparseGaussianProcessRegressor
N, M, D = 100000, 2048, 32
X, y, _ = bench.synthetic(N, D, M, seed=5)
np.random.seed(0)
est = SparseGaussianProcessRegressor(num_inducing=M)
pr = cProfile.Profile()
pr.enable()
est.fit(X, y, max_iters=8)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
