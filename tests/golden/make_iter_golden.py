"""Generates tests/golden/iter_golden.npz: the REFERENCE's iterative EDR (edrgp.edr / edrgp.base, imported from
/root/reference in the build container) on the data of its own -- shipped commented-out -- test_iterative
(edrgp/tests/test_edr.py:120-133), with GP gradients instead of a duck estimator: the GP is this repository's
estimator class answered by the CPU oracle (tests/oracle_evaluator.py; GPy itself is absent).  Pins the adaptive
step rule (_select_n_components, edrgp/base.py:498-509) on real GP gradient matrices.  Inputs and expected
outputs only.  Run from the repo root:  python tests/golden/make_iter_golden.py   (about 10 s)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
from scipy.sparse import random as random_sparse  # noqa: E402
from edrgp_amd import models  # noqa: E402
from oracle_evaluator import OracleEvaluator  # noqa: E402
models.HipEvaluator = lambda device=None: OracleEvaluator()          # CPU answers for the generator only
from edrgp_amd.gp_model import GaussianProcessRegressor  # noqa: E402
from edrgp_amd.datasets import get_beta_inputs, get_edr_target  # noqa: E402
from edrgp.edr import EffectiveDimensionalityReduction  # noqa: E402   (reference)
from edrgp.utils import SVDTransformer, discrepancy  # noqa: E402      (reference)

np.random.seed(0)
X = get_beta_inputs(300, 10)
B = np.linalg.qr(random_sparse(10, 3, density=0.3, random_state=0).toarray())[0]
y = get_edr_target(X.dot(B), 0.1)
out = dict(X=X, y=y, B=B)
edr = EffectiveDimensionalityReduction(GaussianProcessRegressor(['RBF'], [{'ARD': True}]), SVDTransformer(),
                                       n_components=None, step=0.99, normalize=False)
edr.fit(X, y)
out["components"] = edr.components_
out["num_iter"] = np.array(edr.num_iter)
out["subspace_variance_ratio"] = np.atleast_1d(edr.subspace_variance_ratio_)
out["discrepancy"] = np.array(discrepancy(B, edr.components_.T))
out["first_gradients"] = edr._first_gradients_
print("reference: components", edr.components_.shape, "iterations", edr.num_iter, "discrepancy %.4f" % out["discrepancy"])
path = os.path.join(HERE, "iter_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
