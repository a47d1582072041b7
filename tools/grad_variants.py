import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from edrgp_amd.backend import HipGP
N, M, D = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (100000, 2048, 32)
X, y, Z = bench.synthetic(N, D, M, seed=1000)
h = HipGP(0)
h.set_option("overlap", 0)
h.set_data(X, y); h.set_inducing(Z)
h.set_kernel(["RBF"], [1.0], np.full((1, D), np.sqrt(D)), 0.01)
for name, opts in [("rbf_epilogue", dict(rbf_epilogue=1, fuse_grad=1)), ("dist_epilogue", dict(rbf_epilogue=0, fuse_grad=1)),
                   ("unfused", dict(rbf_epilogue=0, fuse_grad=0))]:
    for k, v in opts.items():
        h.set_option(k, v)
    h.set_option("profile", 1)
    h.sparse_lml_grad()
    h.profile_reset()
    for _ in range(3):
        h.sparse_lml_grad()
    ms, n = h.profile_get("GRADGEMM")
    print(name, "GRADGEMM slot %.3f ms per eval" % (ms / 3), flush=True)
    h.set_option("profile", 0)
