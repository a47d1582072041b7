"""RCCL path on one GPU: a single-rank communicator exercises the library's
ncclCommInitRank / ncclAllReduce calls (the all-reduce of [Phi | K_mn y], [N, y'y] and the
gradient partials); results must equal the no-communicator evaluation bit for bit."""
import numpy as np
import pytest

from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


def test_single_rank_allreduce_is_identity():
    from edrgp_amd.backend import HipGP
    N, M, D = 1200, 96, 6
    X, y, Z = go.synthetic_problem(N, D, M, seed=21)
    h = HipGP(0)
    try:
        h.set_data(X, y)
        h.set_inducing(Z)
        h.set_kernel(["Matern52"], [1.2], np.full((1, D), 2.0), 0.05)
        ref = h.sparse_lml_grad()
        h.comm_init(1, 0, HipGP.comm_unique_id())
        got = h.sparse_lml_grad()
        h.comm_destroy()
        assert got[0] == ref[0]
        for a, b in zip(got[1:], ref[1:]):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        with pytest.raises(Exception):
            h.comm_init(1, 0, HipGP.comm_unique_id())
            h.dense_lml()                      # the dense exact GP refuses a communicator: replicas only
        h.comm_destroy()
    finally:
        h.close()
