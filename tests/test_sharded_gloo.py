"""N > 1 path on CPU: two processes (gloo, 127.0.0.1) each hold a row block of
(X, y) and the same Z / hyper-parameters, evaluate the Gram-form pipeline with
torch.distributed all-reduces at exactly the points where the HIP library calls
RCCL (api.hip: [N, y'y], [Phi | K_mn y], gradient partials), and must agree
with the single-process oracle on all rows."""
import os
import socket

import numpy as np
import pytest

from oracle import gp_oracle as go


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, N, M, D, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    from test_gram_formulation import gram_sparse_eval
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, Z = go.synthetic_problem(N, D, M, seed=5)
    lo, hi = (N * rank) // world, (N * (rank + 1)) // world          # ragged shards allowed
    kern = go.make_kernel(["Matern52"], D, [dict(variance=1.1, lengthscale=2.0)])

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    g = gram_sparse_eval(kern, X[lo:hi], y[lo:hi], Z, 0.05, allreduce)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lml=g["lml"], wv=g["wv"], dvar=np.array(g["dvar"]),
             dlen=np.concatenate(g["dlen"]), dnoise=g["dnoise"], dZ=g["dZ"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N", [400, 401])
def test_row_sharded_evaluation_two_ranks(tmp_path, N):
    import torch.multiprocessing as mp
    M, D, world = 24, 3, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, N, M, D, str(tmp_path)), nprocs=world, join=True)
    X, y, Z = go.synthetic_problem(N, D, M, seed=5)
    kern = go.make_kernel(["Matern52"], D, [dict(variance=1.1, lengthscale=2.0)])
    inf = go.vardtc_inference(kern, X, Z, 0.05, y[:, None])
    kg, dn, dZ = go.sparse_gradients(kern, X, Z, inf)
    ref = inf["log_marginal"][0, 0]
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    for k in range(world):
        assert abs(r[k]["lml"] - ref) < 1e-10 * abs(ref)
        assert np.allclose(r[k]["wv"], inf["woodbury_vector"][:, 0], rtol=1e-7, atol=1e-9)
        assert np.isclose(r[k]["dvar"][0], kg[0][0], rtol=1e-8)
        assert np.allclose(r[k]["dlen"], kg[0][1], rtol=1e-8)
        assert np.isclose(r[k]["dnoise"], dn, rtol=1e-8)
        assert np.allclose(r[k]["dZ"], dZ, rtol=1e-7, atol=1e-9)
    # every rank holds the same replicated result
    assert np.array_equal(r[0]["lml"], r[1]["lml"]) and np.array_equal(r[0]["dZ"], r[1]["dZ"])
