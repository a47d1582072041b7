"""Shared by tests/test_sharded_native.py and tests/test_multi_gpu.py: launches tests/sharded_worker.py under
torch.distributed.run and compares every rank's results with the whole-problem evaluation and with the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np

from oracle import gp_oracle as go

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_and_check(tmp_path, world, transport):
    import sharded_worker as sw
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "sharded_worker.py"), "--transport", transport, "--out", str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "worker failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:])
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))   # noqa: E731
    for ci, (name, N, M, D, kind, lsf, noise, chunk, smode, sfree) in enumerate(sw.CASES):
        ranks = [np.load(os.path.join(str(tmp_path), "%s_rank%d.npz" % (name, k))) for k in range(world)]
        whole = np.load(os.path.join(str(tmp_path), "%s_whole.npz" % name))
        # the shards really straddle chunk_rows: at least one rank chunked, and (two ranks) one on a single chunk
        sizes = [int(r["rows"][1] - r["rows"][0]) for r in ranks]
        assert sum(sizes) == N
        if world == 2:
            assert min(sizes) <= chunk < max(sizes)
        # same form on every rank and on the whole problem (the rule looks at the replicated Kmm)
        assert len({float(r["mode"]) for r in ranks} | {float(whole["mode"])}) == 1, name
        assert float(whole["mode"]) == (2.0 if ("illcond" in name or name == "stable_forced") else 1.0), name
        # replicated results: identical bits on every rank
        for k in range(1, world):
            for key in ("lml", "bound_only", "bound_post", "dvar", "dlen", "dnoise", "dZ", "mean", "var", "wv"):
                assert np.array_equal(ranks[0][key], ranks[k][key]), (name, key, k)
        if "collectives" in ranks[0].files:          # every rank issued the same sequence of all-reduces
            for k in range(1, world):
                assert np.array_equal(ranks[0]["collectives"], ranks[k]["collectives"]), \
                    (name, k, ranks[0]["collectives"], ranks[k]["collectives"])
            Mw = (M + 127) // 128 * 128 if (M > 256 and M % 128) else M
            assert Mw * (Mw + 1) // 2 + Mw in set(int(v) for v in ranks[0]["collectives"])   # the packed triangle travelled
        # sharded == whole (different summation order of Phi only)
        cond_slack = 100.0 if "illcond" in name else 1.0
        r0 = ranks[0]
        assert abs(r0["lml"] - whole["lml"]) <= 1e-10 * cond_slack * abs(whole["lml"]), name
        assert abs(r0["bound_only"] - whole["lml"]) <= 1e-10 * cond_slack * abs(whole["lml"])
        assert abs(r0["bound_post"] - whole["lml"]) <= 1e-10 * cond_slack * abs(whole["lml"])
        gs = max(abs(float(whole["dnoise"])), np.abs(whole["dvar"]).max(), np.abs(whole["dlen"]).max())
        assert max(np.abs(r0["dvar"] - whole["dvar"]).max(), np.abs(r0["dlen"] - whole["dlen"]).max(),
                   abs(float(r0["dnoise"] - whole["dnoise"]))) <= 1e-7 * cond_slack * gs, name
        assert rel(r0["dZ"], whole["dZ"]) <= 1e-6 * cond_slack
        assert rel(r0["mean"], whole["mean"]) <= 1e-8 * cond_slack
        # and both equal the oracle on all rows
        X, y, Z = sw.problem(N, D, M, seed=100 + ci)
        kern = go.make_kernel([kind], D, [dict(variance=1.0, lengthscale=lsf * np.sqrt(D))])
        inf = go.vardtc_inference(kern, X, Z, noise, y[:, None])
        ref = float(inf["log_marginal"][0, 0])
        assert abs(r0["lml"] - ref) <= 1e-9 * cond_slack * max(abs(ref), float(N)), (name, float(r0["lml"]), ref)
        mu, _ = go.raw_predict(kern, X[:64], Z, inf["woodbury_vector"], inf["woodbury_inv"])
        assert rel(r0["mean"], mu[:, 0]) <= 1e-7 * cond_slack
    sweep = np.load(os.path.join(str(tmp_path), "sweep.npz"))
    assert bool(sweep["ok"]) and int(sweep["n"]) == int(sweep["n_ref"]) == 12
