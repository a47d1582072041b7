"""bench.py's contract with the driver: one JSON line with the agreed keys (metric, value, unit, n_gpus, steps, warmup,
ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline, cpu_baseline), and the
synthetic inputs of SURVEY 8(d) -- every rank of a sharded run holds rank 0's inducing inputs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_synthetic_inputs_are_deterministic_and_share_inducing_inputs():
    import bench
    X0, y0, Z0 = bench.synthetic(500, 6, 40, seed=bench.RANK0_SEED)
    X1, y1, Z1 = bench.synthetic(500, 6, 40, seed=bench.RANK0_SEED + 1)
    Xa, ya, Za = bench.synthetic(500, 6, 40, seed=bench.RANK0_SEED)
    assert np.array_equal(X0, Xa) and np.array_equal(y0, ya) and np.array_equal(Z0, Za)
    assert not np.array_equal(X0, X1)
    assert np.array_equal(Z0, Z1)                       # rank 1 regenerates rank 0's rows for Z
    assert abs(y0.mean()) < 1e-12 and abs(y0.std() - 1.0) < 1e-12
    rows = {tuple(np.round(r, 12)) for r in X0}
    assert all(tuple(np.round(z, 12)) in rows for z in Z0)          # Z = X[perm[:M]]
    th = [bench.theta_path(i, 6, "RBF") for i in range(8)]
    assert th[0] == th[7] and th[0] != th[1]


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--N", "12000",
           "--M", "512", "--D", "16", "--iters", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "points/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 12000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert c["device_bound_rel_err"] < 1e-9             # the device's bound against the oracle's on the same inputs
    assert d["cholesky"]["isolated_ms"] > 0
