"""A/B timing of Phi = K^T K kernel variants in one process (interleaved rounds)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP  # noqa: E402

N, M = 100000, 2048
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
splits_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
h = HipGP(0)
ms0, ref = h.dbg_syrk_variant(2048, 256, 0, 0, 1, True)
for v in variants:
    if v:
        _, phi = h.dbg_syrk_variant(2048, 256, v, 0, 1, True)
        print("variant", v, "max rel err vs production", np.abs(phi - ref).max() / np.abs(ref).max(), flush=True)
flops = N * M * (M + 1.0)
res = {}
for rnd in range(3):
    for v in variants:
        for sp in splits_list:
            ms, _ = h.dbg_syrk_variant(N, M, v, sp, 3)
            res.setdefault((v, sp), []).append(ms)
for k, t in res.items():
    print("variant %d splits %3d : ms %s  -> %.1f TFLOP/s algorithmic (min)" % (k[0], k[1], ["%.3f" % x for x in t], flops / min(t) * 1e-9))
