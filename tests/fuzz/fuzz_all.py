import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz
from edrgp_amd.backend import HipGP
h = HipGP(0)
tot = []
for seed in (range(int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1, 2, 3, 4, 5)):
    bad = fuzz.run(h, seed=seed, n_cases=70, modes=(0, 0, 2), verbose=True)
    print("seed", seed, "failing", bad, flush=True)
    tot += bad
print("TOTAL failing:", len(tot))
import fuzz_dense, fuzz_fp32, fuzz_ep
print("dense fuzz failing:", fuzz_dense.run(h, seed=11, n_cases=40, verbose=True))
print("fp32 fuzz failing:", fuzz_fp32.run(h, seed=12, n_cases=40, verbose=True))
print("EP / EPDTC fuzz failing:", fuzz_ep.run(h, seed=13, n_cases=60, verbose=True))
