"""The Phi kernel alone (device-generated K) for a PMC pass: python3 tools/syrk_fetch.py [N M reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edrgp_amd.backend import HipGP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
h = HipGP(0)
ms, _ = h.dbg_syrk_time(N, M, 0, reps)
print("syrk %d x %d: %.3f ms" % (N, M, ms))
h.close()
