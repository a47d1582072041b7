#!/bin/bash
# same-box A/B of two builds of the library: isolated Phi launch and Cm / C2 evaluation times
for rep in 1 2; do
  for lib in libedrgp_hip_old.so libedrgp_hip.so; do
    echo "== $lib"
    EDRGP_LIB=$GRAFT_REPO_ROOT/edr-gp_amd/$lib python3 - <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from edrgp_amd.backend import HipGP
h = HipGP(0)
for N, M in ((100000, 2048), (100000, 1024)):
    ms = h.dbg_syrk_time(N, M, 0, 20)[0]
    print("  isolated Phi N=%d M=%d: %.3f ms (%.1f TFLOP/s)" % (N, M, ms, N * M * (M + 1.0) / ms / 1e9))
h.close()
PY
    EDRGP_LIB=$GRAFT_REPO_ROOT/edr-gp_amd/$lib python3 tools/opt_times.py "" 2>&1 | sed 's/^/  /'
  done
done
