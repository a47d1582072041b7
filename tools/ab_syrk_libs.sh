#!/bin/bash
for rep in 1 2; do
  for lib in "$@"; do
    EDRGP_LIB=$GRAFT_REPO_ROOT/edr-gp_amd/$lib python3 - $lib <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from edrgp_amd.backend import HipGP
h = HipGP(0)
out = []
for N, M in ((100000, 2048), (100000, 1024)):
    ms = h.dbg_syrk_time(N, M, 0, 20)[0]
    out.append("N=%d M=%d: %.3f ms" % (N, M, ms))
print("%-28s %s" % (sys.argv[1], "   ".join(out)))
h.close()
PY
  done
done
