#!/bin/bash
# same-box A/B of library builds: Cm / C2 evaluation times
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    EDRGP_LIB=$GRAFT_REPO_ROOT/edr-gp_amd/$lib python3 tools/opt_times.py "" 2>&1 | sed 's/^/  /'
  done
done
