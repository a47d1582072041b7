#!/bin/bash
for lib in "$@"; do
  echo "== $lib"; EDRGP_LIB=$GRAFT_REPO_ROOT/edr-gp_amd/$lib bash tools/mid_gemm_run.sh 0 2>&1 | sed 's/^/  /'
done
bash tools/ab_eval_libs.sh "$@"
