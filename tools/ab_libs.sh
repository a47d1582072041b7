#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for lib in ${LIBS:-edr-gp_amd/libedrgp_hip.so edr-gp_amd/libedrgp_hip_flag.so}; do
  export EDRGP_LIB=$PWD/$lib
  echo -n "$lib  "; python3 tools/eval_trace.py Cm grad 20 2>&1 | grep "per evaluation" | cut -c1-40 | tr '\n' ' '; python3 tools/eval_trace.py Cm lml 20 2>&1 | grep "per evaluation" | cut -c1-40
done; done
