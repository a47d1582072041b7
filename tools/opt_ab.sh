#!/bin/bash
# Same-box A/B of a probe-build option: tools/opt_ab.sh "<opt>=<a>" "<opt>=<b>" [configs...]
R=$GRAFT_REPO_ROOT; cd $R
export EDRGP_LIB=$R/edr-gp_amd/libedrgp_hip_probe.so
A="$1"; B="$2"; shift 2
CFGS="${@:-Cm C2 C3s}"
for r in 1 2 3; do for o in "$A" "$B"; do for cfg in $CFGS; do
  echo -n "[$o] "; timeout -k 5 120 python3 tools/eval_trace.py $cfg grad 20 $o 2>&1 | grep "per evaluation" | cut -c1-40
done; done; done
