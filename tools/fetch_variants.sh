#!/bin/bash
# FETCH_SIZE of the Phi and contraction kernels (and the untraced evaluation time) under probe-build options:
#   tools/fetch_variants.sh "gemm_dephase=0" "gemm_dephase=2" ...     ("-" = defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetchv; mkdir -p $O; cd $R; export TMPDIR=/tmp
export EDRGP_LIB=$R/edr-gp_amd/libedrgp_hip_probe.so
i=0
for opt in "$@"; do
  i=$((i+1)); args=""; [ "$opt" != "-" ] && args="$opt"
  python3 tools/eval_trace.py Cm grad 5 $args 2>&1 | grep "per evaluation"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p$i -- python3 tools/eval_trace.py Cm grad 2 $args > $O/p$i.log 2>&1 || exit 3
  cp $(ls $O/p$i/*/*counter_collection.csv | head -1) $O/p$i.csv; rm -rf $O/p$i
  python3 - <<PY
import csv
from collections import defaultdict
v=defaultdict(list)
for r in csv.DictReader(open("$O/p$i.csv")):
    if r["Counter_Name"]=="FETCH_SIZE" and "gemm_f64_kernel<128, 128" in r["Kernel_Name"]:
        v[r["Kernel_Name"][20:62]].append(float(r["Counter_Value"]))
for k,x in v.items():
    big=[a for a in x if a>0.5*max(x)]
    print("   [$opt]", k, "FETCH %.2f e6 KB (n=%d)" % (sum(big)/len(big)/1e6, len(big)))
PY
done
