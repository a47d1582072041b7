#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch; mkdir -p $O; cd $R; export TMPDIR=/tmp
for i in 1 2; do
  if [ $i = 2 ]; then export EDRGP_LIB=$R/_old/edr-gp_amd/libedrgp_hip.so; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p$i -- python3 tools/eval_trace.py Cm grad 2 > $O/p$i.log 2>&1 || exit 3
  cp $(ls $O/p$i/*/*counter_collection.csv | head -1) $O/p$i.csv; rm -rf $O/p$i
  python3 - <<PY
import csv
from collections import defaultdict
v=defaultdict(list)
for r in csv.DictReader(open("$O/p$i.csv")):
    if r["Counter_Name"]=="FETCH_SIZE" and "gemm_f64_kernel<128, 128" in r["Kernel_Name"]:
        v[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,x in v.items(): print("run $i", k, [round(a/1e6,2) for a in x])
PY
done
