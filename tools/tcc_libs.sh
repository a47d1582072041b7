#!/bin/bash
# L2 counters of the Phi kernel alone for several builds: tools/tcc_libs.sh <lib.so> ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tcc; mkdir -p $O; cd $R; export TMPDIR=/tmp
i=0
for lib in "$@"; do
  export EDRGP_LIB=$R/$lib
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 tools/syrk_fetch.py > $O/p$i.log 2>&1 || { tail -3 $O/p$i.log; continue; }
    cp $(ls $O/p$i/*/*counter_collection.csv | head -1) $O/p$i.csv; rm -rf $O/p$i
    python3 - <<PY
import csv
from collections import defaultdict
v=defaultdict(list)
for r in csv.DictReader(open("$O/p$i.csv")):
    if "true, true, 1>" in r["Kernel_Name"]: v[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   [$lib]", {k: round(sum(x)/len(x)/1e6,2) for k,x in v.items()}, "(millions)")
PY
  done
done
