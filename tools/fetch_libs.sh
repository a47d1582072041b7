#!/bin/bash
# FETCH_SIZE of the Phi kernel alone for several builds of the library: tools/fetch_libs.sh <lib.so> [<lib.so> ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetchl; mkdir -p $O; cd $R; export TMPDIR=/tmp
i=0
for lib in "$@"; do
  i=$((i+1)); export EDRGP_LIB=$R/$lib
  python3 tools/syrk_fetch.py 2>&1 | grep syrk
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p$i -- python3 tools/syrk_fetch.py > $O/p$i.log 2>&1 || exit 3
  cp $(ls $O/p$i/*/*counter_collection.csv | head -1) $O/p$i.csv; rm -rf $O/p$i
  python3 - <<PY
import csv
x=[float(r["Counter_Value"]) for r in csv.DictReader(open("$O/p$i.csv")) if r["Counter_Name"]=="FETCH_SIZE" and "gemm_f64_kernel<128, 128, true, true, 1>" in r["Kernel_Name"]]
print("   [$lib] FETCH e6 KB:", [round(a/1e6,2) for a in x])
PY
done
