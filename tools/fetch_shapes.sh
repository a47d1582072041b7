#!/bin/bash
# FETCH_SIZE of the Phi kernel alone at several shapes: tools/fetch_shapes.sh "N M" ...   (EDRGP_LIB respected)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetchs; mkdir -p $O; cd $R; export TMPDIR=/tmp
i=0
for nm in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p$i -- python3 tools/syrk_fetch.py $nm > $O/p$i.log 2>&1 || exit 3
  cp $(ls $O/p$i/*/*counter_collection.csv | head -1) $O/p$i.csv; rm -rf $O/p$i
  python3 - <<PY
import csv
x=[float(r["Counter_Value"]) for r in csv.DictReader(open("$O/p$i.csv")) if r["Counter_Name"]=="FETCH_SIZE" and "true, true, 1>" in r["Kernel_Name"]]
n,m="$nm".split()[:2]; k=int(n)*int(m)*8/1024
print("   [$nm] FETCH e6 KB:", [round(a/1e6,2) for a in x], " K_nm = %.2f e6 KB, ratio %.2f" % (k/1e6, sum(x)/len(x)/k))
PY
done
