#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ttx
rocprofv3 --kernel-trace --output-format csv -d /tmp/ttx -- python3 $GRAFT_REPO_ROOT/tools/ttx_probe.py
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ttx/*/*kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
for s, e, n in rows:
    if "gemm_f64" in n: print(n[:64], "%.1f us" % ((e - s) / 1e3))
PY
