#!/bin/bash
# kernel durations of tools/mid_gemm_probe.py, per product type (mean of the last 2 of 3 launches)
cd /tmp && export TMPDIR=/tmp
for tile in "$@"; do
  rm -rf /tmp/mg_$tile
  rocprofv3 --kernel-trace --output-format csv -d /tmp/mg_$tile -- python3 $GRAFT_REPO_ROOT/tools/mid_gemm_probe.py 2048 $tile > /tmp/mg_$tile.names 2>/dev/null
  python3 - /tmp/mg_$tile /tmp/mg_$tile.names $tile <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
g = [(e - s) / 1e3 for s, e, n in rows if "gemm_f64_kernel" in n]
red = [(e - s) / 1e3 for s, e, n in rows if "slab_reduce" in n]
names = [l.strip() for l in open(sys.argv[2]) if l.strip()]
print("tile", sys.argv[3], "(%d gemm launches, %d slab reduces)" % (len(g), len(red)))
for i, nm in enumerate(names):
    d = g[3 * i: 3 * i + 3]
    if len(d) == 3: print("  %-40s %7.1f us" % (nm, (d[1] + d[2]) / 2))
PY
done
