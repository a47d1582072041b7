#!/bin/bash
# Timing probe: duration of the gradient contraction with its operand tiles aliased into the L2 (gemm_alias, wrong results)
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/alias_$a -- python3 $GRAFT_REPO_ROOT/tools/eval_trace.py Cm grad 3 gemm_alias=$a > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/gpurun_out/alias_$a/*/*kernel_trace.csv | head -1)
  python3 - "$f" $a <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "false, true, 2>" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("alias", sys.argv[2], "contraction us:", " ".join("%.1f" % x for x in d))
PY
done
