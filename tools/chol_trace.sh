#!/bin/bash
# Per-launch durations of the factorisation + inverse at n = 2048 alone (kernel trace of tools/chol_time.py):
#   tools/chol_trace.sh <outdir under gpurun_out> [option=value ...]
set -e
OUT=gpurun_out/${1:-chol}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/chol_time.py 2048 "${@:2}" > $OUT/chol_time.log 2>&1
python3 - "$(ls $OUT/trace/*/*kernel_trace.csv | head -1)" > $OUT/chol_steps.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# last factorisation: from the last diag_block_kernel to the take_lower that follows
i0 = max(i for i, k in enumerate(ks) if "diag_block_kernel" in k[2])
i1 = next(i for i in range(i0, len(ks)) if "take_lower" in ks[i][2])
seg = ks[i0:i1 + 1]
print("factorisation + inverse: %d launches, %.1f us from first start to last end" % (len(seg), (seg[-1][1] - seg[0][0]) / 1e3))
steps = [(e - s) / 1e3 for s, e, n in seg if "chol_step" in n]
print("steps: n=%d mean %.2f min %.2f max %.2f us; first 4: %s; last 4: %s" % (len(steps), sum(steps) / len(steps), min(steps), max(steps),
      " ".join("%.1f" % v for v in steps[:4]), " ".join("%.1f" % v for v in steps[-4:])))
print("diag_block %.1f us" % ((seg[0][1] - seg[0][0]) / 1e3))
PY
rm -rf $OUT/trace
cat $OUT/chol_time.log | grep -v rocprofv3; cat $OUT/chol_steps.txt
