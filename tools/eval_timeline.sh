#!/bin/bash
# Kernel-trace timelines of single evaluations at a BASELINE config (one GPU call):
#   tools/eval_timeline.sh <outdir under gpurun_out> [config=Cm]
# -> <outdir>/timeline_{lml,grad}.txt (tools/trace_timeline.py, verbose), chol_time.log
set -e
OUT=gpurun_out/${1:-timeline}
CFG=${2:-Cm}
mkdir -p $OUT
export TMPDIR=/tmp
for what in lml grad; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$what -- python3 tools/eval_trace.py $CFG $what 3 > $OUT/trace_$what.log 2>&1
  python3 tools/trace_timeline.py $(ls $OUT/trace_$what/*/*kernel_trace.csv | head -1) v > $OUT/timeline_$what.txt
  cp $(ls $OUT/trace_$what/*/*kernel_trace.csv | head -1) $OUT/kernel_trace_$what.csv
  rm -rf $OUT/trace_$what
done
tail -n 3 $OUT/*.log
