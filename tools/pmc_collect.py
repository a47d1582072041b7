"""Averages per kernel of a rocprofv3 --pmc counter_collection csv:  python tools/pmc_collect.py <csv> [substring ...]"""
import csv, sys
from collections import defaultdict
path = sys.argv[1]
want = sys.argv[2:] or ["gemm_f64_kernel<128, 128"]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if any(w in n for w in want):
        acc[n.replace("void edrgp::", "").replace("(edrgp::GemmArgs)", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        print("   %-28s avg %.6g over %d dispatches" % (c, sum(v) / len(v), len(v)))
