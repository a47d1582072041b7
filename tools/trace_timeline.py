"""Per-evaluation timeline out of a rocprofv3 kernel trace (csv): kernels of the LAST evaluation in start order with
start offset, duration and the idle gap in front of each on the merged timeline; a summary by kernel family."""
import csv, sys, re
from collections import defaultdict
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
def short(n):
    n = re.sub(r"^void ", "", n); n = n.replace("edrgp::", "")
    return n[:70]
# the last evaluation starts behind the last idle gap > 2 ms (tools/eval_trace.py sleeps in front of it)
starts = [i for i in range(1, len(ks)) if ks[i][0] - max(k[1] for k in ks[max(0, i - 50):i]) > 2000000]
i0 = starts[-1] if starts else 0
ev = ks[i0:]
t0 = ev[0][0]
busy_end = t0
fam = defaultdict(lambda: [0, 0.0])
print("evaluation of %d kernels, %.3f ms from first start to last end" % (len(ev), (max(k[1] for k in ev) - t0) / 1e6))
verbose = len(sys.argv) > 2
for s, e, n, q in ev:
    gap = max(0, s - busy_end)
    if verbose or (e - s) > 100000 or gap > 20000:
        print("%9.1f us  +%8.1f us  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, short(n)))
    busy_end = max(busy_end, e)
    f = fam[short(n)[:48]]
    f[0] += 1; f[1] += (e - s) / 1e3
print("--- by kernel (count, total us)")
for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%6d %10.1f  %s" % (c, t, n))
