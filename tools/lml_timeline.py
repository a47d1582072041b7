"""Reads a rocprofv3 kernel-trace CSV and prints the device timeline of the last bound-only evaluation
(kernel, stream/queue, start offset, duration, gap to the previous end)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
# evaluations start with the kernel-matrix build of Kmm; find the last occurrence of the Phi syrk and walk back
idx = [i for i, r in enumerate(rows) if "Li1EL" in r[2] or "tag1" in r[2] or "gemm_kernel" in r[2] and False]
names = [r[2] for r in rows]
syrk = [i for i, n in enumerate(names) if n.startswith("void edrgp::gemm_mfma_kernel") and "1," in n and False]
# simpler: take the last `count` kernels
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sel = rows[-count:]
t0 = sel[0][0]
prev_end = t0
for s, e, n, q in sel:
    short = n.replace("edrgp::", "").split("(")[0][:70]
    print("%9.1f us  dur %8.1f  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, short))
    prev_end = max(prev_end, e)
