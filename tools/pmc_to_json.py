"""rocprofv3 --pmc counter_collection csv files -> one JSON per kernel family, averaged over the FULL-SIZE dispatches of
that kernel (a kernel launched at several sizes -- the gradient contraction's whole rounds and its last partial round --
is reported for its largest launches only: dispatches whose counter value is at least half the maximum).
    python tools/pmc_to_json.py <out.json> <kernel-name substring> <csv> [<csv> ...]"""
import csv, json, sys
from collections import defaultdict
out, want, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
vals = defaultdict(list)
name = None
for p in paths:
    for r in csv.DictReader(open(p)):
        if want in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"kernel": (name or want).replace("void edrgp::", "").replace("(edrgp::GemmArgs)", "")}
for c, v in sorted(vals.items()):
    big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
    res[c] = sum(big) / len(big)
    res[c + "_dispatches"] = len(big)
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    # gfx950: FETCH_SIZE (KB) reports half of a wide (16 B / lane) coalesced read stream -> doubled; WRITE_SIZE exact
    # (MI355X_MICROARCH.md, HBM section)
    res["hbm_bytes_per_launch"] = (2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0
if "SQ_LDS_BANK_CONFLICT" in res and res.get("SQ_LDS_IDX_ACTIVE"):
    res["lds_conflict_share"] = res["SQ_LDS_BANK_CONFLICT"] / res["SQ_LDS_IDX_ACTIVE"]
if res.get("TCC_HIT_sum") is not None and res.get("TCC_MISS_sum") is not None:
    res["l2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
if res.get("TCC_EA0_RDREQ_sum") is not None:
    # every fabric read request of these kernels is a 128-byte line (TCC_EA0_RDREQ_32B = 0): the doubled FETCH_SIZE
    res["fabric_read_bytes_from_rdreq"] = res["TCC_EA0_RDREQ_sum"] * 128.0
# what the numbers are (tools/r03_profiles.sh collects them; N = 100000, M = 2048, D = 32)
LABELS = {
    "true, true, 1>": ("Phi = K_mn K_nm (N=100000, M=2048), persistent grid: K_nm once + 12 slabs",
                       100000 * 2048 * 8 + 12 * (2048 * 2049 // 2) * 8),
    "false, true, 2>": ("gradient contraction K_nm S2 with the fused RBF epilogue (whole rounds launch: 98304 rows, M=2048): "
                        "K_nm read twice (operand + epilogue), T written, S2 panels",
                        3 * 98304 * 2048 * 8 + 2048 * 2048 * 8),
}
for key, (what, alg) in LABELS.items():
    if key in want:
        res["what"] = what
        res["algorithmic_bytes_per_launch"] = alg
res["correction"] = ("gfx950: FETCH_SIZE (KB) reports 1/2 of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE exact "
                     "(MI355X_MICROARCH.md, HBM section); separate --pmc passes.  FETCH counts the L2's fabric read requests "
                     "(TCC_EA0_RDREQ x 64 B; all of them 128-byte lines here), i.e. L2 misses whether the Infinity Cache or HBM "
                     "serves them: an upper bound of the HBM bytes")
res["commands"] = ["tools/r03_profiles.sh: six passes of  rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 "
                   "tools/eval_trace.py Cm grad 2  (SQ set 1, SQ set 2, FETCH_SIZE, WRITE_SIZE, TCC hit/miss, TCC_EA0_RDREQ), "
                   "averaged over the full-size dispatches by tools/pmc_to_json.py"]
res["round"] = 3
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
