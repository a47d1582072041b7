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
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
