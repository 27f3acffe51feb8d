"""Per-dispatch averages of the PMC passes of tools/profile_cfg4.sh for the SpMV kernels of config 4."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

out, tag = Path(sys.argv[1]), sys.argv[2]
dst = out / "summary"
dst.mkdir(exist_ok=True)
res = defaultdict(dict)
for f in sorted(out.rglob("*counter_collection.csv")):
    acc, cnt = defaultdict(float), defaultdict(int)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            if "spmv" not in k:
                continue
            key = (k, row.get("Counter_Name"))
            acc[key] += float(row["Counter_Value"])
            cnt[key] += 1
    for (k, c), v in acc.items():
        res[k][c] = {"dispatches": cnt[(k, c)], "avg_per_dispatch": v / cnt[(k, c)]}
stats = {}
for f in sorted((out / "trace").rglob("*kernel_stats.csv")):
    (dst / f"{tag}_cfg4_kernel_stats.csv").write_text(f.read_text())
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "spmv" in row.get("Name", ""):
                stats[row["Name"].split("(")[0]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
summary = {"_comment": "config 4 (5M x 1M, 20 nnz/row), column tiles of 393216 columns: per-dispatch counter averages of the tile kernels; "
                       "FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE reads half of a wide coalesced stream on gfx950, MI355X_MICROARCH.md)",
           "counters": res, "kernel_stats": stats}
(dst / f"{tag}_cfg4_pmc.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1)[:3000])
