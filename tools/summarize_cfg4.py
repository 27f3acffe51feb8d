"""Per-dispatch averages of the PMC passes of tools/profile_cfg4.sh for the SpMV kernels of config 4."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

out, tag = Path(sys.argv[1]), sys.argv[2]
# optional: label of the summary file and comma-separated kernel-name filters (default: the SpMV kernels of config 4)
label = sys.argv[3] if len(sys.argv) > 3 else "cfg4"
filters = sys.argv[4].split(",") if len(sys.argv) > 4 else ["spmv"]
comment = sys.argv[5] if len(sys.argv) > 5 else ("config 4 (5M x 1M, 20 nnz/row), column tiles of 393216 columns: per-dispatch counter "
                                                "averages of the tile kernels")
dst = out / "summary"
dst.mkdir(exist_ok=True)
res = defaultdict(dict)
for f in sorted(out.rglob("*counter_collection.csv")):
    acc, cnt = defaultdict(float), defaultdict(int)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            if not any(f_ in k for f_ in filters):
                continue
            key = (k, row.get("Counter_Name"))
            acc[key] += float(row["Counter_Value"])
            cnt[key] += 1
    for (k, c), v in acc.items():
        res[k][c] = {"dispatches": cnt[(k, c)], "avg_per_dispatch": v / cnt[(k, c)]}
stats = {}
for f in sorted((out / "trace").rglob("*kernel_stats.csv")):
    (dst / f"{tag}_{label}_kernel_stats.csv").write_text(f.read_text())
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if any(f_ in row.get("Name", "") for f_ in filters):
                stats[row["Name"].split("(")[0]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
summary = {"_comment": comment + "; "
                       "FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE reads half of a wide coalesced stream on gfx950, MI355X_MICROARCH.md)",
           "counters": res, "kernel_stats": stats}
(dst / f"{tag}_{label}_pmc.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary, indent=1)[:3000])
