"""Condenses rocprofv3 output (kernel stats + PMC passes) into small files for profiles/."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

out, tag = Path(sys.argv[1]), sys.argv[2]
dst = out / "summary"
dst.mkdir(exist_ok=True)


def find(sub, pat):
    return sorted((out / sub).rglob(pat))


# 1. kernel stats
for f in find("trace", "*kernel_stats.csv"):
    (dst / f"{tag}_kernel_stats.csv").write_text(f.read_text())
# 2. PMC: sum counter per kernel name, average per dispatch
res = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in find(sub, "*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != ctr:
                    continue
                k = row.get("Kernel_Name", "?").split("(")[0]
                acc[k] += float(row["Counter_Value"])
                cnt[k] += 1
    res[ctr] = {k: {"dispatches": cnt[k], "sum": acc[k], "avg_per_dispatch": acc[k] / cnt[k]} for k in acc}
(dst / f"{tag}_pmc.json").write_text(json.dumps(res, indent=1))
for ctr, d in res.items():
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["sum"])[:8]:
        print(ctr, k[:60], v["dispatches"], f"{v['avg_per_dispatch']:.4g}")
