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
# (rocprofv3 follows child processes -- bench.py's read-ceiling probe tools/bin/hbm_peak -- and writes one file per process: the bench's
#  own is the big one)
ks = find("trace", "*kernel_stats.csv")
if ks:
    (dst / f"{tag}_kernel_stats.csv").write_text(max(ks, key=lambda f: f.stat().st_size).read_text())
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

# 3. HBM bytes per launch for bench.py's roofline.traffic: (2*FETCH_SIZE + WRITE_SIZE) * 1024 per the gfx950 correction of
# MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read stream), stamped with the hash of the kernel
# sources it was measured on -- bench.py drops the figure when the sources have changed since.
if "FETCH_SIZE" in res and "WRITE_SIZE" in res and "--no-traffic" not in sys.argv:
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from bench import kernel_source_sha  # noqa: E402
    traffic = {"_comment": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 2 "
                           "--warmup 1`, (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes for gfx950, averaged over the "
                           "launches of the run", "source_sha": kernel_source_sha(), "stamped_by": f"tools/profile_gpu.sh {tag}"}
    for short in ("k_project", "k_unproject", "k_spmv_ell", "k_spmv_dia", "k_scal", "k_mgs_step", "k_mgs_persist", "k_unproj_proj"):
        f = [v for k, v in res["FETCH_SIZE"].items() if k.startswith("void " + short) or k.startswith(short)]
        w = [v for k, v in res["WRITE_SIZE"].items() if k.startswith("void " + short) or k.startswith(short)]
        if f and w:
            nf = sum(v["dispatches"] for v in f)
            nw = sum(v["dispatches"] for v in w)
            traffic[short] = round((2 * sum(v["sum"] for v in f) / nf + sum(v["sum"] for v in w) / nw) * 1024)
    # average duration per launch of the same classes from the kernel-trace pass (rocprofv3 --kernel-trace --stats of the same
    # command): what bench.py quotes for SHORT kernels, whose HIP-event brackets in the stream also contain the launch gaps
    avg_us = {}
    for f in find("trace", "*kernel_stats.csv"):
        with open(f) as fh:
            agg = defaultdict(lambda: [0, 0.0])
            for row in csv.DictReader(fh):
                nm = row["Name"]
                nm = nm[5:] if nm.startswith("void ") else nm
                for short in ("k_project", "k_unproject", "k_spmv_ell", "k_spmv_dia", "k_scal", "k_mgs_step", "k_mgs_persist", "k_unproj_proj", "k_mgs_panel"):
                    # (k_spmv_dia_sw is the sweeping form of the same class: bench.py brackets both as "k_spmv_dia")
                    if nm.startswith(short + "<") or nm.startswith(short + "(") or (short == "k_spmv_dia" and nm.startswith("k_spmv_dia_sw<")):
                        agg[short][0] += int(row["Calls"]); agg[short][1] += float(row["TotalDurationNs"])
            for k_, (n_, t_) in agg.items():
                if n_:
                    avg_us[k_] = round(t_ / n_ / 1e3, 3)
    traffic["rocprof_avg_us"] = avg_us
    (dst / "traffic.json").write_text(json.dumps(traffic, indent=1))
    print("traffic.json:", traffic)


# 4. the same per entry of the bench line's `configs` block: bytes per SWEEP and kernel class ((2*FETCH_SIZE + WRITE_SIZE)*1024
# summed over the launches of a class, divided by the sweeps `bench.py --only-leg` ran) -> traffic_configs.json
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import LEG_CLASSES, kernel_source_sha  # noqa: E402,F811


def per_class(sub, ctr):
    acc = defaultdict(float)
    for f in find(sub, "*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != ctr:
                    continue
                k = row.get("Kernel_Name", "?").split("(")[0]
                k = k[5:] if k.startswith("void ") else k
                for cls in sorted(LEG_CLASSES, key=len, reverse=True):
                    if k.startswith(cls):
                        acc[cls] += float(row["Counter_Value"])
                        break
    return acc


cfgs = {}
for leg in ("lanczos_ell", "gmres", "block", "gkl"):
    jf, jw = out / f"leg_{leg}_fetch.json", out / f"leg_{leg}_write.json"
    if not (jf.exists() and jw.exists()):
        continue
    try:
        sweeps_f = json.loads([l for l in jf.read_text().splitlines() if l.startswith("{")][-1])["sweeps"]
        sweeps_w = json.loads([l for l in jw.read_text().splitlines() if l.startswith("{")][-1])["sweeps"]
    except Exception as e:
        print("leg", leg, "no JSON line:", e)
        continue
    fe, wr = per_class(f"pmc_fetch_{leg}", "FETCH_SIZE"), per_class(f"pmc_write_{leg}", "WRITE_SIZE")
    cfgs[leg] = {cls: round((2 * fe.get(cls, 0.0) / sweeps_f + wr.get(cls, 0.0) / sweeps_w) * 1024) for cls in set(fe) | set(wr)}
    cfgs[leg]["_sweeps_profiled"] = sweeps_f
    for f in find(f"trace_{leg}", "*kernel_stats.csv"):
        (dst / f"{tag}_{leg}_kernel_stats.csv").write_text(f.read_text())
if cfgs and "--no-traffic" not in sys.argv:
    tc = {"_comment": "HBM bytes per SWEEP and kernel class of every entry of bench.py's `configs` block, from separate rocprofv3 --pmc FETCH_SIZE / "
                      "WRITE_SIZE passes of `bench.py --only-leg <name>`: (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md) "
                      "summed over the launches of the class, divided by the sweeps of the run",
          "source_sha": kernel_source_sha(), "stamped_by": f"tools/profile_gpu.sh {tag}", "configs": cfgs}
    (dst / "traffic_configs.json").write_text(json.dumps(tc, indent=1))
    print("traffic_configs.json:", json.dumps(cfgs))
