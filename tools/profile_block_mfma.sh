#!/bin/bash
# Is the two-panel Gram kernel of the block step (k_block_gram2) limited by the f64 matrix cores or by memory?  One
# rocprofv3 --pmc pass with the SQ busy / MFMA-busy counters (nothing combined with a trace domain other than --kernel-trace).
# usage (GPU box, repo root): bash tools/profile_block_mfma.sh <tag>
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_block_mfma_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export KK_BENCH_BLOCK_MODES=1
CMD="python $REPO/tools/bench_configs.py block"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pass1" -- $CMD > "$OUT/pass1.txt" 2> "$OUT/pass1.err" || echo "pass 1 failed"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$OUT/pass2" -- $CMD > "$OUT/pass2.txt" 2> "$OUT/pass2.err" || echo "pass 2 failed"
cd "$REPO"
python - "$OUT" "$TAG" <<'PY'
import csv, json, sys
from collections import defaultdict
from pathlib import Path
out, tag = Path(sys.argv[1]), sys.argv[2]
res = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in out.rglob("*counter_collection.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            if not ("k_block" in k or "k_spmm" in k):
                continue
            a = res[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
summary = {k: {c: {"avg_per_dispatch": v[0] / v[1], "dispatches": v[1]} for c, v in d.items()} for k, d in res.items()}
(out / f"{tag}_cfg5_mfma_pmc.json").write_text(json.dumps(summary, indent=1))
for k, d in summary.items():
    print(k[:70], {c: round(v["avg_per_dispatch"]) for c, v in d.items()})
PY
