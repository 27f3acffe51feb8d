#!/bin/bash
# Counter evidence for the column-tiled SpMV of config 4 (k_spmv_sellw): one rocprofv3 --pmc pass per counter group (TCC has
# 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2; nothing is combined with a trace domain other than --kernel-trace).
# usage (GPU box, repo root): bash tools/profile_cfg4.sh <tag>     -> gpurun_out/prof_cfg4_<tag>/summary/<tag>_cfg4_pmc.json
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_cfg4_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/spmv_bench.py 393216"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- $CMD > "$OUT/pass$i.txt" 2> "$OUT/pass$i.err" || echo "pass $i ($grp) failed"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.txt" 2> "$OUT/trace.err"
cd "$REPO"
python tools/summarize_cfg4.py "$OUT" "$TAG"
