#!/bin/bash
# Counter evidence for the BlockLanczos kernels of config 5 (k_spmm_ell, k_block_gram, k_block_update): one rocprofv3 --pmc
# pass per counter group, nothing combined with a trace domain other than --kernel-trace.
# usage (GPU box, repo root): bash tools/profile_block.sh <tag>   -> gpurun_out/prof_block_<tag>/summary/<tag>_cfg5_pmc.json
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_block_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export KK_BENCH_BLOCK_MODES=1
CMD="python $REPO/tools/bench_configs.py block"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- $CMD > "$OUT/pass$i.txt" 2> "$OUT/pass$i.err" || echo "pass $i ($grp) failed"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.txt" 2> "$OUT/trace.err"
cd "$REPO"
python tools/summarize_cfg4.py "$OUT" "$TAG" cfg5 "k_spmm,k_block" "config 5 (BlockLanczos, 10M rows, block size 16, krylovdim 100): per-dispatch counter averages"
