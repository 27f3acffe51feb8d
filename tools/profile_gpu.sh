#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then
# separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC has 4 slots, 3 + 2 needed), for the headline
# workload and -- unless NOLEGS=1 -- for every entry of the line's `configs` block (bench.py --only-leg).
# usage (on the GPU box, from the repo root): bash tools/profile_gpu.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-sharded-leg $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
if [ "${NOLEGS:-0}" != "1" ]; then
  for LEG in ${LEGS:-lanczos_ell gmres block gkl}; do
    LB="python $REPO/bench.py --only-leg $LEG --config-steps 2"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$LEG" -- $LB > "$OUT/leg_${LEG}_trace.json" 2> "$OUT/leg_${LEG}_trace.err"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$LEG" -- $LB > "$OUT/leg_${LEG}_fetch.json" 2> "$OUT/leg_${LEG}_fetch.err"
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$LEG" -- $LB > "$OUT/leg_${LEG}_write.json" 2> "$OUT/leg_${LEG}_write.err"
  done
fi
cd "$REPO"
find "$OUT" -name "*.csv" | head -80
python tools/summarize_prof.py "$OUT" "$TAG"
