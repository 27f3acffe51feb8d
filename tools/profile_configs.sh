#!/bin/bash
# rocprofv3 kernel-trace stats of the secondary configs (3, 5, restart path) for profiles/.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_cfg; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg35" -- python $REPO/tools/bench_configs.py gmres block > "$OUT/cfg35.json" 2> "$OUT/cfg35.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/restart" -- python $REPO/tools/restart_bench.py > "$OUT/restart.txt" 2> "$OUT/restart.err"
cd "$REPO"; find "$OUT" -name "*kernel_stats.csv" | head
# config 4 at full size (column-tiled SpMV) and the short-recurrence solvers of SURVEY 8(f)-3
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg4" -- python $REPO/tools/bench_configs.py gkl --full > "$OUT/cfg4.json" 2> "$OUT/cfg4.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/short" -- python $REPO/tools/bench_configs.py cg bicgstab lsmr > "$OUT/short.json" 2> "$OUT/short.err"
cd "$REPO"; find "$OUT" -name "*kernel_stats.csv" | head
