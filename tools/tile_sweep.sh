#!/bin/bash
# run the headline bench once per tile-shape variant library (lib/libkk_*.so built with
# make DEFS="-DKK_RG_P=.. -DKK_CB_P=.. -DKK_RG_U=.. -DKK_CB_U=.."); prints it/s + per-kernel ms
mkdir -p gpurun_out
for rep in 1; do
for f in krylovkit.jl_amd/lib/libkk_*.so; do
  v=$(basename $f .so)
  KRYLOV_HIP_LIB=$PWD/$f timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --orth ${ORTH:-cgs2} 2>/dev/null | tail -1 > gpurun_out/sweep_$v.json
  python - "$v" gpurun_out/sweep_$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print(sys.argv[1], round(d["value"],1), d["roofline"]["timed_region_kernel_ms"])
PY
done
done
