#!/bin/bash
# two ranks of tools/bin/xs_probe on the one GPU of a gpurun box: without / with a CU mask, fine-grained / plain memory
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/xs_probe}; mkdir -p $OUT
run() {  # name G rounds fine mask0 mask1
  local d=$(mktemp -d /tmp/xsp.XXXXXX)
  ( [ -n "$5" ] && export HSA_CU_MASK="$5"; timeout 60 tools/bin/xs_probe 0 $2 $3 $d $4 ) > $OUT/$1.r0 2>&1 &
  local p0=$!
  ( [ -n "$6" ] && export HSA_CU_MASK="$6"; timeout 60 tools/bin/xs_probe 1 $2 $3 $d $4 ) > $OUT/$1.r1 2>&1
  wait $p0
  echo "== $1"; cat $OUT/$1.r0 $OUT/$1.r1
  rm -rf $d
}
run g1_fine 1 20000 1 "" ""
run g1_plain 1 20000 0 "" ""
run g128_nomask 128 20000 1 "" ""
run g128_mask 128 20000 1 "0:0-127" "0:128-255"
run g256_nomask 256 2000 1 "" ""
