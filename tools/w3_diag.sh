export HSA_ENABLE_IPC_MODE_LEGACY=0 KK_XSYNC_DEBUG=1
make -s -C tests/fake_rccl
for cus in 72 72 64; do
  d=$(mktemp -d /tmp/w3.XXXX)
  export KK_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so KK_FAKE_RCCL_DIR=$d KK_FAKE_RCCL_TIMEOUT=90 KK_NUM_CUS=$cus
  for r in 0 1 2; do timeout 300 python tests/world2_worker.py xsync $r 3 $d > gpurun_out/c1/w3_${cus}_r$r.log 2>&1 & done
  wait
  echo "== num_cus $cus"; grep -h "xsync rank\|Error\|OK\|kk_xsync" gpurun_out/c1/w3_${cus}_r*.log | grep -v "timeouts 0.0" | tail -30
done
