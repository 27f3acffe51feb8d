cd /root/repo
export KK_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so KK_FAKE_RCCL_TIMEOUT=15 KK_FAKE_RCCL_TRACE=1
D=$(mktemp -d)
python tests/world2_worker.py block 0 2 $D > gpurun_out/r3a/block0.log 2>&1 &
python tests/world2_worker.py block 1 2 $D > gpurun_out/r3a/block1.log 2>&1
wait
tail -30 gpurun_out/r3a/block0.log; echo ======; tail -30 gpurun_out/r3a/block1.log
