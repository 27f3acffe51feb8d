mkdir -p gpurun_out/r6g; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6g;
(time timeout 300 python -m pytest tests/test_gpu_fstep.py -q -m gpu) > $O/t_fstep.log 2>&1; echo "fstep rc=$?"; tail -3 $O/t_fstep.log
timeout 900 bash tools/profile_block.sh r06 > $O/profile_block.log 2>&1; echo "profile_block rc=$?"; tail -40 $O/profile_block.log | cut -c1-220
