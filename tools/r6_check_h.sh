mkdir -p gpurun_out/r6h; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6h;
(time timeout 3000 python -m pytest tests -m gpu -q --durations=25) > $O/t_allgpu.log 2>&1; echo "allgpu rc=$?"; tail -n 45 $O/t_allgpu.log | cut -c1-220
