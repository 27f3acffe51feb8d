mkdir -p gpurun_out/r6h; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6h;
(time timeout 3000 python -m pytest tests -m gpu -q --durations=12) > $O/t_allgpu.log 2>&1; echo "allgpu rc=$?"; tail -n 30 $O/t_allgpu.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
