mkdir -p gpurun_out/r6i; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6i;
(time timeout 600 python -m pytest tests/test_gpu_fstep.py -q -m gpu --durations=3) > $O/t_fstep.log 2>&1; echo "fstep rc=$?"; tail -n 30 $O/t_fstep.log | cut -c1-220
(time timeout 900 python -m pytest tests/test_gpu_state_machine.py tests/test_gpu_lookahead.py tests/test_gpu_parity.py tests/test_gpu_random.py -q -m gpu -x --durations=3) > $O/t_par.log 2>&1; echo "par rc=$?"; tail -n 12 $O/t_par.log | cut -c1-220
