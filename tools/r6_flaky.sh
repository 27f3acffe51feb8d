mkdir -p gpurun_out/r6z; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6z;
for rep in 1 2; do (time timeout 1500 python -m pytest tests/test_gpu_world2.py -q -m gpu -x) > $O/w_$rep.log 2>&1; echo "world rep $rep rc=$?"; tail -n 3 $O/w_$rep.log | cut -c1-200; done
for rep in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_fstep.py tests/test_gpu_state_machine.py tests/test_gpu_lookahead.py tests/test_gpu_panel.py tests/test_gpu_cotenant.py tests/test_gpu_persist_recovery.py -q -m gpu -x > $O/s_$rep.log 2>&1; echo "small rep $rep rc=$?"; tail -n 2 $O/s_$rep.log | cut -c1-200; done
