mkdir -p gpurun_out/r6j; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6j;
(time timeout 600 python -m pytest tests/test_gpu_panel.py -q -m gpu -k "applies_the_stencil" --durations=3) > $O/t_pa.log 2>&1; echo "pa rc=$?"; tail -n 25 $O/t_pa.log | cut -c1-220
(time timeout 900 python -m pytest tests/test_gpu_lookahead.py tests/test_gpu_panel.py tests/test_gpu_state_machine.py tests/test_gpu_fullsize.py -q -m gpu -k "not applies_the_stencil and not blocklanczos and not gkl" -x --durations=3) > $O/t_rest.log 2>&1; echo "rest rc=$?"; tail -n 8 $O/t_rest.log | cut -c1-220
timeout 300 python tools/gmres_ab.py base panel_apply=0 panel_apply=1 > $O/gmres_ab.jsonl 2> $O/gmres_ab.err; cat $O/gmres_ab.jsonl; tail -2 $O/gmres_ab.err
