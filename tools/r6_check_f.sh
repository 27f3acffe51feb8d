mkdir -p gpurun_out/r6f; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6f;
(time timeout 600 python -m pytest tests/test_gpu_fstep.py -q -m gpu --durations=3) > $O/t_fstep.log 2>&1; echo "fstep rc=$?";
timeout 900 python tools/fstep_probe.py 1024 10000 40000 102400 200000 > $O/fstep_probe.jsonl 2> $O/fstep_probe.err; echo "probe rc=$?";
tail -n 8 $O/t_fstep.log | cut -c1-250; cat $O/fstep_probe.jsonl; tail -3 $O/fstep_probe.err
