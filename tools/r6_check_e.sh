mkdir -p gpurun_out/r6e; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6e;
(time timeout 600 python -m pytest tests/test_gpu_fstep.py -q -m gpu --durations=5) > $O/t_fstep.log 2>&1; echo "fstep rc=$?";
timeout 600 python tools/fstep_probe.py > $O/fstep_probe.jsonl 2> $O/fstep_probe.err; echo "probe rc=$?";
cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -- python /root/repo/tools/fstep_probe.py 102400 > /root/repo/$O/prof.out 2> /root/repo/$O/prof.err; cd /root/repo; echo "prof rc=$?";
tail -n 12 $O/t_fstep.log | cut -c1-250; cat $O/fstep_probe.jsonl; tail -3 $O/fstep_probe.err; find $O/prof -name "*kernel_stats.csv" | head -2 | xargs -I{} sh -c 'head -12 {} | cut -c1-200'
