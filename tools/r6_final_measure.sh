# round-6 evidence on the FINAL kernel sources: rocprofv3 traces + PMC passes (headline and every leg), the default bench line, probes
mkdir -p gpurun_out/r6m; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6m;
timeout 1800 bash tools/profile_gpu.sh r06 > $O/prof.log 2>&1; echo "prof rc=$?"; tail -2 $O/prof.log | cut -c1-300
cp gpurun_out/prof_r06/summary/traffic.json gpurun_out/prof_r06/summary/traffic_configs.json profiles/ 2>/dev/null
(time timeout 900 python bench.py) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python tools/benchsum.py $O/bench_default.json | head -30
timeout 600 python tools/fstep_probe.py 1024 10000 40000 102400 > $O/fstep_probe.jsonl 2> $O/fstep_probe.err; echo "probe rc=$?"
timeout 600 python tools/small_n.py $O/small_n.jsonl > /dev/null 2> $O/small_n.err; echo "small_n rc=$?"
