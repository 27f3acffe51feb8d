mkdir -p gpurun_out/r6n; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6n;
(time timeout 900 python bench.py) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python tools/benchsum.py $O/bench_default.json | head -40
