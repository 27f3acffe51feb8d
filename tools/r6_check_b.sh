mkdir -p gpurun_out/r6b; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6b;
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_state_machine.py tests/test_gpu_block_commit.py -q -m gpu -k "sweeping or constant_coefficient or grid_stencil or block" --durations=8) > $O/t_sw.log 2>&1; echo "sw rc=$?";
timeout 300 python tools/spmv_dia_sw_sweep.py > $O/sw_sweep.jsonl 2> $O/sw_sweep.err; echo "sweep rc=$?";
tools/bin/graph_launch_cost > $O/graph_cost.json 2>&1; echo "graph rc=$?";
tools/bin/gather_rate --json > $O/gather.json 2>&1;
(time timeout 700 python -m pytest tests/test_gpu_world2.py -q -m gpu -s -k "in_kernel and 8 or gave_up and 8 or random_interleavings" --durations=8) > $O/t_w8.log 2>&1; echo "w8 rc=$?";
(time timeout 400 python bench.py --steps 5 --warmup 1 --no-sharded-leg --parity-seeds 1 --config-steps 2) > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc=$?";
tail -n 14 $O/t_sw.log; cat $O/sw_sweep.jsonl; tail -3 $O/sw_sweep.err; cat $O/graph_cost.json $O/gather.json; tail -n 25 $O/t_w8.log | cut -c1-250; python tools/benchsum.py $O/bench_quick.json 2>/dev/null | head -40 || tail -c 1500 $O/bench_quick.json
