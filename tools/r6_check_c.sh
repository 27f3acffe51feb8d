mkdir -p gpurun_out/r6c; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6c;
(time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block_commit.py tests/test_gpu_fullsize.py -q -m gpu -k "block or constant_coefficient" --durations=5) > $O/t_blk.log 2>&1; echo "blk rc=$?";
(time timeout 400 python -m pytest tests/test_gpu_world2.py -q -m gpu -s -k "gave_up and 8" --durations=3) > $O/t_w8.log 2>&1; echo "w8 rc=$?";
for sw in 0 1 2; do timeout 200 python bench.py --steps 5 --warmup 1 --no-configs --no-sharded-leg --no-cpu-baseline --no-strict-leg --opt spmv_dia_sw=$sw > $O/ab_sw$sw.json 2> $O/ab_sw$sw.err; echo "ab $sw rc=$?"; done
NOLEGS=1 timeout 600 bash tools/profile_gpu.sh r06a > $O/prof.log 2>&1; echo "prof rc=$?";
NOLEGS=1 timeout 400 bash tools/profile_gpu.sh r06a_sw0 --opt spmv_dia_sw=0 > $O/prof0.log 2>&1; echo "prof0 rc=$?";
tail -n 8 $O/t_blk.log; tail -n 8 $O/t_w8.log | cut -c1-200;
for sw in 0 1 2; do python - <<PY
import json
d=json.loads([l for l in open("$O/ab_sw$sw.json") if l.startswith("{")][-1])
print("sw=$sw", d["value"], d["ms_per_step"], d["roofline"].get("second_kernel",{}).get("avg_launch_ms"))
PY
done
grep -h "k_spmv_dia\|k_mgs_persist" gpurun_out/prof_r06a/summary/*kernel_stats.csv gpurun_out/prof_r06a_sw0/summary/*kernel_stats.csv | cut -c1-200
cat gpurun_out/prof_r06a/summary/traffic.json 2>/dev/null | head -30
