mkdir -p gpurun_out/r6k; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6k;
(time timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_panel.py -q -m gpu -k "apply_inside or applies_the_stencil or lost_launch or config2_lanczos_10M_parity_with_cpu or register_file" -x --durations=4) > $O/t_a.log 2>&1; echo "a rc=$?"; tail -n 25 $O/t_a.log | cut -c1-240
for rep in 1 2 3; do for pa in 1 0; do timeout 200 python bench.py --steps 5 --warmup 1 --no-configs --no-sharded-leg --no-cpu-baseline --no-strict-leg --opt persist_apply=$pa > $O/ab_${pa}_$rep.json 2> /dev/null; python - <<PY
import json
d=json.loads([l for l in open("$O/ab_${pa}_$rep.json") if l.startswith("{")][-1])
print("rep $rep persist_apply=$pa", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
done; done
