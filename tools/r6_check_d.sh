mkdir -p gpurun_out/r6d; export HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r6d;
(time timeout 600 python -m pytest tests/test_gpu_fstep.py -q -m gpu --durations=5 -x) > $O/t_fstep.log 2>&1; echo "fstep rc=$?";
(time timeout 900 python -m pytest tests/test_gpu_state_machine.py tests/test_gpu_lookahead.py tests/test_gpu_fold_scale.py -q -m gpu --durations=5) > $O/t_sm.log 2>&1; echo "sm rc=$?";
timeout 600 python tools/small_n.py $O/small_n.jsonl > $O/small_n.out 2> $O/small_n.err; echo "small_n rc=$?";
for rep in 1 2 3; do for sw in 1 2; do timeout 200 python bench.py --steps 5 --warmup 1 --no-configs --no-sharded-leg --no-cpu-baseline --no-strict-leg --opt spmv_dia_sw=$sw > $O/ab_sw${sw}_$rep.json 2> /dev/null; python - <<PY
import json
d=json.loads([l for l in open("$O/ab_sw${sw}_$rep.json") if l.startswith("{")][-1])
print("rep $rep sw=$sw", d["value"], d["ms_per_step"])
PY
done; done
tail -n 25 $O/t_fstep.log | cut -c1-250; tail -n 12 $O/t_sm.log | cut -c1-250; cat $O/small_n.jsonl; tail -3 $O/small_n.err
