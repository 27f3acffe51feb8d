#!/bin/bash
# One gpurun call: new tests first, then bench lines.  Everything lands in gpurun_out/<tag>/.
TAG=${1:-chk}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for step in "$@"; do
  case $step in
    native)   timeout 900 python -m pytest tests/test_gpu_native_comm.py -x -q > $OUT/t_native.log 2>&1; echo "native rc=$?" ;;
    hooks)    timeout 900 python -m pytest tests/test_gpu_sharded_hooks.py -x -q > $OUT/t_hooks.log 2>&1; echo "hooks rc=$?" ;;
    parityfs) timeout 1500 python -m pytest tests/test_gpu_fullsize.py -k parity -x -q > $OUT/t_parityfs.log 2>&1; echo "parityfs rc=$?" ;;
    fullsize) timeout 1800 python -m pytest tests/test_gpu_fullsize.py -x -q > $OUT/t_fullsize.log 2>&1; echo "fullsize rc=$?" ;;
    parity)   timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q > $OUT/t_parity.log 2>&1; echo "parity rc=$?" ;;
    allgpu)   timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/t_allgpu.log 2>&1; echo "allgpu rc=$?" ;;
    bench)    timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json ;;
    benchdist) KK_BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; echo "benchdist rc=$?"; tail -c 1500 $OUT/bench_forcedist.json ;;
    bench2)   timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; echo "bench2 rc=$? (expected to fail on a 1-GPU box)"; tail -5 $OUT/bench_gpus2.err ;;
    gkl)      KK_BENCH_FORCE_DIST=1 timeout 900 python bench.py --config gkl --steps 2 --warmup 1 > $OUT/bench_gkl_dist.json 2> $OUT/bench_gkl_dist.err; echo "gkl dist rc=$?"; tail -c 1200 $OUT/bench_gkl_dist.json
              timeout 900 python bench.py --config gkl --steps 2 --warmup 1 > $OUT/bench_gkl.json 2> $OUT/bench_gkl.err; echo "gkl rc=$?"; tail -c 1200 $OUT/bench_gkl.json ;;
    block)    KK_BENCH_FORCE_DIST=1 timeout 900 python bench.py --config block --steps 2 --warmup 1 > $OUT/bench_block_dist.json 2> $OUT/bench_block_dist.err; echo "block dist rc=$?"; tail -c 1200 $OUT/bench_block_dist.json
              timeout 900 python bench.py --config block --steps 2 --warmup 1 > $OUT/bench_block.json 2> $OUT/bench_block.err; echo "block rc=$?"; tail -c 1200 $OUT/bench_block.json ;;
    configs)  timeout 1200 python tools/bench_configs.py gmres block > $OUT/configs.jsonl 2> $OUT/configs.err; echo "configs rc=$?"; cat $OUT/configs.jsonl ;;
    strict)   timeout 900 python tools/strict_sweep.py > $OUT/strict.jsonl 2> $OUT/strict.err; echo "strict rc=$?"; cat $OUT/strict.jsonl; tail -3 $OUT/strict.err ;;
    orthtests) timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "orthogonalize or lanczos or arnoldi or gkl or mgs" > $OUT/t_orth.log 2>&1; echo "orthtests rc=$?" ;;
    blocksweep) timeout 900 python tools/block_sweep.py > $OUT/block_sweep.jsonl 2> $OUT/block_sweep.err; echo "blocksweep rc=$?"; cat $OUT/block_sweep.jsonl; tail -3 $OUT/block_sweep.err ;;
    blocktests) timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "block" > $OUT/t_block.log 2>&1; echo "blocktests rc=$?" ;;
    blockmicro) timeout 900 python tools/block_micro.py > $OUT/block_micro.jsonl 2> $OUT/block_micro.err; echo "blockmicro rc=$?"; cat $OUT/block_micro.jsonl; tail -3 $OUT/block_micro.err ;;
    advice)   timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end or wide_basis" > $OUT/t_advice.log 2>&1; echo "advice rc=$?" ;;
    cfg4pmc)  bash tools/profile_cfg4.sh $TAG > $OUT/cfg4pmc.log 2>&1; echo "cfg4pmc rc=$?"; tail -40 $OUT/cfg4pmc.log ;;
    cfg5pmc)  bash tools/profile_block.sh $TAG > $OUT/cfg5pmc.log 2>&1; echo "cfg5pmc rc=$?"; tail -60 $OUT/cfg5pmc.log ;;
    prof)     bash tools/profile_gpu.sh $TAG > $OUT/prof.log 2>&1; echo "prof rc=$?"; tail -5 $OUT/prof.log ;;
    *) echo "unknown step $step" ;;
  esac
done
for f in $OUT/t_*.log; do [ -f "$f" ] && { echo "== $f"; tail -n 15 "$f"; }; done
