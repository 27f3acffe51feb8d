mkdir -p gpurun_out/r6w; export HSA_ENABLE_IPC_MODE_LEGACY=0
for t in 256 512; do for b in 32 64 128; do FSTEP_THREADS=$t FSTEP_BLOCKS=$b python tools/fstep_where.py >> gpurun_out/r6w/where2.jsonl 2>> gpurun_out/r6w/where2.err; done; done
cat gpurun_out/r6w/where2.jsonl; tail -3 gpurun_out/r6w/where2.err
