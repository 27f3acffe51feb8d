# timing variants of the one-launch step (see tools/fstep_where.py); the variant libraries are built on the CPU box:
#   for e in 9 10 12 15; do hipcc ... -DKK_FS_EXP=$e -c csrc/kk_kernels_fstep.hip ...; done   (krylovkit.jl_amd/lib_exp/)
mkdir -p gpurun_out/r6w; export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/fstep_where.py > gpurun_out/r6w/where.jsonl 2> gpurun_out/r6w/where.err
for e in 9 10 12 15; do KRYLOV_HIP_LIB=$PWD/krylovkit.jl_amd/lib_exp/libkrylov_hip_exp$e.so python tools/fstep_where.py >> gpurun_out/r6w/where.jsonl 2>> gpurun_out/r6w/where.err; done
cat gpurun_out/r6w/where.jsonl; tail -3 gpurun_out/r6w/where.err
