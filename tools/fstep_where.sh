# timing variants of the one-launch step (see tools/fstep_where.py).  Build the variant libraries on the CPU box first (they travel with the snapshot):
#   cd krylovkit.jl_amd && mkdir -p lib_exp && for e in 9 10 12 15; do
#     hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DKK_FS_EXP=$e -c csrc/kk_kernels_fstep.hip -o build/fs_exp_$e.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o lib_exp/libkrylov_hip_exp$e.so $(ls build/kk_*.o | grep -v kk_kernels_fstep.o) build/fs_exp_$e.o -ldl; done
mkdir -p gpurun_out/r6w; export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/fstep_where.py > gpurun_out/r6w/where.jsonl 2> gpurun_out/r6w/where.err
for e in 9 10 12 15; do KRYLOV_HIP_LIB=$PWD/krylovkit.jl_amd/lib_exp/libkrylov_hip_exp$e.so python tools/fstep_where.py >> gpurun_out/r6w/where.jsonl 2>> gpurun_out/r6w/where.err; done
cat gpurun_out/r6w/where.jsonl; tail -3 gpurun_out/r6w/where.err
