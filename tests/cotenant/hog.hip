// TEST INFRASTRUCTURE -- never shipped, never linked into libkrylov_hip.so.
// libhog.so: a co-tenant for the persistent kernels.  hog_start(blocks, ms) launches, on a stream of its own, `blocks`
// workgroups that each take a WHOLE CU (1024 threads = 4 waves per SIMD x 128 VGPRs = the whole register file, + 152 KB of the
// 160 KB of LDS: nothing else fits beside one) and spin for `ms` milliseconds of the wall clock: while
// they run the chip has `blocks` CUs less than hipDeviceProp says, which is what an ordinary launch of one block per CU
// (csrc/kk_kernels_persist.hip::kk_launch_resident) cannot know.  tests/test_gpu_cotenant.py drives it.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void k_hog(long long ticks, int* sink) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) lds[0] = 1;
    asm volatile("v_mov_b32 v120, 0" ::: "v120");   // 121 -> 128 allocated VGPRs per wave
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (ticks < 0) sink[0] = lds[0];
}

static hipStream_t g_stream = nullptr;
static int* g_sink = nullptr;

extern "C" __attribute__((visibility("default"))) int hog_start(int blocks, double ms) {
    if (!g_stream) {
        if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 1;
        if (hipMalloc(&g_sink, 64) != hipSuccess) return 2;
        if (hipFuncSetAttribute((const void*)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) return 3;
    }
    hipLaunchKernelGGL(k_hog, dim3(blocks), dim3(1024), 152 * 1024, g_stream, (long long)(ms * 1e5), g_sink);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
extern "C" __attribute__((visibility("default"))) int hog_wait(void) {
    return g_stream && hipStreamSynchronize(g_stream) != hipSuccess ? 1 : 0;
}
