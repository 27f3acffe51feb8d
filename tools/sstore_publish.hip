// Can a block's partial be PUBLISHED past its own queued loads?  A CU's vector-memory instructions leave through one in-order queue:
// in k_mgs_panel a publication (sc1 store) issued behind the ~128 KB of loads of the next panel reaches the fabric when most of them
// have been served, which is why the kernel publishes FIRST and requests the panel afterwards -- and why the memory pipe runs empty
// once per panel (dots + hand-off + publication happen with nothing in flight).  The scalar unit has its own path to the L2
// (s_store_dwordx4 through the scalar data cache, s_dcache_wb): if a granule stored that way into FINE-GRAINED memory is seen by the
// other blocks' sc1 sweeps, the panel can be requested before the inner products and the bubble closes.
//   variants per iteration (256 blocks x 512 threads, 16 x 16-byte loads per thread = one panel of two 2M-row vectors):
//     0  dots -> publish (vector sc1) -> request next panel -> sweep -> update                 (k_mgs_panel today)
//     1  request next panel -> dots -> publish (vector sc1) -> sweep -> update                 (publication queued behind the panel)
//     2  request next panel -> dots -> publish (SCALAR store + s_dcache_wb) -> sweep -> update
//     3  no reduction at all (the bare stream)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/sstore_publish tools/sstore_publish.hip ; run: tools/bin/sstore_publish
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define NLOAD 16

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>
__global__ __launch_bounds__(512) void k_iter(const d2* __restrict__ data, size_t n16_per_block, char* gran, int steps, int* bad, long long* clk, double* sink) {
    __shared__ double smA[8];
    __shared__ double smB[2];
    const int G = gridDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned gbytes = (unsigned)(2 * G * 16);
    const d2* base = data + (size_t)blockIdx.x * n16_per_block;
    const size_t chunk = (size_t)512 * NLOAD;                 // double2 per iteration and block
    const size_t nchunk = n16_per_block / chunk;
    d2 cur[NLOAD], nxt[NLOAD];
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) cur[u] = __builtin_nontemporal_load(base + (size_t)u * 512 + tid);
    double wacc = 0;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        const unsigned epoch = (unsigned)s + 1u;
        const d2* src = base + ((size_t)(s + 1) % nchunk) * chunk;
        if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int u = 0; u < NLOAD; ++u) nxt[u] = __builtin_nontemporal_load(src + (size_t)u * 512 + tid);
        }
        // "inner products" of the landed panel
        double a = 0;
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) a += cur[u].x * 1e-3 + cur[u].y * 1e-3;
        double total = 0;
        if (MODE != 3) {
            double t = a;
            for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) smA[wave] = t;
            lds_barrier();
            if (tid < 64) {
                double b = 0;
                for (int k = 0; k < 8; ++k) b += smA[k];
                b = (double)(blockIdx.x + 1) * 0.5 + (double)(s & 7) + 0.0 * b;      // a known partial (the streamed values only keep the loads alive)
                const unsigned long long bits = (unsigned long long)__double_as_longlong(b);
                const unsigned off = (unsigned)(((size_t)(s & 1) * G + blockIdx.x) * 16);
                if (MODE == 2) {
                    v4u t4;
                    t4.x = __builtin_amdgcn_readfirstlane(epoch); t4.y = __builtin_amdgcn_readfirstlane((unsigned)(bits >> 32));
                    t4.z = __builtin_amdgcn_readfirstlane((unsigned)bits); t4.w = __builtin_amdgcn_readfirstlane(epoch);
                    const unsigned long long addr = (unsigned long long)(gran + off);
                    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr), hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
                    const unsigned long long sa = ((unsigned long long)hi << 32) | lo;
                    asm volatile("s_store_dwordx4 %0, %1, 0x0 glc\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::"s"(t4), "s"(sa) : "memory");
                } else if (lane == 0) {
                    v4u t4; t4.x = epoch; t4.y = (unsigned)(bits >> 32); t4.z = (unsigned)bits; t4.w = epoch;
                    __builtin_amdgcn_raw_buffer_store_b128(t4, rsrc(gran, gbytes), off, 0, 16);
                }
            }
            lds_barrier();
        }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < NLOAD; ++u) nxt[u] = __builtin_nontemporal_load(src + (size_t)u * 512 + tid);
        }
        if (MODE != 3) {
            if (tid < 64) {
                const __amdgpu_buffer_rsrc_t r = rsrc(gran, gbytes);
                for (;;) {
                    asm volatile("" ::: "memory");
                    bool ok = true; double x = 0;
                    v4u t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int b = i * 64 + lane; const int bb = b < G ? b : 0; t[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(((size_t)(s & 1) * G + bb) * 16), 0, 17); }
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (i * 64 + lane < G) {
                        ok = ok && t[i].x == epoch && t[i].w == epoch;
                        x += __longlong_as_double((long long)(((unsigned long long)t[i].y << 32) | t[i].z));
                    }
                    if (__all(ok)) { for (int o = 32; o; o >>= 1) x += __shfl_xor(x, o); total = x; break; }
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 500000000ll) { if (lane == 0) atomicAdd(bad, 1000000); total = -1; break; }
                }
                if (lane == 0) smB[0] = total;
            }
            lds_barrier();
            total = smB[0];
            const double expect = 0.25 * (double)G * (double)(G + 1) + (double)G * (double)(s & 7);
            if (tid == 0 && total != expect) atomicAdd(bad, 1);
            if (total < 0) break;
        }
        // "update": uses the totals and the panel, frees its registers
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) { wacc += cur[u].x * total * 1e-30; cur[u] = nxt[u]; }
    }
    if (tid == 0 && blockIdx.x == 0) *clk = wall_clock64() - t0;
    if (wacc == 12345.678) sink[0] = wacc;
}

template <int MODE>
static void run(const char* name, int G, int steps, const d2* data, size_t n16_per_block, bool fine) {
    char* gran; int* bad; long long* clk; double* sink;
    if (fine) CK(hipExtMallocWithFlags((void**)&gran, 2 * G * 16 + 4096, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void**)&gran, 2 * G * 16 + 4096));
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&clk, 8)); CK(hipMalloc(&sink, 8));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(gran, 0, 2 * G * 16 + 4096)); CK(hipMemset(bad, 0, 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_iter<MODE>, dim3(G), dim3(512), 0, 0, data, n16_per_block, gran, steps, bad, clk, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hbad; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        if (rep == 2) printf("{\"variant\": \"%s\", \"fine_grained_area\": %d, \"blocks\": %d, \"steps\": %d, \"us_per_iteration\": %.3f, \"TBps\": %.2f, \"wrong_totals\": %d}\n", name, (int)fine, G, steps,
                             ms * 1e3 / steps, (double)G * 512 * NLOAD * 16 / (ms * 1e-3 / steps) / 1e12, hbad);
    }
    CK(hipFree(gran)); CK(hipFree(bad)); CK(hipFree(clk)); CK(hipFree(sink));
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int G = p.multiProcessorCount, steps = 4000;
    const size_t n16_per_block = (size_t)512 * NLOAD * 64;      // 64 chunks of 128 KB per block: 8 MB per block, 2 GB in all (beyond the Infinity Cache)
    d2* data;
    CK(hipMalloc(&data, (size_t)G * n16_per_block * 16));
    CK(hipMemset(data, 0, (size_t)G * n16_per_block * 16));
    run<3>("3: bare stream, no reduction", G, steps, data, n16_per_block, true);
    run<0>("0: dots, publish (vector sc1), request panel, sweep", G, steps, data, n16_per_block, true);
    run<0>("0: the same, coarse-grained area", G, steps, data, n16_per_block, false);
    run<1>("1: request panel, dots, publish (vector sc1), sweep", G, steps, data, n16_per_block, true);
    run<2>("2: request panel, dots, publish (scalar store + s_dcache_wb), sweep", G, steps, data, n16_per_block, true);
    run<2>("2: the same, coarse-grained area", G, steps, data, n16_per_block, false);
    return 0;
}
