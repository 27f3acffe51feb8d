// Wall-clock trace of k_mgs_panel (csrc/kk_kernels_panel.hip compiled with -DKK_PANEL_TRACE): per panel, when block 0's
// data wave requested the next panel (0), had its partial sums (1), passed barrier 1 (2) / barrier 2 (3), finished the
// update (4); when its reduction wave had published (8) and had all totals (9), and how many sweep passes that took (10).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DKK_PANEL_TRACE -I krylovkit.jl_amd/csrc -o tools/bin/panel_trace tools/panel_trace.hip
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../krylovkit.jl_amd/csrc/kk_kernels_panel.hip"
void kk_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vprintf(fmt, a); va_end(a); printf("\n"); }
int kk_hip_fail(hipError_t e, const char* what, const char*, int line) { printf("HIP error %s: %s (line %d)\n", what, hipGetErrorString(e), line); return KK_ERR_HIP; }
int kk_launch_resident(kk_ctx ctx, const void* fn, int threads, void** args, size_t dyn, const char*) { return hipLaunchCooperativeKernel(fn, dim3(ctx->num_cus), dim3(threads), args, dyn, ctx->stream) == hipSuccess ? KK_OK : KK_ERR_HIP; }
void kk_prof_begin(kk_ctx, const char*) {}
void kk_prof_end(kk_ctx) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_fill(double* x, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        x[i] = ((double)(z >> 11) / 9007199254740992.0 - 0.5) * 1e-3;
    }
}
int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 2000000;
    const int m = argc > 2 ? atoi(argv[2]) : 32;
    const int width = argc > 3 ? atoi(argv[3]) : 0;
    kk_ctx_s c;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    c.num_cus = p.multiProcessorCount; c.device = 0;
    CK(hipStreamCreate(&c.stream));
    CK(hipMalloc(&c.ws, WS_TOTAL * 8)); CK(hipMemset(c.ws, 0, WS_TOTAL * 8));
    CK(hipMalloc(&c.d_sync, KK_SYNC_BYTES)); CK(hipMemset(c.d_sync, 0, KK_SYNC_BYTES));
    c.panel_width = width;
    long long ld = (n + 511) / 512 * 512; if (((ld / 512) & 1) == 0) ld += 512;
    double* V; CK(hipMalloc(&V, (size_t)ld * (m + 1) * 8));
    k_fill<<<2048, 256>>>(V, (size_t)ld * (m + 1), 7);
    long long* tr; const int np_max = m + 2;
    CK(hipMalloc(&tr, np_max * 16 * 8)); CK(hipMemset(tr, 0, np_max * 16 * 8));
#ifdef KK_PANEL_TRACE
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_panel_trace), &tr, sizeof(tr)));
#endif
    CK(hipDeviceSynchronize());
    double* w = V + (size_t)ld * m;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, c.stream));
        int st = kk_launch_mgs_panel(&c, V, ld, m, 1, w, nullptr, nullptr, c.ws + WS_S, KK_MAX_M, c.ws + WS_SCAL + SC_NRM2, false, false);
        CK(hipEventRecord(e1, c.stream)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (st != KK_OK) { printf("launch failed %d\n", st); return 1; }
        if (rep < 2) continue;
        std::vector<long long> h(np_max * 16);
        CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        const int P = kk_mgs_panel_width(&c, ld, false), npanels = (m + P - 1) / P;
        printf("{\"rows\": %lld, \"m\": %d, \"P\": %d, \"kernel_us\": %.2f, \"us_per_vector\": %.3f, \"stream_us_per_vector_at_8TBps\": %.3f}\n", n, m, P, ms * 1e3, ms * 1e3 / m, n * 8.0 / 8e12 * 1e6);
#ifndef KK_PANEL_TRACE
        continue;
#endif
        printf("# panel: request_next  partials(+us)  barrier1  totals(barrier2)  update_done | reduction wave: published  totals  passes | period\n");
        const long long t00 = h[0];
        long long prev = t00;
        for (int q = 0; q < npanels; ++q) {
            const long long* r = &h[q * 16];
            auto us = [&](long long t) { return (t - t00) * 0.01; };
            printf("%3d: %8.2f  %8.2f  %8.2f  %8.2f  %8.2f | %8.2f  %8.2f  %lld | %6.2f\n", q, us(r[0]), us(r[1]), us(r[2]), us(r[3]), us(r[4]), us(r[8]), us(r[9]), r[10],
                   (r[0] - prev) * 0.01);
            prev = r[0];
        }
    }
    return 0;
}
