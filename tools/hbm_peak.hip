// Empirical HBM read ceiling of the box (exploration tool): grid-strided 16-byte loads, U loads in flight per lane.
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_peak.hip -o tools/bin/hbm_peak ; run: tools/bin/hbm_peak [GiB]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ x, size_t n16, double* out) {
    // block-contiguous: each block owns a contiguous range, steps of 256*U elements
    size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    per = (per + 256 * U - 1) / (256 * U) * (256 * U);
    size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
    double acc = 0;
    for (size_t i = b0 + threadIdx.x; i + (size_t)(U - 1) * 256 < b1; i += (size_t)256 * U) {
        d2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(x + i + (size_t)k * 256) : x[i + (size_t)k * 256];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y;
    }
    if (acc == 12345.678) out[0] = acc;
}
template <int U, bool NT>
void run(const d2* x, size_t n16, double* out, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_read<U, NT><<<grid, 256>>>(x, n16, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        k_read<U, NT><<<grid, 256>>>(x, n16, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("U=%2d nt=%d grid=%5d  %.3f ms  %.1f GB/s\n", U, (int)NT, grid, best, n16 * 16.0 / best / 1e6);
}
// --json: the quick form bench.py runs at its start (~0.3 s of GPU time): best of the non-temporal and of the plain 16-byte
// read streams over a 2 GiB buffer, one JSON line -- the box's own attainable read ceiling next to the 8 TB/s of the data sheet
template <int U, bool NT>
static double best_gbps(const d2* x, size_t n16, double* out, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_read<U, NT><<<grid, 256>>>(x, n16, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        k_read<U, NT><<<grid, 256>>>(x, n16, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return n16 * 16.0 / best / 1e6;
}
static int json_mode() {
    const size_t n16 = (size_t)(2.0 * (1ull << 30)) / 16;
    d2* x; double* out;
    if (hipMalloc(&x, n16 * 16) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    hipMemset(x, 0, n16 * 16);
    double nt = 0, plain = 0;
    for (int grid : {2048, 4096}) {
        nt = fmax(nt, best_gbps<16, true>(x, n16, out, grid));
        nt = fmax(nt, best_gbps<8, true>(x, n16, out, grid));
        plain = fmax(plain, best_gbps<8, false>(x, n16, out, grid));
        plain = fmax(plain, best_gbps<16, false>(x, n16, out, grid));
    }
    printf("{\"nt_read_GBps\": %.1f, \"plain_read_GBps\": %.1f, \"buffer_GiB\": 2.0, \"tool\": \"tools/hbm_peak.hip --json\"}\n", nt, plain);
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--json")) return json_mode();
    double gib = argc > 1 ? atof(argv[1]) : 8.0;
    size_t n16 = (size_t)(gib * (1ull << 30)) / 16;
    d2* x; double* out;
    hipMalloc(&x, n16 * 16); hipMalloc(&out, 8);
    hipMemset(x, 0, n16 * 16);
    for (int grid : {512, 1024, 2048, 4096, 16384}) {
        run<4, false>(x, n16, out, grid);
        run<8, false>(x, n16, out, grid);
        run<16, false>(x, n16, out, grid);
        run<8, true>(x, n16, out, grid);
        run<16, true>(x, n16, out, grid);
    }
    return 0;
}
