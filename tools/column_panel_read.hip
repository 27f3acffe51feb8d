// What does the MEMORY SYSTEM deliver for the access patterns a Gram-panel kernel can choose from?  (exploration tool, round 4)
// A column-major panel of C columns x N rows (leading dimension ld) is read once; three lane -> address maps:
//   A  "mfma": lane (c = l & 15, kq = l >> 4) reads 16 B of column c at row 2 kq of an 8-row group: one instruction = 16 columns x 64 B
//      (the operand layout of v_mfma_f64_16x16x4: what k_block_gram2p does);
//   B  "line": lane (c = l & 7, h = l >> 3) reads 16 B of column c at row 2 h of a 16-row group: one instruction = 8 columns x 128 B
//      (whole cache lines; an MFMA kernel would have to permute registers or go through LDS);
//   C  "row":  a wave instruction reads 1 KB of ONE column (what k_project / k_block_update do; LDS transposition for MFMA).
// Every wave keeps DEPTH chunks of 16 loads in flight; sums are kept so that nothing is optimised away.
// build: hipcc --offload-arch=gfx950 -O3 tools/column_panel_read.hip -o tools/bin/column_panel_read ; run: tools/bin/column_panel_read [C] [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ d2 bl(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    d2 o;
    o.x = __longlong_as_double((long long)(((unsigned long long)t.y << 32) | t.x));
    o.y = __longlong_as_double((long long)(((unsigned long long)t.w << 32) | t.z));
    return o;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const double* p, long long bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// PAT 0 = A (16 cols x 64 B), 1 = B (8 cols x 128 B), 2 = C (1 col x 1 KB).  A "unit" = 16 loads of one wave:
//   A: one 16-column group, 32 rows (4 loads per lane);  B: one 16-column group, 32 rows (2 x 2 loads: 8 columns each, 32 rows = 2 x 16);
//   C: 16 columns x 128 rows (one load per column).  Bytes per unit: A, B: 4 KB; C: 16 KB.
template <int PAT, int AUX, int DEPTH>
__global__ __launch_bounds__(256) void k_panel(const double* __restrict__ X, long long ld, int G, long long rpb, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * rpb;
    const int rows_unit = PAT == 2 ? 128 : 32;
    const int nunit = (int)(rpb / (4 * rows_unit));         // units per wave along the rows (per group)
    double acc = 0.0;
    for (int g = 0; g < G; ++g) {
        const __amdgpu_buffer_rsrc_t R = rsrc(X + (long long)g * 16 * ld, (15 * ld + ld) * 8);
        unsigned vo[16];
        if (PAT == 0) { const int c = lane & 15, kq = lane >> 4; for (int t = 0; t < 4; ++t) vo[t] = (unsigned)(c * ld * 8) + kq * 16 + t * 64; }
        if (PAT == 1) { const int c = lane & 7, h = lane >> 3; for (int t = 0; t < 4; ++t) vo[t] = (unsigned)((c + 8 * (t & 1)) * ld * 8) + h * 16 + (t >> 1) * 128; }
        if (PAT == 2) { for (int t = 0; t < 16; ++t) vo[t] = (unsigned)(t * ld * 8) + lane * 16; }
        constexpr int NL = PAT == 2 ? 16 : 4;
        d2 buf[DEPTH][NL];
        unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((r0 + (long long)wave * rows_unit) * 8));
        const unsigned step = 4 * rows_unit * 8;
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {
#pragma unroll
            for (int t = 0; t < NL; ++t) buf[d][t] = bl<AUX>(R, vo[t], so + d * step);
        }
        for (int u = 0; u < nunit; u += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int nx = (d + DEPTH - 1) % DEPTH;
#pragma unroll
                for (int t = 0; t < NL; ++t) buf[nx][t] = bl<AUX>(R, vo[t], so + (unsigned)(u + d + DEPTH - 1) * step);   // past the end: zeros (descriptor bound)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int t = 0; t < NL; ++t) acc += buf[d][t].x + buf[d][t].y;
                asm volatile("" ::: "memory");
            }
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

template <int PAT, int AUX, int DEPTH>
void run(const double* X, long long ld, int C, int bpc, double* out, const char* name) {
    const int ncu = 256;
    const long long nsub = ld / 512;
    const long long maxb = (long long)bpc * ncu;
    const long long spb = (nsub + maxb - 1) / maxb;
    const long long rpb = spb * 512;
    const int nblk = (int)((nsub + spb - 1) / spb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_panel<PAT, AUX, DEPTH><<<nblk, 256>>>(X, ld, C / 16, rpb, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(a);
        k_panel<PAT, AUX, DEPTH><<<nblk, 256>>>(X, ld, C / 16, rpb, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("{\"pattern\": \"%s\", \"aux\": %d, \"chunks_in_flight\": %d, \"blocks_per_cu\": %d, \"ms\": %.3f, \"TBps\": %.2f}\n", name, AUX, DEPTH, bpc, best,
           (double)C * ld * 8.0 / best / 1e9);
    fflush(stdout);
}
int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 64;
    const long long ld = argc > 2 ? atoll(argv[2]) : 10000896;
    double *X, *out;
    hipMalloc(&X, (size_t)C * ld * 8); hipMalloc(&out, 8);
    hipMemset(X, 0, (size_t)C * ld * 8);
    for (int bpc : {2, 4, 8}) {
        run<0, 0, 2>(X, ld, C, bpc, out, "A mfma 16x64B");
        run<0, 0, 4>(X, ld, C, bpc, out, "A mfma 16x64B");
        run<0, 2, 4>(X, ld, C, bpc, out, "A mfma 16x64B");
        run<1, 0, 2>(X, ld, C, bpc, out, "B line 8x128B");
        run<1, 0, 4>(X, ld, C, bpc, out, "B line 8x128B");
        run<1, 2, 4>(X, ld, C, bpc, out, "B line 8x128B");
        run<2, 0, 2>(X, ld, C, bpc, out, "C row 1x1KB");
        run<2, 2, 2>(X, ld, C, bpc, out, "C row 1x1KB");
    }
    return 0;
}
