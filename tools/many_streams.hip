// What does the memory system give a kernel that reads MANY column streams at once?  The two big kernels of the block step
// (k_block_gram2p, k_block_update*) read kn = 32 .. 112 basis columns of 10 M rows side by side: a block walks its row range in
// chunks of 512 rows and, per chunk, touches every column (4 KB each, 80 MB apart).  Their MARGINAL cost per 16 more columns is
// 250-270 us = 4.9-5.1 TB/s, against 7.1 TB/s for one contiguous stream (tools/hbm_peak.hip).  This probe reads ncol columns of nrow
// doubles (column stride ld) with exactly that loop shape -- 1024 blocks x 256 threads, 16 bytes per lane and load, F loads in flight --
// and reports TB/s for ncol = 1 (one long column of the same total size) .. 112, for two chunk shapes:
//   chunk 512  rows: per iteration a block reads 4 KB of every column                (the block kernels' shape)
//   chunk 2048 rows: per iteration a block reads 16 KB of every column (4 loads per lane and column back to back)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/many_streams tools/many_streams.hip ; run: tools/bin/many_streams [random]
// (`random`: the buffer holds pseudo-random doubles instead of zeros -- the rate must not depend on the data)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CH /* 16-byte loads per lane, column and iteration: chunk = 512 * CH rows */, int F /* columns in flight */>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ V, long ld, int ncol, long nrow, long rpb, double* __restrict__ sink) {
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < nrow ? r0 + rpb : nrow;
    d2 acc = {0.0, 0.0};
    for (long r = r0 + threadIdx.x * 2; r < r1; r += 512 * CH) {
        for (int c = 0; c < ncol; c += F) {
            d2 x[F][CH];
#pragma unroll
            for (int u = 0; u < F; ++u)
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const long rr = r + (long)k * 512;
                    x[u][k] = (c + u < ncol && rr < r1) ? __builtin_nontemporal_load((const d2*)(V + (long)(c + u) * ld + rr)) : d2{0.0, 0.0};
                }
#pragma unroll
            for (int u = 0; u < F; ++u)
#pragma unroll
                for (int k = 0; k < CH; ++k) acc += x[u][k];
        }
    }
    if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

__global__ void k_fill(double* V, size_t n) {   // pseudo-random doubles in (-1, 1): a zero-filled buffer toggles no data lines
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        V[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    }
}
template <int CH, int F>
static void run(const double* V, long ld, int ncol, long nrow, const char* what) {
    double* sink; CK(hipMalloc(&sink, 8));
    const int nblk = 1024;
    long rpb = (nrow + nblk - 1) / nblk; rpb = (rpb + 512 * CH - 1) / (512 * CH) * (512 * CH);
    const int grid = (int)((nrow + rpb - 1) / rpb);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_read<CH, F>), dim3(grid), dim3(256), 0, 0, V, ld, ncol, nrow, rpb, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("{\"shape\": \"%s\", \"columns\": %d, \"rows\": %ld, \"chunk_rows\": %d, \"columns_in_flight\": %d, \"GB\": %.2f, \"ms\": %.3f, \"TBps\": %.2f}\n", what, ncol, nrow,
           512 * CH, F, (double)ncol * nrow * 8 / 1e9, best, (double)ncol * nrow * 8 / (best * 1e-3) / 1e12);
    CK(hipFree(sink));
}
int main(int argc, char** argv) {
    const long N = 10000896, ld = N;
    const int maxcol = 112;
    double* V; CK(hipMalloc(&V, (size_t)maxcol * ld * 8));
    const bool random = argc > 1 && argv[1][0] == 'r';
    if (random) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, V, (size_t)maxcol * ld); CK(hipDeviceSynchronize()); }
    else CK(hipMemset(V, 0, (size_t)maxcol * ld * 8));
    printf("{\"data\": \"%s\"}\n", random ? "pseudo-random doubles" : "zeros");
    // one long column of the same bytes as 112 columns
    run<1, 4>(V, ld, 1, (long)maxcol * N, "one contiguous column");
    run<4, 1>(V, ld, 1, (long)maxcol * N, "one contiguous column");
    const int cols[] = {16, 64, 112};
    for (int nc : cols) {
        run<1, 4>(V, ld, nc, N, "columns side by side");
        run<1, 8>(V, ld, nc, N, "columns side by side");
        run<4, 1>(V, ld, nc, N, "columns side by side");
        run<4, 2>(V, ld, nc, N, "columns side by side");
    }
    return 0;
}
