// Where does the block update (W <- W - V S: kn basis columns streamed, 16 residual columns read-modify-written; k_block_update_lds /
// k_block_update_commit, csrc/kk_kernels_block.hip) lose the 17-28 % between its marginal 4.9-5.7 TB/s per basis column and the 6.85 TB/s
// the same access pattern reaches with nothing but loads (tools/many_streams.hip)?  One loop shape, the suspects switched one by one:
//   mode 0  the kernel's inner loop: 16 B of a basis column per lane and load, 4 loads in flight + 4 prefetched, per column 8 broadcast
//           ds_read_b128 of the coefficient row and 32 FMAs
//   mode 1  the same loads and stores, ONE add per load (no coefficients, no FMAs): the access pattern alone, RMW included
//   mode 2  32 FMAs per load, coefficients from registers (no LDS traffic)
//   mode 3  8 LDS reads per load, 4 FMAs per load (LDS traffic without the arithmetic)
//   RP = 2  two row pairs per lane (rows r and r + 512): every coefficient read serves 64 FMAs, 2 blocks per CU instead of 4
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/update_probe tools/update_probe.hip ; run: tools/bin/update_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define NB 16
__device__ __forceinline__ d2 ldnt(const double* p) { return __builtin_nontemporal_load((const d2*)p); }
__device__ __forceinline__ void stnt(double* p, d2 v) { __builtin_nontemporal_store(v, (d2*)p); }

template <int MODE, int RP>
__global__ __launch_bounds__(256, (RP == 1 ? 4 : 2)) void k_upd(const double* __restrict__ V, long ld, int m, double* __restrict__ W, const double* __restrict__ S, long rpb, long nrow) {
    extern __shared__ __attribute__((aligned(16))) double ssm[];
    const int tid = threadIdx.x;
    for (int e = tid; e < m * NB; e += 256) ssm[e] = S[e];
    __syncthreads();
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < nrow ? r0 + rpb : nrow;
    for (long r = r0 + tid * 2; r < r1; r += 512 * RP) {
        d2 acc[RP][NB];
#pragma unroll
        for (int p = 0; p < RP; ++p)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[p][j] = ldnt(W + (long)j * ld + r + p * 512);
        d2 xn[RP][4];
#pragma unroll
        for (int p = 0; p < RP; ++p)
#pragma unroll
            for (int u = 0; u < 4; ++u) xn[p][u] = ldnt(V + (long)u * ld + r + p * 512);
        for (int c = 0; c + 4 <= m; c += 4) {
            d2 x[RP][4];
#pragma unroll
            for (int p = 0; p < RP; ++p)
#pragma unroll
                for (int u = 0; u < 4; ++u) x[p][u] = xn[p][u];
            if (c + 8 <= m) {
#pragma unroll
                for (int p = 0; p < RP; ++p)
#pragma unroll
                    for (int u = 0; u < 4; ++u) xn[p][u] = ldnt(V + (long)(c + 4 + u) * ld + r + p * 512);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 1) {
#pragma unroll
                    for (int p = 0; p < RP; ++p) acc[p][(u * 4) & 15] += x[p][u];
                } else if (MODE == 2) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int p = 0; p < RP; ++p) { acc[p][j].x = fma(1e-3 * (j + 1), x[p][u].x, acc[p][j].x); acc[p][j].y = fma(1e-3 * (j + 1), x[p][u].y, acc[p][j].y); }
                } else {
                    const d2* Sc = reinterpret_cast<const d2*>(ssm + (size_t)(c + u) * NB);
#pragma unroll
                    for (int j2 = 0; j2 < NB / 2; ++j2) {
                        const d2 sv = Sc[j2];
#pragma unroll
                        for (int p = 0; p < RP; ++p) {
                            if (MODE == 0) {
                                acc[p][2 * j2].x = fma(sv.x, x[p][u].x, acc[p][2 * j2].x); acc[p][2 * j2].y = fma(sv.x, x[p][u].y, acc[p][2 * j2].y);
                                acc[p][2 * j2 + 1].x = fma(sv.y, x[p][u].x, acc[p][2 * j2 + 1].x); acc[p][2 * j2 + 1].y = fma(sv.y, x[p][u].y, acc[p][2 * j2 + 1].y);
                            } else {   // MODE 3: the LDS reads stay live through one FMA each
                                acc[p][2 * j2].x = fma(sv.x + sv.y, x[p][u].x, acc[p][2 * j2].x);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < RP; ++p)
#pragma unroll
            for (int j = 0; j < NB; ++j) stnt(W + (long)j * ld + r + p * 512, acc[p][j]);
    }
}

template <int MODE, int RP>
static void run(const double* V, long ld, int m, double* W, const double* S, long nrow, const char* what) {
    const int nblk = 1024;
    long rpb = (nrow + nblk - 1) / nblk; rpb = (rpb + 1023) / 1024 * 1024;
    const int grid = (int)((nrow + rpb - 1) / rpb);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_upd<MODE, RP>), dim3(grid), dim3(256), (size_t)m * NB * 8, 0, V, ld, m, W, S, rpb, nrow);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double gb = ((double)m * 8 + 256) * nrow / 1e9;
    printf("{\"mode\": \"%s\", \"row_pairs_per_lane\": %d, \"basis_columns\": %d, \"GB\": %.2f, \"ms\": %.3f, \"TBps\": %.2f}\n", what, RP, m, gb, best, gb / best);
    fflush(stdout);
}
__global__ void k_fill(double* V, size_t n, double scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        V[i] = ((double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0) * scale;
    }
}
int main() {
    const long N = 10000896 / 1024 * 1024, ld = 10000896;
    const int maxcol = 112;
    double *V, *W, *S;
    CK(hipMalloc(&V, (size_t)maxcol * ld * 8)); CK(hipMalloc(&W, (size_t)NB * ld * 8)); CK(hipMalloc(&S, (size_t)maxcol * NB * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, V, (size_t)maxcol * ld, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, W, (size_t)NB * ld, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, S, (size_t)maxcol * NB, 1e-4);
    CK(hipDeviceSynchronize());
    const int ms[] = {48, 112};
    for (int m : ms) {
        run<0, 1>(V, ld, m, W, S, N, "0: LDS coefficients + 32 FMAs per load (the kernel)");
        run<1, 1>(V, ld, m, W, S, N, "1: loads and stores only");
        run<2, 1>(V, ld, m, W, S, N, "2: 32 FMAs per load, no LDS");
        run<3, 1>(V, ld, m, W, S, N, "3: 8 LDS reads per load, 4 FMAs");
        run<0, 2>(V, ld, m, W, S, N, "0: LDS coefficients + 32 FMAs per load (the kernel)");
        run<1, 2>(V, ld, m, W, S, N, "1: loads and stores only");
        run<2, 2>(V, ld, m, W, S, N, "2: 32 FMAs per load, no LDS");
    }
    return 0;
}
