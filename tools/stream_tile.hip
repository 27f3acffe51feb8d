// Exploration tool: how the per-stream granularity of a many-column read affects the achieved bandwidth on MI355X.
// M basis columns (ld rows each, column-major) are read once by 256-thread blocks that own contiguous row ranges:
//   A<CB>      : for every 512-row chunk, loop over the columns CB at a time (1 KiB per wave per column visit: the block-update
//                / Gram access shape), CB 16-byte loads in flight per lane
//   B<RG, CB>  : for every group of RG chunks, loop over the columns CB at a time (RG KiB per wave per column visit: the
//                project / unproject access shape), RG*CB loads in flight
//   F = extra f64 FMAs per loaded double (emulates the arithmetic of a 16-column update at F = 16)
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_tile.hip -o tools/bin/stream_tile ; run: tools/bin/stream_tile [M] [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ldnt(const double* p) { return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p)); }

template <int RG, int CB, int F>
__global__ __launch_bounds__(256) void k_tile(const double* __restrict__ V, long ld, int m, long rpb, double* out) {
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < ld ? r0 + rpb : ld;
    double acc[F > 0 ? F : 1];
#pragma unroll
    for (int f = 0; f < (F > 0 ? F : 1); ++f) acc[f] = 0;
    for (long rb = r0; rb + (long)RG * 512 <= r1; rb += (long)RG * 512) {
        const long r = rb + threadIdx.x * 2;
        for (int c = 0; c + CB <= m; c += CB) {
            d2 v[RG][CB];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int g = 0; g < RG; ++g) v[g][u] = ldnt(V + (long)(c + u) * ld + r + (long)g * 512);
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    if (F == 0) acc[0] += v[g][u].x + v[g][u].y;
                    else {
#pragma unroll
                        for (int f = 0; f < F; ++f) { acc[f] = fma(v[g][u].x, 1.0000001, acc[f]); acc[f] = fma(v[g][u].y, 0.9999999, acc[f]); }
                    }
                }
        }
    }
    double t = 0;
#pragma unroll
    for (int f = 0; f < (F > 0 ? F : 1); ++f) t += acc[f];
    if (t == 12345.678) out[0] = t;
}
template <int RG, int CB, int F>
void run(const double* V, long ld, int m, double* out, int bpc) {
    const long nsub = ld / 512, target = 256L * bpc;
    long spb = (nsub + target - 1) / target;
    spb = (spb + RG - 1) / RG * RG;
    const int nblk = (int)((nsub + spb - 1) / spb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_tile<RG, CB, F><<<nblk, 256>>>(V, ld, m, spb * 512, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        k_tile<RG, CB, F><<<nblk, 256>>>(V, ld, m, spb * 512, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("rows/visit=%5d cols/batch=%2d fma/double=%2d blocks/CU=%d  %.3f ms  %.2f TB/s\n", RG * 512, CB, F, bpc, best,
           (double)m * ld * 8 / best / 1e9);
}
// ---- dissection of the 16-column block update: which ingredient costs the bandwidth?
//   STORE: write the 16 output columns;  COEF: 0 = compile-time constants, 1 = LDS broadcast reads, 2 = scalar loads
template <int NB, int STORE, int COEF, int PFB>
__global__ __launch_bounds__(256, 4) void k_upd(const double* V, long ld, int m, double* W, long ldw, const double* __restrict__ S,
                                                long rpb, double* out) {
    extern __shared__ __attribute__((aligned(16))) double ssm[];
    const int tid = threadIdx.x;
    if (COEF == 1) { for (int e = tid; e < m * NB; e += 256) ssm[e] = S[e]; __syncthreads(); }
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < ld ? r0 + rpb : ld;
    double chk = 0;
    for (long r = r0 + tid * 2; r < r1; r += 512) {
        d2 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = d2{0.0, 0.0};
        int c = 0;
        d2 xn[PFB];
#pragma unroll
        for (int u = 0; u < PFB; ++u) if (u < m) xn[u] = ldnt(V + (long)u * ld + r);
        for (; c + PFB <= m; c += PFB) {
            d2 x[PFB];
#pragma unroll
            for (int u = 0; u < PFB; ++u) x[u] = xn[u];
            if (c + 2 * PFB <= m) {
#pragma unroll
                for (int u = 0; u < PFB; ++u) xn[u] = ldnt(V + (long)(c + PFB + u) * ld + r);
            }
#pragma unroll
            for (int u = 0; u < PFB; ++u) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double sv;
                    if (COEF == 0) sv = 1.0 + 1e-7 * j;
                    else if (COEF == 1) sv = ssm[(c + u) * NB + j];
                    else sv = S[(long)(c + u) * NB + j];
                    acc[j].x = fma(sv, x[u].x, acc[j].x); acc[j].y = fma(sv, x[u].y, acc[j].y);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (STORE == 1) *reinterpret_cast<d2*>(W + (long)j * ldw + r) = acc[j];
            else if (STORE == 2) __builtin_nontemporal_store(acc[j], reinterpret_cast<d2*>(W + (long)j * ldw + r));
            else chk += acc[j].x + acc[j].y;
        }
    }
    if (chk == 12345.678) out[0] = chk;
}
template <int STORE, int COEF, int PFB>
void run_upd(const double* V, long ld, int m, double* W, const double* S, double* out) {
    const long nsub = ld / 512, target = 1024;
    const long spb = (nsub + target - 1) / target;
    const int nblk = (int)((nsub + spb - 1) / spb);
    const size_t shm = COEF == 1 ? (size_t)m * 16 * 8 : 0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_upd<16, STORE, COEF, PFB><<<nblk, 256, shm>>>(V, ld, m, W, ld, S, spb * 512, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        k_upd<16, STORE, COEF, PFB><<<nblk, 256, shm>>>(V, ld, m, W, ld, S, spb * 512, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("update m=%3d store=%s coef=%s prefetch=%d  %.3f ms  %.2f TB/s (reads + writes)\n", m, STORE == 0 ? "none" : (STORE == 1 ? "plain" : "nt"),
           COEF == 0 ? "const" : (COEF == 1 ? "lds" : "scalar"), PFB, best, ((double)m + (STORE ? 16 : 0)) * ld * 8 / best / 1e9);
}

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 112;
    long ld = argc > 2 ? atol(argv[2]) : 10000896L;
    double *V, *out;
    hipMalloc(&V, (size_t)m * ld * 8); hipMalloc(&out, 8);
    hipMemset(V, 0, (size_t)m * ld * 8);
    printf("M = %d columns of %ld rows (%.1f GB)\n", m, ld, m * ld * 8e-9);
    for (int bpc : {4}) {
        run<1, 4, 0>(V, ld, m, out, bpc);
        run<1, 8, 0>(V, ld, m, out, bpc);
        run<1, 16, 0>(V, ld, m, out, bpc);
        run<2, 4, 0>(V, ld, m, out, bpc);
        run<4, 4, 0>(V, ld, m, out, bpc);
        run<8, 2, 0>(V, ld, m, out, bpc);
        run<16, 2, 0>(V, ld, m, out, bpc);
        run<16, 1, 0>(V, ld, m, out, bpc);
    }
    {
        double *W, *S;
        hipMalloc(&W, (size_t)16 * ld * 8); hipMalloc(&S, (size_t)m * 16 * 8);
        hipMemset(W, 0, (size_t)16 * ld * 8); hipMemset(S, 0, (size_t)m * 16 * 8);
        for (int mm : {16, 32, 112}) {
            run_upd<0, 1, 4>(V, ld, mm, W, S, out);
            run_upd<1, 1, 4>(V, ld, mm, W, S, out);
            run_upd<2, 1, 4>(V, ld, mm, W, S, out);
        }
        // pure write of 16 columns (m = 0 reads): the write ceiling
        run_upd<1, 0, 4>(V, ld, 0, W, S, out);
        run_upd<2, 0, 4>(V, ld, 0, W, S, out);
    }
    run<1, 4, 16>(V, ld, m, out, 4);
    run<1, 8, 16>(V, ld, m, out, 4);
    run<2, 4, 16>(V, ld, m, out, 4);
    run<1, 4, 4>(V, ld, m, out, 4);
    run<1, 4, 8>(V, ld, m, out, 4);
    return 0;
}
