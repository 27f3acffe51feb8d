// gfx950 block (multi-vector) kernels of the BlockLanczos path: MFMA f64 Gram panels and the multi-right-hand-side update.
#include "kk_device.h"

// ------------------------------------------------------------------------------------------
// Block (multi-vector) kernels for BlockLanczos (src/factorizations/blocklanczos.jl)
// ------------------------------------------------------------------------------------------
// block_inner / the tall-skinny panel  C = X' Y  (p x q, q <= 16 per launch): the one place where
// the path is a genuine dense contraction, done on v_mfma_f64_16x16x4_f64.
//   D[i][j] += sum_k A[i][k] B[k][j],  i = X column (16 per group), j = Y column, k = 4 rows.
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][j = l&15];
//   C/D (f64 map): lane l, reg r -> row i = (l>>4) + 4r, col j = l&15.
// Lane (c = l&15, kq = l>>4) streams BG_T rows of column c of a 32-row chunk with 16 B loads, the four lanes of a
// column covering 64 contiguous bytes per load instruction; MFMA t uses element t of every lane: any row->k-slot
// map is valid as long as A and B use the same one.  X is read exactly once, Y once per launch.
#define BG_T 8                       // rows per lane per chunk (4 x dwordx4)
#define BG_CHUNK (4 * BG_T)          // rows per wave chunk

template <int NG>  // NG groups of 16 X-columns
__global__ __launch_bounds__(KK_TPB) void k_block_gram(const double* __restrict__ X, int64_t ldx, int p,
                                                       const double* __restrict__ Y, int64_t ldy, int q, int64_t ld,
                                                       int64_t rpb, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [NG][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    v4d acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = v4d{0.0, 0.0, 0.0, 0.0};
    const bool yok = c < q;
    for (int64_t rc = r0 + wave * BG_CHUNK; rc < r1; rc += 4 * BG_CHUNK) {
        // rows of lane (c, kq): {rc + 4t + 2kq, +1 : t = 0,2,4,6}  -- the four lanes of one column read 64
        // contiguous bytes per load instruction (same row -> k-slot map for X and Y, so the contraction is unchanged)
        const int64_t row = rc + kq * 2;
        double yv[BG_T];
        if (yok) {
            const double* yp = Y + (int64_t)c * ldy + row;
#pragma unroll
            for (int t = 0; t < BG_T; t += 2) { d2 v = ld2(yp + 4 * t); yv[t] = v.x; yv[t + 1] = v.y; }
        } else {
#pragma unroll
            for (int t = 0; t < BG_T; ++t) yv[t] = 0.0;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int col = g * 16 + c;
            double xv[BG_T];
            if (col < p) {
                const double* xp = X + (int64_t)col * ldx + row;
#pragma unroll
                for (int t = 0; t < BG_T; t += 2) { d2 v = ld2(xp + 4 * t); xv[t] = v.x; xv[t + 1] = v.y; }  // plain: X == Y panels re-hit L2
            } else {
#pragma unroll
                for (int t = 0; t < BG_T; ++t) xv[t] = 0.0;
            }
#pragma unroll
            for (int t = 0; t < BG_T; ++t) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[t], yv[t], acc[g], 0, 0, 0);
        }
    }
    // combine the 4 waves through LDS in a fixed order, then one coalesced partial tile per block
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* a = &lds[(g * 4 + r) * 64 + lane];
                    *a = (w == 0) ? acc[g][r] : (*a + acc[g][r]);
                }
        }
        __syncthreads();
    }
    double* dst = part + (int64_t)blockIdx.x * (NG * 256);
    for (int e = tid; e < NG * 256; e += KK_TPB) dst[e] = lds[e];
}

// C[i + ldc*j] = sum_b part[b][e(i,j)]   (one thread per output entry)
__global__ __launch_bounds__(KK_TPB) void k_finalize_gram(const double* __restrict__ part, int nblk, int ng, int p, int q,
                                                          double* __restrict__ C, int ldc) {
    const int idx = blockIdx.x * KK_TPB + threadIdx.x;
    if (idx >= p * q) return;
    const int i = idx % p, j = idx / p;
    const int g = i >> 4, ii = i & 15;          // ii = (lane>>4) + 4 r  ->  r = ii>>2, lane>>4 = ii&3
    const int r = ii >> 2, lane = ((ii & 3) << 4) | j;
    const int e = (g * 4 + r) * 64 + lane;
    const int64_t stride = (int64_t)ng * 256;
    double a = 0;
    for (int b = 0; b < nblk; ++b) a += part[(int64_t)b * stride + e];
    C[i + (int64_t)ldc * j] = a;
}

// W[:, j] = beta*W[:, j] + alpha * sum_c V[:, c] S[c*NB + j]   for j < nb <= NB, c < m   (S rows padded to NB)
// (three-term block update, block_reorthogonalize! panel update, CholQR back-substitution).
// S lives in device memory (scalar loads); fused column norms |W_j|^2 -> partials.
template <int NB, bool BZERO>
__global__ __launch_bounds__(KK_TPB) void k_block_update(const double* V, int64_t ld, int m, const double* Win,
                                                         double* Wout, int64_t ldw_in, int64_t ldw_out, int nb,
                                                         const double* __restrict__ S, double alpha, double beta,
                                                         int64_t rpb, double* __restrict__ part_nrm) {
    // per-thread column-norm accumulators live in LDS (slot [j][tid], touched by its owner only): the 2*NB VGPRs
    // they would cost are what keeps the NB=16 instantiation at 4 waves/SIMD with the load pipeline below
    __shared__ double nsm[NB * KK_TPB];
    __shared__ double sm[4];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    if (part_nrm) {
#pragma unroll
        for (int j = 0; j < NB; ++j) nsm[j * KK_TPB + tid] = 0.0;
    }
    for (int64_t r = r0 + tid * 2; r < r1; r += KK_SUB) {
        d2 acc[NB];          // acc_j = sum_c S[c][j] V_c   (S straight from scalar registers into the FMA)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = d2{0.0, 0.0};
        int c = 0;
        d2 xn[4];            // software pipeline: the loads of batch c+4 are in flight while batch c is multiplied
        if (m >= 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) xn[u] = ld2s(V + (int64_t)u * ld + r);
        }
        for (; c + 4 <= m; c += 4) {
            d2 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = xn[u];
            if (c + 8 <= m) {
#pragma unroll
                for (int u = 0; u < 4; ++u) xn[u] = ld2s(V + (int64_t)(c + 4 + u) * ld + r);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double* Sc = S + (int64_t)(c + u) * NB;   // rows padded to NB by the caller (zeros beyond nb)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const double sv = Sc[j];
                    acc[j].x = fma(sv, x[u].x, acc[j].x); acc[j].y = fma(sv, x[u].y, acc[j].y);
                }
            }
        }
        for (; c < m; ++c) {
            const d2 x = ld2(V + (int64_t)c * ld + r);
            const double* Sc = S + (int64_t)c * NB;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const double sv = Sc[j];
                acc[j].x = fma(sv, x.x, acc[j].x); acc[j].y = fma(sv, x.y, acc[j].y);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < nb) {
                d2 w{alpha * acc[j].x, alpha * acc[j].y};
                if (!BZERO) {
                    const d2 wi = ld2(Win + (int64_t)j * ldw_in + r);
                    w.x = fma(beta, wi.x, w.x); w.y = fma(beta, wi.y, w.y);
                }
                st2(Wout + (int64_t)j * ldw_out + r, w);
                if (part_nrm) nsm[j * KK_TPB + tid] += fma(w.x, w.x, w.y * w.y);
            }
        }
    }
    if (part_nrm) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < nb) {
                double t = block_sum(nsm[j * KK_TPB + tid], sm);
                if (tid == 0) part_nrm[(int64_t)j * KK_MAX_BLOCKS + blockIdx.x] = t;
            }
        }
    }
}

// ---- launchers
// ---- block launchers ---------------------------------------------------------------------
// C (p x q, column-major ldc) = X' Y on device memory `C_dev`; p <= 128, q <= 16 per call
int kk_launch_block_gram(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
                         double* C_dev, int ldc) {
    if (p <= 0 || q <= 0) return KK_OK;
    if (p > 128 || q > 16) { kk_set_error("kk_launch_block_gram: p=%d q=%d exceed one launch (128 x 16)", p, q); return KK_ERR_INVALID; }
    kk_part pt = kk_partition(ctx, ld);
    // cap the grid: every block leaves an NG*256-double partial tile
    int nblk = pt.nblk;
    int64_t rpb = pt.rpb;
    const int maxb = 2 * ctx->num_cus;
    if (nblk > maxb) {
        const int64_t nsub = ld / KK_SUB;
        const int64_t spb = (nsub + maxb - 1) / maxb;
        rpb = spb * KK_SUB;
        nblk = (int)((nsub + spb - 1) / spb);
    }
    const int ng = (p + 15) / 16;
    int NG = 1;
    while (NG < ng) NG *= 2;
    const size_t shm = (size_t)NG * 256 * sizeof(double);
    double* part = ctx->partials;
    {
        kk_prof_scope ps(ctx, "k_block_gram");
        dim3 g(nblk), b(KK_TPB);
        switch (NG) {
            case 1: hipLaunchKernelGGL((k_block_gram<1>), g, b, shm, ctx->stream, X, ldx, p, Y, ldy, q, ld, rpb, part); break;
            case 2: hipLaunchKernelGGL((k_block_gram<2>), g, b, shm, ctx->stream, X, ldx, p, Y, ldy, q, ld, rpb, part); break;
            case 4: hipLaunchKernelGGL((k_block_gram<4>), g, b, shm, ctx->stream, X, ldx, p, Y, ldy, q, ld, rpb, part); break;
            default: hipLaunchKernelGGL((k_block_gram<8>), g, b, shm, ctx->stream, X, ldx, p, Y, ldy, q, ld, rpb, part); break;
        }
    }
    KK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_finalize_gram, dim3((p * q + KK_TPB - 1) / KK_TPB), dim3(KK_TPB), 0, ctx->stream, part, nblk, NG, p, q,
                       C_dev, ldc);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

// Wout[:, j] = beta*Win[:, j] + alpha * sum_c V[:, c] S_dev[c*nb + j], j < nb <= 16; optional norms2_dev[nb]
int kk_launch_block_update(kk_ctx ctx, const double* V, int64_t ld, int m, const double* Win, double* Wout, int64_t ldw_in,
                           int64_t ldw_out, int nb, const double* S_dev, double alpha, double beta, double* norms2_dev) {
    if (nb <= 0) return KK_OK;
    if (nb > 16) { kk_set_error("kk_launch_block_update: nb=%d > 16", nb); return KK_ERR_INVALID; }
    kk_part p = kk_partition(ctx, ld);
    dim3 g(p.nblk), b(KK_TPB);
    double* part = norms2_dev ? ctx->partials : nullptr;
    const bool bz = (beta == 0.0);
    {
        kk_prof_scope ps(ctx, "k_block_update");
#define BU_CASE(NBT) \
        if (bz) hipLaunchKernelGGL((k_block_update<NBT, true>), g, b, 0, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, alpha, beta, p.rpb, part); \
        else hipLaunchKernelGGL((k_block_update<NBT, false>), g, b, 0, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, alpha, beta, p.rpb, part);
        if (nb <= 4) { BU_CASE(4) } else if (nb <= 8) { BU_CASE(8) } else { BU_CASE(16) }
#undef BU_CASE
    }
    KK_HIP(hipGetLastError());
    if (norms2_dev) {
        KK_TRY(finalize_rows(ctx, part, p.nblk, nb, norms2_dev, nullptr));
        KK_TRY(kk_allreduce(ctx, norms2_dev, nb));
    }
    return KK_OK;
}

