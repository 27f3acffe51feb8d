// Hand-written gfx950 (CDNA4, wave64) kernels of the Krylov expand! hot path.
//
// Every kernel here is HBM-bandwidth bound (<= 0.25 flop/byte).  Common design:
//   * 256-thread blocks (4 waves), every lane moves 16 B per load (global_load_dwordx4),
//     a wave instruction covers 1 KiB contiguous -> fully coalesced column streams.
//   * static even row partition: block b owns rows [b*rpb, (b+1)*rpb), rpb a multiple of 512,
//     so all blocks carry the same byte count and reductions are summed in a fixed order
//     (bitwise reproducible run to run).
//   * reductions: per-lane FMA chains -> DPP butterfly inside each row of 16 lanes ->
//     v_readlane across the 4 rows -> LDS across the 4 waves -> one partial per block in
//     HBM -> tiny finalize kernel.  No atomics.
//   * rows n..ld-1 of every column are zero and stay zero, so no kernel needs a row bound check.
#include "kk_internal.h"
#include "kk_device.h"

// BLAS-1 verbs
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KK_TPB) void k_dot(const double* __restrict__ x, const double* __restrict__ y,
                                                int64_t ld, int64_t rpb, double* __restrict__ part) {
    __shared__ double sm[4];
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int64_t r = r0 + threadIdx.x * 2;
    for (; r + 3 * KK_SUB < r1; r += 4 * KK_SUB) {
        d2 x0 = ld2(x + r), x1 = ld2(x + r + KK_SUB), x2 = ld2(x + r + 2 * KK_SUB), x3 = ld2(x + r + 3 * KK_SUB);
        d2 y0 = ld2(y + r), y1 = ld2(y + r + KK_SUB), y2 = ld2(y + r + 2 * KK_SUB), y3 = ld2(y + r + 3 * KK_SUB);
        a0 = fma(x0.x, y0.x, a0); a0 = fma(x0.y, y0.y, a0);
        a1 = fma(x1.x, y1.x, a1); a1 = fma(x1.y, y1.y, a1);
        a2 = fma(x2.x, y2.x, a2); a2 = fma(x2.y, y2.y, a2);
        a3 = fma(x3.x, y3.x, a3); a3 = fma(x3.y, y3.y, a3);
    }
    for (; r < r1; r += KK_SUB) {
        d2 x0 = ld2(x + r), y0 = ld2(y + r);
        a0 = fma(x0.x, y0.x, a0); a0 = fma(x0.y, y0.y, a0);
    }
    double t = block_sum((a0 + a1) + (a2 + a3), sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// out[0] = sum(part[0..n)), out[1] = sqrt, out[2] = 1/sqrt   (single block)
__global__ __launch_bounds__(KK_TPB) void k_finalize_scalar(const double* __restrict__ part, int n,
                                                            double* __restrict__ out, int with_sqrt) {
    __shared__ double sm[4];
    double a = 0;
    for (int i = threadIdx.x; i < n; i += KK_TPB) a += part[i];
    double t = block_sum(a, sm);
    if (threadIdx.x == 0) {
        out[0] = t;
        if (with_sqrt) {
            double s = sqrt(t);
            out[1] = s;
            out[2] = 1.0 / s;
        }
    }
}

// t[1] = sqrt(t[0]), t[2] = 1/sqrt(t[0])  (after an all-reduce of a squared norm)
__global__ void k_sqrt_triple(double* __restrict__ t) {
    const double s = sqrt(t[0]);
    t[1] = s;
    t[2] = 1.0 / s;
}

// y = b*y + a*x ; a = a_host, or a_sign * (*a_dev) [mode 1], or a_sign / (*a_dev) [mode 2]
template <bool BZERO>
__global__ __launch_bounds__(KK_TPB) void k_axpby(double* __restrict__ y, const double* __restrict__ x, int64_t ld,
                                                  int64_t rpb, double a, double b, const double* __restrict__ a_dev,
                                                  double a_sign, int a_mode) {
    if (a_dev) a = (a_mode == 2) ? a_sign / *a_dev : a_sign * *a_dev;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 xv = ld2(x + r), yv;
        if (BZERO) {
            yv.x = a * xv.x; yv.y = a * xv.y;
        } else {
            yv = ld2(y + r);
            yv.x = fma(a, xv.x, b * yv.x); yv.y = fma(a, xv.y, b * yv.y);
        }
        st2(y + r, yv);
    }
}

__global__ __launch_bounds__(KK_TPB) void k_scal(double* __restrict__ x, int64_t ld, int64_t rpb, double a,
                                                 const double* __restrict__ a_dev, int rsqrt_mode) {
    if (a_dev) a = rsqrt_mode == 1 ? 1.0 / sqrt(*a_dev) : *a_dev;
    // mode 2: *a_dev is 1 / |w| of a step whose norm the host has not seen yet (run-ahead): a zero or overflowing norm leaves the
    // vector as it is -- the test kk_persist_norm_applies makes on the host's copy when it arrives
    if (rsqrt_mode == 2 && !(a > 0.0 && a <= 1.79769313486231570815e308)) a = 1.0;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 v = ld2(x + r);
        v.x *= a; v.y *= a;
        st2(x + r, v);
    }
}

// counter-based uniform [0,1): splitmix64 of (seed, row) -> 53-bit mantissa. Independent of grid.
__global__ __launch_bounds__(KK_TPB) void k_fill_random(double* __restrict__ x, int64_t n, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * KK_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * KK_TPB) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}

__global__ __launch_bounds__(KK_TPB) void k_gather(const double* __restrict__ x, const int64_t* __restrict__ idx,
                                                   int64_t count, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * KK_TPB + threadIdx.x; i < count; i += (int64_t)gridDim.x * KK_TPB)
        out[i] = x[idx[i]];
}

// ------------------------------------------------------------------------------------------
// project: s[j] = <V_j, w'>, j < m, with w' = w - a*pre (optional) and an optional second
// right-hand side g[j] = <V_j, rhs2>.  V is read exactly once (8 m N bytes), w once.
// Lane-distributed accumulators: lane l of every wave owns columns l, 64+l, 128+l, 192+l.
// ------------------------------------------------------------------------------------------
// A block's row range is cut into row groups of RG 512-row chunks (2*RG rows per lane, held in registers
// against CB columns per load batch); what is left after the full KK_RG_P groups goes through the same code
// at RG/2, RG/4, .. 1 chunks with CB widened to keep the loads in flight (no masked slow path).
template <int RG, int CB, bool RHS2, bool KEEP = false>
__device__ __forceinline__ void proj_batch(const double* __restrict__ Vc, int64_t ld, const d2 (&wv)[RG], const d2 (&gv)[RG],
                                           int lane, int jj, double& acc, double& acc2) {
    d2 x[CB][RG];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
#pragma unroll
        for (int k = 0; k < RG; ++k) x[c][k] = KEEP ? ld2(Vc + (int64_t)c * ld + k * KK_SUB) : ld2s(Vc + (int64_t)c * ld + k * KK_SUB);
    }
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < RG; ++k) {
            t = fma(x[c][k].x, wv[k].x, t);
            t = fma(x[c][k].y, wv[k].y, t);
        }
        double tot = wave_sum(t);
        acc += (lane == jj + c) ? tot : 0.0;
        if (RHS2) {
            double t2 = 0;
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                t2 = fma(x[c][k].x, gv[k].x, t2);
                t2 = fma(x[c][k].y, gv[k].y, t2);
            }
            double tot2 = wave_sum(t2);
            acc2 += (lane == jj + c) ? tot2 : 0.0;
        }
    }
}

template <int RG, int CB, bool PRE, bool RHS2>
__device__ __forceinline__ void proj_group(const double* __restrict__ V, int64_t ld, int m, const double* __restrict__ w,
                                           const double* __restrict__ pre_vec, double a, const double* __restrict__ rhs2,
                                           int64_t off, int lane, double* smw, int keep) {
    d2 wv[RG], gv[RG];
#pragma unroll
    for (int k = 0; k < RG; ++k) {
        wv[k] = ld2(w + off + k * KK_SUB);
        if (PRE) {
            d2 p = ld2(pre_vec + off + k * KK_SUB);
            wv[k].x = fma(-a, p.x, wv[k].x);
            wv[k].y = fma(-a, p.y, wv[k].y);
        }
        if (RHS2) gv[k] = ld2(rhs2 + off + k * KK_SUB);
        else gv[k] = d2{0.0, 0.0};
    }
    // columns in segments of 64: lane l accumulates column jq + l of the segment, then folds it into its own
    // LDS slot (same thread reads and writes the slot: no barrier) -- keeps the q loop rolled (code size)
#pragma unroll 1
    for (int jq = 0; jq < m; jq += 64) {
        const int jn = min(64, m - jq);
        const double* Vq = V + (int64_t)jq * ld + off;
        double acc = 0, acc2 = 0;
        int jj = 0;
        // the last `keep` columns with plain (cache-allocating) loads: the unproject pass that follows starts with
        // exactly those columns and finds them in the Infinity Cache (measured -1.6 % on k_unproject at keep = 2)
        const int jkeep = max(0, min(jn, (m - keep) - jq));
        for (; jj + CB <= jkeep; jj += CB) proj_batch<RG, CB, RHS2>(Vq + (int64_t)jj * ld, ld, wv, gv, lane, jj, acc, acc2);
        for (; jj < jkeep; ++jj) proj_batch<RG, 1, RHS2>(Vq + (int64_t)jj * ld, ld, wv, gv, lane, jj, acc, acc2);
        for (; jj + CB <= jn; jj += CB) proj_batch<RG, CB, RHS2, true>(Vq + (int64_t)jj * ld, ld, wv, gv, lane, jj, acc, acc2);
        for (; jj < jn; ++jj) proj_batch<RG, 1, RHS2, true>(Vq + (int64_t)jj * ld, ld, wv, gv, lane, jj, acc, acc2);
        smw[jq + lane] += acc;
        if (RHS2) smw[4 * KK_MAX_M + jq + lane] += acc2;
    }
}

template <bool RHS2> struct proj_tile {
    static constexpr int RG = RHS2 ? KK_RG_P2 : KK_RG_P;
    static constexpr int CB = RHS2 ? KK_CB_P2 : KK_CB_P;
};

template <int H, bool PRE, bool RHS2>
__device__ __forceinline__ void proj_tail(const double* __restrict__ V, int64_t ld, int m, const double* __restrict__ w,
                                          const double* __restrict__ pre_vec, double a, const double* __restrict__ rhs2,
                                          int64_t& rg, int64_t r1, int tid, int lane, double* smw, int keep) {
    if constexpr (H >= 1) {
        constexpr int LOADS = proj_tile<RHS2>::RG * proj_tile<RHS2>::CB;
        constexpr int CBT = (LOADS / H) > 8 ? 8 : (LOADS / H);
        // at most one group of H chunks, then H/2, ...; single chunks until the range is used up
        while (rg + (int64_t)H * KK_SUB <= r1) {
            proj_group<H, CBT, PRE, RHS2>(V, ld, m, w, pre_vec, a, rhs2, rg + tid * 2, lane, smw, keep);
            rg += (int64_t)H * KK_SUB;
            if (H > 1) break;
        }
        proj_tail<H / 2, PRE, RHS2>(V, ld, m, w, pre_vec, a, rhs2, rg, r1, tid, lane, smw, keep);
    }
}

template <bool PRE, bool RHS2>
__global__ __launch_bounds__(KK_TPB) void k_project(const double* __restrict__ V, int64_t ld, int m,
                                                    const double* __restrict__ w, const double* __restrict__ pre_vec,
                                                    const double* __restrict__ pre_a, const double* __restrict__ rhs2,
                                                    int64_t rpb, double* __restrict__ part, int keep) {
    __shared__ double sm[(RHS2 ? 2 : 1) * 4 * KK_MAX_M];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double* smw = sm + wave * KK_MAX_M;     // this wave's accumulator row; lane l owns slots l, 64+l, 128+l, 192+l
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        smw[q * 64 + lane] = 0.0;
        if (RHS2) smw[4 * KK_MAX_M + q * 64 + lane] = 0.0;
    }
    double a = 0;
    if (PRE) a = *pre_a;
    int64_t rg = r0;
    constexpr int RG = proj_tile<RHS2>::RG, CB = proj_tile<RHS2>::CB;
    for (; rg + (int64_t)RG * KK_SUB <= r1; rg += (int64_t)RG * KK_SUB)
        proj_group<RG, CB, PRE, RHS2>(V, ld, m, w, pre_vec, a, rhs2, rg + tid * 2, lane, smw, keep);
    proj_tail<RG / 2, PRE, RHS2>(V, ld, m, w, pre_vec, a, rhs2, rg, r1, tid, lane, smw, keep);
    __syncthreads();
    if (tid < m) {
        double t = (sm[tid] + sm[KK_MAX_M + tid]) + (sm[2 * KK_MAX_M + tid] + sm[3 * KK_MAX_M + tid]);
        part[(int64_t)tid * KK_MAX_BLOCKS + blockIdx.x] = t;
        if (RHS2) {
            const double* s2 = sm + 4 * KK_MAX_M;
            double t2 = (s2[tid] + s2[KK_MAX_M + tid]) + (s2[2 * KK_MAX_M + tid] + s2[3 * KK_MAX_M + tid]);
            part[(int64_t)(KK_MAX_M + tid) * KK_MAX_BLOCKS + blockIdx.x] = t2;
        }
    }
}

// one wave per output value: ws_a[j] = sum_b part[j][b]; second segment (rows KK_MAX_M + j) -> ws_b[j]
__global__ __launch_bounds__(KK_TPB) void k_finalize_project(const double* __restrict__ part, int nblk, int m,
                                                             double* __restrict__ ws_a, double* __restrict__ ws_b) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = ws_b ? 2 * m : m;
    if (v >= total) return;
    const int row = (v < m) ? v : KK_MAX_M + (v - m);
    const double* p = part + (int64_t)row * KK_MAX_BLOCKS;
    double a = 0;
    for (int b = lane; b < nblk; b += 64) a += p[b];
    a = wave_sum(a);
    if (lane == 0) {
        if (v < m) ws_a[v] = a;
        else ws_b[v - m] = a;
    }
}

// ------------------------------------------------------------------------------------------
// unproject: w_out = beta*w_in + alpha * sum_j c[j] V_j, with optional fused |w_out|^2.
// Coefficients come from the kernarg segment (host vector, scalar loads) or from device memory;
// coefficient add_idx may get a device scalar added (folds the Lanczos "w -= alpha v" into the pass).
// ------------------------------------------------------------------------------------------
template <int RG, int CB, bool NORM, bool BZERO>
__device__ __forceinline__ void unproj_group(const double* __restrict__ V, int64_t ld, int m, const double* w_in, double* w_out,
                                             const double* sc, double beta, int64_t off, double& nacc) {
    d2 wv[RG];
#pragma unroll
    for (int k = 0; k < RG; ++k) {
        if (!BZERO) {
            wv[k] = ld2(w_in + off + k * KK_SUB);
            wv[k].x *= beta; wv[k].y *= beta;
        } else {
            wv[k] = d2{0.0, 0.0};
        }
    }
    const double* Vo = V + off;
    // columns from the LAST to the first: the project pass that precedes an unproject streams the basis in ascending
    // column order, so its tail (the last ~256 MB = 3 columns of a 10M-row basis) is still in the Infinity Cache
    int j = m - CB;
    for (; j >= 0; j -= CB) {
        d2 x[CB][RG];
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int k = 0; k < RG; ++k) x[c][k] = ld2s(Vo + (int64_t)(j + c) * ld + k * KK_SUB);
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const double s = sc[j + c];
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                wv[k].x = fma(s, x[c][k].x, wv[k].x);
                wv[k].y = fma(s, x[c][k].y, wv[k].y);
            }
        }
    }
    const int mrem = j + CB;     // columns [0, mrem) are left (mrem < CB)
    for (j = 0; j < mrem; ++j) {
        const double s = sc[j];
        d2 x[RG];
#pragma unroll
        for (int k = 0; k < RG; ++k) x[k] = ld2s(Vo + (int64_t)j * ld + k * KK_SUB);
#pragma unroll
        for (int k = 0; k < RG; ++k) {
            wv[k].x = fma(s, x[k].x, wv[k].x);
            wv[k].y = fma(s, x[k].y, wv[k].y);
        }
    }
#pragma unroll
    for (int k = 0; k < RG; ++k) {
        st2s(w_out + off + k * KK_SUB, wv[k]);
        if (NORM) {
            nacc = fma(wv[k].x, wv[k].x, nacc);
            nacc = fma(wv[k].y, wv[k].y, nacc);
        }
    }
}

template <int H, bool NORM, bool BZERO>
__device__ __forceinline__ void unproj_tail(const double* __restrict__ V, int64_t ld, int m, const double* w_in, double* w_out,
                                            const double* sc, double beta, int64_t& rg, int64_t r1, int tid, double& nacc) {
    if constexpr (H >= 1) {
        constexpr int CBT = (KK_RG_U * KK_CB_U / H) > 8 ? 8 : (KK_RG_U * KK_CB_U / H);
        while (rg + (int64_t)H * KK_SUB <= r1) {
            unproj_group<H, CBT, NORM, BZERO>(V, ld, m, w_in, w_out, sc, beta, rg + tid * 2, nacc);
            rg += (int64_t)H * KK_SUB;
            if (H > 1) break;
        }
        unproj_tail<H / 2, NORM, BZERO>(V, ld, m, w_in, w_out, sc, beta, rg, r1, tid, nacc);
    }
}

template <bool NORM, bool BZERO>
__global__ __launch_bounds__(KK_TPB) void k_unproject(const double* __restrict__ V, int64_t ld, int m,
                                                      const double* w_in, double* w_out,
                                                      kk_coef ch, const double* __restrict__ coef_dev, double alpha,
                                                      double beta, int add_idx, const double* __restrict__ add_dev,
                                                      int64_t rpb, double* __restrict__ part) {
    __shared__ double sc[KK_MAX_M];
    __shared__ double sm[4];
    const int tid = threadIdx.x;
    if (tid < m) {
        double c = coef_dev ? coef_dev[tid] : ch.v[tid];
        if (tid == add_idx) c += *add_dev;
        sc[tid] = alpha * c;
    }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double nacc = 0;
    int64_t rg = r0;
    for (; rg + (int64_t)KK_RG_U * KK_SUB <= r1; rg += (int64_t)KK_RG_U * KK_SUB)
        unproj_group<KK_RG_U, KK_CB_U, NORM, BZERO>(V, ld, m, w_in, w_out, sc, beta, rg + tid * 2, nacc);
    unproj_tail<KK_RG_U / 2, NORM, BZERO>(V, ld, m, w_in, w_out, sc, beta, rg, r1, tid, nacc);
    if (NORM) {
        double t = block_sum(nacc, sm);
        if (tid == 0) part[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// fused  unproject(pass i) + project(pass i+1):   w1 = w - V c ;  s2 = V' w1      (V read ONCE)
// For the 2-pass orthogonalisers (CGS2 / low-sync MGS2, orthonormal.jl:394-399,434-439) this turns
// 4 sweeps over the basis into 3.  A block holds a 128-row x m tile of V in registers: wave v owns
// columns v, v+4, v+8, ... (CT per wave), every lane 2 rows (16 B, 1 KiB contiguous per wave load).
//   step 1: per-wave partial of (V c) over its columns -> LDS -> all waves get w1 for the 128 rows
//   step 2: every wave dots ITS columns (still in registers) with w1 -> per-lane accumulators
// ------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(KK_TPB) void k_unproj_proj(const double* __restrict__ V, int64_t ld, int m, const double* w_in,
                                                        double* w_out, kk_coef ch, const double* __restrict__ coef_dev,
                                                        int64_t rpb, double* __restrict__ part, double* __restrict__ part_nrm) {
    __shared__ d2 red[4][64];
    __shared__ double sc[KK_MAX_M];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < m) sc[tid] = -(coef_dev ? coef_dev[tid] : ch.v[tid]);
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = 0.0;
    double nacc = 0.0;
    for (int64_t r = r0 + lane * 2; r < r1; r += 128) {
        d2 x[CT];
        d2 u{0.0, 0.0};
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int c = wave + 4 * i;
            if (c < m) x[i] = ld2s(V + (int64_t)c * ld + r);
            else x[i] = d2{0.0, 0.0};
        }
        d2 wv = ld2(w_in + r);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int c = wave + 4 * i;
            if (c < m) {
                const double s = sc[c];
                u.x = fma(s, x[i].x, u.x); u.y = fma(s, x[i].y, u.y);
            }
        }
        red[wave][lane] = u;
        __syncthreads();
        const d2 a = red[0][lane], b = red[1][lane], cc = red[2][lane], d = red[3][lane];
        wv.x += (a.x + b.x) + (cc.x + d.x);
        wv.y += (a.y + b.y) + (cc.y + d.y);
        if (wave == 0) {
            st2(w_out + r, wv);
            nacc = fma(wv.x, wv.x, nacc); nacc = fma(wv.y, wv.y, nacc);
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            acc[i] = fma(x[i].x, wv.x, acc[i]);
            acc[i] = fma(x[i].y, wv.y, acc[i]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int c = wave + 4 * i;
        const double tot = wave_sum(acc[i]);
        if (c < m && lane == 0) part[(int64_t)c * KK_MAX_BLOCKS + blockIdx.x] = tot;
    }
    if (part_nrm && wave == 0) {
        const double tn = wave_sum(nacc);
        if (lane == 0) part_nrm[blockIdx.x] = tn;
    }
}

// low-sync MGS coefficient solve on the device: (I + L) s = p, L = strictly-lower Gram matrix of the
// basis (row-major, leading dimension cap).  One 256-thread block, exact column-oriented forward
// substitution in LDS (m <= 256 barriers of a single block, ~5 us).  If g_ride != nullptr the Gram
// row of the newest basis vector (g_ride[0..m-2]) is first stored into L[newest][.].  coef_out gets
// s (+ *a0 on the last entry: the Lanczos "w -= alpha0 v" folded into the update), s_out the plain s.
#define KK_LS_AHEAD 8
__global__ __launch_bounds__(KK_TPB) void k_lowsync_solve(const double* __restrict__ p, const double* __restrict__ g_ride,
                                                          double* L, int cap, int m, int newest,
                                                          const double* __restrict__ a0, double* __restrict__ coef_out,
                                                          double* __restrict__ s_out) {
    __shared__ double rhs[KK_MAX_M];
    const int i = threadIdx.x;
    if (g_ride && i < m - 1) L[(int64_t)newest * cap + i] = g_ride[i];
    if (i < m) rhs[i] = p[i];
    // thread i walks row i of L; its entries are fetched KK_LS_AHEAD steps ahead of their use so that the global-load
    // latency is not paid once per barrier (it was: ~130 ns x m per solve)
    const double* lrow = (g_ride && i == newest) ? g_ride : L + (int64_t)i * cap;
    const bool act = i < m;
    double la[KK_LS_AHEAD];
#pragma unroll
    for (int u = 0; u < KK_LS_AHEAD; ++u) la[u] = (act && u < i) ? lrow[u] : 0.0;
    __syncthreads();
    for (int j0 = 0; j0 < m - 1; j0 += KK_LS_AHEAD) {
#pragma unroll
        for (int u = 0; u < KK_LS_AHEAD; ++u) {
            const int j = j0 + u;
            if (j < m - 1) {   // uniform
                const double lij = la[u];
                const int jn = j + KK_LS_AHEAD;
                la[u] = (act && jn < i) ? lrow[jn] : 0.0;
                const double sj = rhs[j];
                if (i > j && act) rhs[i] = fma(-lij, sj, rhs[i]);
                __syncthreads();
            }
        }
    }
    if (i < m) {
        const double s = rhs[i];
        s_out[i] = s;
        coef_out[i] = (a0 && i == m - 1) ? s + *a0 : s;
    }
}

// Row-sharded Lanczos step (krylovkit_hip/dist.py), coefficient algebra between the two all-reduces in ONE launch:
//   buf = [alpha0 | p = V'w (m) | g = V'v (m)] (already summed over the ranks);  rhs = p - alpha0 g = V'(w - alpha0 v);
//   lowsync != 0: g[0:m-1] is stored as the Gram row of the newest vector and (I + L) s = rhs is solved exactly;
//   coef = s with alpha0 folded into the last entry (the update w -= V coef then also removes alpha0 v);
//   res[0] = alpha0, res[1] = s[m-1]  (alpha = res[0] + res[1] on the host).
__global__ __launch_bounds__(KK_TPB) void k_lanczos_coef(const double* __restrict__ buf, double* L, int cap, int m, int lowsync,
                                                         double* __restrict__ coef_out, double* __restrict__ res) {
    __shared__ double rhs[KK_MAX_M];
    const int i = threadIdx.x;
    const double a0 = buf[0];
    const double* p = buf + 1;
    const double* g = buf + 1 + m;
    if (i < m) rhs[i] = fma(-a0, g[i], p[i]);
    if (lowsync && i < m - 1) L[(int64_t)(m - 1) * cap + i] = g[i];
    __syncthreads();
    if (lowsync) {
        const double* lrow = (i == m - 1) ? g : L + (int64_t)i * cap;
        const bool act = i < m;
        double la[KK_LS_AHEAD];
#pragma unroll
        for (int u = 0; u < KK_LS_AHEAD; ++u) la[u] = (act && u < i) ? lrow[u] : 0.0;
        for (int j0 = 0; j0 < m - 1; j0 += KK_LS_AHEAD) {
#pragma unroll
            for (int u = 0; u < KK_LS_AHEAD; ++u) {
                const int j = j0 + u;
                if (j < m - 1) {   // uniform
                    const double lij = la[u];
                    const int jn = j + KK_LS_AHEAD;
                    la[u] = (act && jn < i) ? lrow[jn] : 0.0;
                    const double sj = rhs[j];
                    if (i > j && act) rhs[i] = fma(-lij, sj, rhs[i]);
                    __syncthreads();
                }
            }
        }
    }
    if (i < m) coef_out[i] = (i == m - 1) ? rhs[i] + a0 : rhs[i];
    if (i == 0) { res[0] = a0; res[1] = rhs[m - 1]; }
}
// sc = {1/sqrt(nrm2), sqrt(nrm2)}, res2 = nrm2: the device scalars of the speculative next-step apply
__global__ void k_norm_scalars(const double* __restrict__ nrm2, double* __restrict__ sc, double* __restrict__ res2) {
    const double n2 = *nrm2, r = sqrt(n2);
    sc[0] = 1.0 / r; sc[1] = r; *res2 = n2;
}

// ------------------------------------------------------------------------------------------
// strict modified Gram-Schmidt step (src/orthonormal.jl:417-421), fused across the j boundary:
//   w -= s_prev * q_prev   (axpy of step j-1, skipped if q_prev == nullptr)
//   partial <q_next, w>    (dot of step j, skipped if q_next == nullptr)
//   partial |w|^2          (when NORM)
// 32 N bytes per basis vector instead of 40 N for separate dot + axpy.
// ------------------------------------------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(KK_TPB) void k_mgs_step(double* __restrict__ w, int64_t ld, int64_t rpb,
                                                     const double* __restrict__ q_prev,
                                                     const double* __restrict__ s_prev, const double* __restrict__ q_next,
                                                     double* __restrict__ part_dot, double* __restrict__ part_nrm) {
    __shared__ double sm[4];
    const double s = q_prev ? *s_prev : 0.0;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double a0 = 0, a1 = 0, n0 = 0;
    int64_t r = r0 + threadIdx.x * 2;
    for (; r + KK_SUB < r1; r += 2 * KK_SUB) {
        d2 w0 = ld2(w + r), w1 = ld2(w + r + KK_SUB);
        if (q_prev) {
            d2 p0 = ld2(q_prev + r), p1 = ld2(q_prev + r + KK_SUB);
            w0.x = fma(-s, p0.x, w0.x); w0.y = fma(-s, p0.y, w0.y);
            w1.x = fma(-s, p1.x, w1.x); w1.y = fma(-s, p1.y, w1.y);
            st2(w + r, w0); st2(w + r + KK_SUB, w1);
        }
        if (q_next) {
            d2 q0 = ld2(q_next + r), q1 = ld2(q_next + r + KK_SUB);
            a0 = fma(q0.x, w0.x, a0); a0 = fma(q0.y, w0.y, a0);
            a1 = fma(q1.x, w1.x, a1); a1 = fma(q1.y, w1.y, a1);
        }
        if (NORM) {
            n0 = fma(w0.x, w0.x, n0); n0 = fma(w0.y, w0.y, n0);
            n0 = fma(w1.x, w1.x, n0); n0 = fma(w1.y, w1.y, n0);
        }
    }
    for (; r < r1; r += KK_SUB) {
        d2 w0 = ld2(w + r);
        if (q_prev) {
            d2 p0 = ld2(q_prev + r);
            w0.x = fma(-s, p0.x, w0.x); w0.y = fma(-s, p0.y, w0.y);
            st2(w + r, w0);
        }
        if (q_next) {
            d2 q0 = ld2(q_next + r);
            a0 = fma(q0.x, w0.x, a0); a0 = fma(q0.y, w0.y, a0);
        }
        if (NORM) { n0 = fma(w0.x, w0.x, n0); n0 = fma(w0.y, w0.y, n0); }
    }
    if (q_next) {
        double t = block_sum(a0 + a1, sm);
        if (threadIdx.x == 0) part_dot[blockIdx.x] = t;
    }
    if (NORM) {
        double t = block_sum(n0, sm);
        if (threadIdx.x == 0) part_nrm[blockIdx.x] = t;
    }
}

// ==========================================================================================
// host-side launchers
// ==========================================================================================
// ws_a[j] = sum over blocks of partial row j (j < m); optional second segment (rows KK_MAX_M + j) -> ws_b[j]
int finalize_rows(kk_ctx ctx, const double* part, int nblk, int m, double* ws_a, double* ws_b) {
    hipLaunchKernelGGL(k_finalize_project, dim3(((ws_b ? 2 * m : m) + 3) / 4), dim3(KK_TPB), 0, ctx->stream, part, nblk, m, ws_a, ws_b);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int finalize_scalar(kk_ctx ctx, int part_row_idx, int n, double* out, bool with_sqrt) {
    const bool sharded = kk_sharded(ctx);
    hipLaunchKernelGGL(k_finalize_scalar, dim3(1), dim3(KK_TPB), 0, ctx->stream, part_row(ctx, part_row_idx), n,
                       out, (with_sqrt && !sharded) ? 1 : 0);
    KK_HIP(hipGetLastError());
    if (sharded) {  // sum the local partial over the ranks, then (re)derive sqrt and 1/sqrt
        KK_TRY(kk_allreduce(ctx, out, 1));
        if (with_sqrt) {
            hipLaunchKernelGGL(k_sqrt_triple, dim3(1), dim3(1), 0, ctx->stream, out);
            KK_HIP(hipGetLastError());
        }
    }
    return KK_OK;
}

int kk_launch_dot(kk_ctx ctx, const double* x, const double* y, int64_t ld, double* out) {
    kk_part p = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_dot");
        hipLaunchKernelGGL(k_dot, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, x, y, ld, p.rpb, part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, p.nblk, out, false);
}

int kk_launch_nrm2(kk_ctx ctx, const double* x, int64_t ld, double* out3) {
    kk_part p = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_dot");
        hipLaunchKernelGGL(k_dot, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, x, x, ld, p.rpb, part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, p.nblk, out3, true);
}

int kk_launch_axpby(kk_ctx ctx, double* y, const double* x, int64_t ld, double a, double b, const double* a_dev,
                    double a_dev_sign, int a_dev_mode) {
    kk_prof_scope ps(ctx, "k_axpby");
    kk_part p = kk_partition(ctx, ld);
    if (b == 0.0)
        hipLaunchKernelGGL(k_axpby<true>, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, y, x, ld, p.rpb, a, b, a_dev,
                           a_dev_sign, a_dev_mode);
    else
        hipLaunchKernelGGL(k_axpby<false>, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, y, x, ld, p.rpb, a, b, a_dev,
                           a_dev_sign, a_dev_mode);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_scal(kk_ctx ctx, double* x, int64_t ld, double a, const double* a_dev, int rsqrt_mode) {
    kk_prof_scope ps(ctx, "k_scal");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_scal, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, x, ld, p.rpb, a, a_dev, rsqrt_mode);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_copy_scal(kk_ctx ctx, double* y, const double* x, int64_t ld, double a) {
    return kk_launch_axpby(ctx, y, x, ld, a, 0.0, nullptr, 1.0, 0);
}

int kk_launch_fill_random(kk_ctx ctx, double* x, int64_t n, uint64_t seed) {
    kk_prof_scope ps(ctx, "k_fill_random");
    int nb = (int)std::min<int64_t>((n + KK_TPB - 1) / KK_TPB, 4096);
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_fill_random, dim3(nb), dim3(KK_TPB), 0, ctx->stream, x, n, seed);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_gather(kk_ctx ctx, const double* x, const int64_t* idx, int64_t count, double* out) {
    kk_prof_scope ps(ctx, "k_gather");
    if (count <= 0) return KK_OK;
    int nb = (int)std::min<int64_t>((count + KK_TPB - 1) / KK_TPB, 4096);
    hipLaunchKernelGGL(k_gather, dim3(nb), dim3(KK_TPB), 0, ctx->stream, x, idx, count, out);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_project(kk_ctx ctx, const double* V, int64_t ld, int m, const double* w, const double* pre_vec,
                      const double* pre_a_dev, const double* rhs2, double* out_s, double* out_g) {
    kk_part p = kk_partition(ctx, ld);
    dim3 g(p.nblk), b(KK_TPB);
    double* part = ctx->partials;
    // columns read last stay cache-allocated for the unproject pass that follows: as many as fit keep_mb (default
    // 160 MB of the 256 MB Infinity Cache = 2 columns of a 10M-row basis), at most 16
    const int keep = (int)std::min<int64_t>(16, (int64_t)ctx->keep_mb * 1000000 / (ld * (int64_t)sizeof(double)));
    std::unique_ptr<kk_prof_scope> ps(new kk_prof_scope(ctx, "k_project"));
    if (pre_vec && rhs2)
        hipLaunchKernelGGL((k_project<true, true>), g, b, 0, ctx->stream, V, ld, m, w, pre_vec, pre_a_dev, rhs2, p.rpb, part, keep);
    else if (pre_vec)
        hipLaunchKernelGGL((k_project<true, false>), g, b, 0, ctx->stream, V, ld, m, w, pre_vec, pre_a_dev, rhs2, p.rpb, part, keep);
    else if (rhs2)
        hipLaunchKernelGGL((k_project<false, true>), g, b, 0, ctx->stream, V, ld, m, w, pre_vec, pre_a_dev, rhs2, p.rpb, part, keep);
    else
        hipLaunchKernelGGL((k_project<false, false>), g, b, 0, ctx->stream, V, ld, m, w, pre_vec, pre_a_dev, rhs2, p.rpb, part, keep);
    ps.reset();
    KK_HIP(hipGetLastError());
    const int total = rhs2 ? 2 * m : m;
    hipLaunchKernelGGL(k_finalize_project, dim3((total + 3) / 4), dim3(KK_TPB), 0, ctx->stream, part, p.nblk, m,
                       out_s, rhs2 ? out_g : (double*)nullptr);
    KK_HIP(hipGetLastError());
    KK_TRY(kk_allreduce(ctx, out_s, m));
    if (rhs2) KK_TRY(kk_allreduce(ctx, out_g, m));
    return KK_OK;
}

int kk_launch_unproject(kk_ctx ctx, const double* V, int64_t ld, int m, const double* w_in, double* w_out,
                        const kk_coef* coef_host, const double* coef_dev, double alpha, double beta, int add_idx,
                        const double* add_dev, double* nrm_out3) {
    kk_part p = kk_partition(ctx, ld);
    dim3 g(p.nblk), b(KK_TPB);
    static const kk_coef zero_coef = {};
    const kk_coef& ch = coef_host ? *coef_host : zero_coef;
    double* part = part_row(ctx, PART_SCAL_A);
    const bool norm = nrm_out3 != nullptr, bzero = (beta == 0.0);
    std::unique_ptr<kk_prof_scope> ps(new kk_prof_scope(ctx, "k_unproject"));
    if (norm && bzero)
        hipLaunchKernelGGL((k_unproject<true, true>), g, b, 0, ctx->stream, V, ld, m, w_in, w_out, ch, coef_dev, alpha, beta, add_idx, add_dev, p.rpb, part);
    else if (norm)
        hipLaunchKernelGGL((k_unproject<true, false>), g, b, 0, ctx->stream, V, ld, m, w_in, w_out, ch, coef_dev, alpha, beta, add_idx, add_dev, p.rpb, part);
    else if (bzero)
        hipLaunchKernelGGL((k_unproject<false, true>), g, b, 0, ctx->stream, V, ld, m, w_in, w_out, ch, coef_dev, alpha, beta, add_idx, add_dev, p.rpb, part);
    else
        hipLaunchKernelGGL((k_unproject<false, false>), g, b, 0, ctx->stream, V, ld, m, w_in, w_out, ch, coef_dev, alpha, beta, add_idx, add_dev, p.rpb, part);
    ps.reset();
    KK_HIP(hipGetLastError());
    if (norm) return finalize_scalar(ctx, PART_SCAL_A, p.nblk, nrm_out3, true);
    return KK_OK;
}

int kk_launch_mgs_step(kk_ctx ctx, double* w, int64_t ld, const double* q_prev, const double* s_prev_dev,
                       const double* q_next, double* dot_out, double* nrm_out3) {
    kk_part p = kk_partition(ctx, ld);
    dim3 g(p.nblk), b(KK_TPB);
    double* pd = part_row(ctx, PART_SCAL_A);
    double* pn = part_row(ctx, PART_SCAL_B);
    std::unique_ptr<kk_prof_scope> ps(new kk_prof_scope(ctx, "k_mgs_step"));
    if (nrm_out3)
        hipLaunchKernelGGL((k_mgs_step<true>), g, b, 0, ctx->stream, w, ld, p.rpb, q_prev, s_prev_dev, q_next, pd, pn);
    else
        hipLaunchKernelGGL((k_mgs_step<false>), g, b, 0, ctx->stream, w, ld, p.rpb, q_prev, s_prev_dev, q_next, pd, pn);
    ps.reset();
    KK_HIP(hipGetLastError());
    if (q_next) KK_TRY(finalize_scalar(ctx, PART_SCAL_A, p.nblk, dot_out, false));
    if (nrm_out3) KK_TRY(finalize_scalar(ctx, PART_SCAL_B, p.nblk, nrm_out3, true));
    return KK_OK;
}

// fused w_out = w_in - V c ; out_s = V' w_out  (m <= 128).  Optional |w_out|^2 -> nrm_out3.
int kk_launch_unproj_proj(kk_ctx ctx, const double* V, int64_t ld, int m, const double* w_in, double* w_out,
                          const kk_coef* coef_host, const double* coef_dev, double* out_s, double* nrm_out3) {
    if (m > 128) { kk_set_error("kk_launch_unproj_proj: m=%d > 128", m); return KK_ERR_INVALID; }
    // rows per block: multiple of 512 (128 | 512); cap the grid so the per-column partial rows fit
    kk_part p = kk_partition(ctx, ld);
    static const kk_coef zero_coef = {};
    const kk_coef& ch = coef_host ? *coef_host : zero_coef;
    double* part = ctx->partials;
    double* pn = part_row(ctx, PART_SCAL_A);
    dim3 g(p.nblk), b(KK_TPB);
    const int ct = (m + 3) / 4;
    {
        kk_prof_scope ps(ctx, "k_unproj_proj");
#define UP_CASE(CTT) hipLaunchKernelGGL((k_unproj_proj<CTT>), g, b, 0, ctx->stream, V, ld, m, w_in, w_out, ch, coef_dev, p.rpb, part, nrm_out3 ? pn : (double*)nullptr)
        if (ct <= 4) UP_CASE(4);
        else if (ct <= 8) UP_CASE(8);
        else if (ct <= 16) UP_CASE(16);
        else if (ct <= 24) UP_CASE(24);
        else UP_CASE(32);
#undef UP_CASE
    }
    KK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_finalize_project, dim3((m + 3) / 4), dim3(KK_TPB), 0, ctx->stream, part, p.nblk, m, out_s, (double*)nullptr);
    KK_HIP(hipGetLastError());
    KK_TRY(kk_allreduce(ctx, out_s, m));
    if (nrm_out3) return finalize_scalar(ctx, PART_SCAL_A, p.nblk, nrm_out3, true);
    return KK_OK;
}

int kk_launch_lowsync_solve(kk_ctx ctx, const double* p, const double* g_ride, double* L, int cap, int m, int newest,
                            const double* a0_dev, double* coef_out, double* s_out) {
    hipLaunchKernelGGL(k_lowsync_solve, dim3(1), dim3(KK_TPB), 0, ctx->stream, p, g_ride, L, cap, m, newest, a0_dev, coef_out,
                       s_out);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_lanczos_coef(kk_ctx ctx, const double* buf, double* L, int cap, int m, int lowsync, double* coef_out, double* res) {
    hipLaunchKernelGGL(k_lanczos_coef, dim3(1), dim3(KK_TPB), 0, ctx->stream, buf, L, cap, m, lowsync, coef_out, res);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_norm_scalars(kk_ctx ctx, const double* nrm2, double* sc, double* res2) {
    hipLaunchKernelGGL(k_norm_scalars, dim3(1), dim3(1), 0, ctx->stream, nrm2, sc, res2);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
