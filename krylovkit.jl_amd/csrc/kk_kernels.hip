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
#include <memory>

typedef double2 d2;
typedef double v4d __attribute__((ext_vector_type(4)));  // MFMA f64 16x16x4 accumulator fragment
__device__ __forceinline__ int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }

__device__ __forceinline__ d2 ld2(const double* p) { return *reinterpret_cast<const d2*>(p); }
// streaming (read-once) basis loads: non-temporal so that the 8 GB basis stream does not evict the
// work vector w / the coefficient tables from L2 and the Infinity Cache
// (measured on the 10M-row Lanczos sweep: 560 -> 611 it/s).  -DKK_NO_NT_LOADS restores plain loads.
__device__ __forceinline__ d2 ld2s(const double* p) {
#ifndef KK_NO_NT_LOADS
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
    return d2{t.x, t.y};
#else
    return *reinterpret_cast<const d2*>(p);
#endif
}
__device__ __forceinline__ int2 ldi2s(const int32_t* p) {
#ifndef KK_NO_NT_LOADS
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i t = __builtin_nontemporal_load(reinterpret_cast<const v2i*>(p));
    return int2{t.x, t.y};
#else
    return *reinterpret_cast<const int2*>(p);
#endif
}
__device__ __forceinline__ void st2(double* p, d2 v) { *reinterpret_cast<d2*>(p) = v; }
// work-vector store of the unproject pass: non-temporal (write-around) -- the 8(m+1)N-byte basis stream of the
// same kernel would evict it before its next reader anyway; measured -1..-3 % on k_unproject
__device__ __forceinline__ void st2s(double* p, d2 v) {
#ifndef KK_NO_NT_LOADS
    typedef double v2d __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(v2d{v.x, v.y}, reinterpret_cast<v2d*>(p));
#else
    *reinterpret_cast<d2*>(p) = v;
#endif
}


template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// Sum over the 64 lanes of a wave; result is wave-uniform (same bits in every lane).
__device__ __forceinline__ double wave_sum(double v) {
#ifndef KK_NO_DPP
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror  -> every lane holds the sum of its row of 16
    double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
    return (r0 + r1) + (r2 + r3);
#else
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
#endif
}
// Sum over the 32-lane half a lane belongs to (DPP row ops + one cross-row readlane pair).
__device__ __forceinline__ double block_sum(double v, double* sm /* >= 4 doubles */) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[wave] = v;
    __syncthreads();
    double t = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------------------------------
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
    if (a_dev) a = rsqrt_mode ? 1.0 / sqrt(*a_dev) : *a_dev;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 v = ld2(x + r);
        v.x *= a; v.y *= a;
        st2(x + r, v);
    }
}

// CG update fused (linsolve/cg.jl:63-66): x += alpha p ; r -= alpha q ; partial |r|^2
__global__ __launch_bounds__(KK_TPB) void k_cg_update(double* __restrict__ x, const double* __restrict__ p, double* __restrict__ r,
                                                      const double* __restrict__ q, int64_t ld, int64_t rpb, double alpha,
                                                      const double* __restrict__ pq_dev, double* __restrict__ part) {
    __shared__ double sm[4];
    if (pq_dev) alpha = alpha / *pq_dev;   // alpha = rho / <p, q> with the inner product still on the device
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double acc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 xv = ld2(x + i), pv = ld2(p + i), rv = ld2(r + i), qv = ld2(q + i);
        xv.x = fma(alpha, pv.x, xv.x); xv.y = fma(alpha, pv.y, xv.y);
        rv.x = fma(-alpha, qv.x, rv.x); rv.y = fma(-alpha, qv.y, rv.y);
        st2(x + i, xv); st2(r + i, rv);
        acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
    }
    double t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---- BiCGStab (linsolve/bicgstab.jl:118-199) as three fused vector kernels; every scalar of the recurrence
// stays in the context's device scalars sc[] = {rho, rho_old, sigma, alpha, omega, <t,s>, <t,t>}
enum { BI_RHO = 0, BI_RHO_OLD = 1, BI_SIGMA = 2, BI_ALPHA = 3, BI_OMEGA = 4, BI_TS = 5, BI_TT = 6 /* triple 6..8 */,
       BI_ALPHA_OLD = 15 /* alpha of the last completed full step: a run-ahead half overwrites BI_ALPHA */ };
__global__ void k_set_scalar(double* dst, double v) { *dst = v; }
// p_out = r + beta (p - omega v),  beta = (rho/rho_old)(alpha/omega)          (:121-125)
__global__ __launch_bounds__(KK_TPB) void k_bicg_p(double* __restrict__ p_out, const double* __restrict__ p,
                                                   const double* __restrict__ r, const double* __restrict__ v, int64_t ld,
                                                   int64_t rpb, const double* __restrict__ sc) {
    const double omega = sc[BI_OMEGA];
    const double beta = (sc[BI_RHO] / sc[BI_RHO_OLD]) * (sc[BI_ALPHA_OLD] / omega);
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 pv = ld2(p + i), rv = ld2(r + i), vv = ld2(v + i);
        pv.x = fma(-omega, vv.x, pv.x); pv.y = fma(-omega, vv.y, pv.y);
        pv.x = fma(beta, pv.x, rv.x); pv.y = fma(beta, pv.y, rv.y);
        st2(p_out + i, pv);
    }
}
// alpha = rho/sigma ; s = r - alpha v ; partial |s|^2                          (:130-139)
__global__ __launch_bounds__(KK_TPB) void k_bicg_s(double* __restrict__ s, const double* __restrict__ r,
                                                   const double* __restrict__ v, int64_t ld, int64_t rpb,
                                                   double* __restrict__ sc, double* __restrict__ part) {
    __shared__ double sm[4];
    const double alpha = sc[BI_RHO] / sc[BI_SIGMA];
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[BI_ALPHA] = alpha;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double acc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 rv = ld2(r + i), vv = ld2(v + i);
        rv.x = fma(-alpha, vv.x, rv.x); rv.y = fma(-alpha, vv.y, rv.y);
        st2(s + i, rv);
        acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
    }
    double t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// omega = <t,s>/<t,t> ; x += alpha p + omega s ; r = s - omega t ; partials |r|^2 and <r_shadow, r>   (:160-169,120)
__global__ __launch_bounds__(KK_TPB) void k_bicg_xr(double* __restrict__ x, const double* __restrict__ p,
                                                    const double* __restrict__ s, const double* __restrict__ t,
                                                    double* __restrict__ r, const double* __restrict__ rs, int64_t ld,
                                                    int64_t rpb, double* __restrict__ sc, double* __restrict__ part_n,
                                                    double* __restrict__ part_d) {
    __shared__ double sm[4];
    const double alpha = sc[BI_ALPHA];
    const double omega = sc[BI_TS] / sc[BI_TT];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[BI_OMEGA] = omega;
        sc[BI_ALPHA_OLD] = alpha;
        sc[BI_RHO_OLD] = sc[BI_RHO];   // the finalize of <r_shadow, r> (next kernel in the stream) overwrites BI_RHO
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double nacc = 0, dacc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 xv = ld2(x + i), pv = ld2(p + i), sv = ld2(s + i), tv = ld2(t + i), zv = ld2(rs + i);
        xv.x = fma(alpha, pv.x, xv.x); xv.y = fma(alpha, pv.y, xv.y);
        xv.x = fma(omega, sv.x, xv.x); xv.y = fma(omega, sv.y, xv.y);
        sv.x = fma(-omega, tv.x, sv.x); sv.y = fma(-omega, tv.y, sv.y);
        st2(x + i, xv); st2(r + i, sv);
        nacc = fma(sv.x, sv.x, nacc); nacc = fma(sv.y, sv.y, nacc);
        dacc = fma(zv.x, sv.x, dacc); dacc = fma(zv.y, sv.y, dacc);
    }
    double a = block_sum(nacc, sm);
    if (threadIdx.x == 0) part_n[blockIdx.x] = a;
    __syncthreads();
    double b = block_sum(dacc, sm);
    if (threadIdx.x == 0) part_d[blockIdx.x] = b;
}

// ---- LSMR (lssolve/lsmr.jl:61-110) vector updates, fused
// Ah = Av - c Ah ; u = Av - alpha u ; partial |u|^2                              (:64-68)
__global__ __launch_bounds__(KK_TPB) void k_lsmr_u(const double* __restrict__ av, double* __restrict__ ah,
                                                   double* __restrict__ u, int64_t ld, int64_t rpb, double c, double alpha,
                                                   double* __restrict__ part) {
    __shared__ double sm[4];
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double acc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        const d2 a = ld2(av + i);
        d2 h = ld2(ah + i), uv = ld2(u + i);
        h.x = fma(-c, h.x, a.x); h.y = fma(-c, h.y, a.y);
        uv.x = fma(-alpha, uv.x, a.x); uv.y = fma(-alpha, uv.y, a.y);
        st2(ah + i, h); st2(u + i, uv);
        acc = fma(uv.x, uv.x, acc); acc = fma(uv.y, uv.y, acc);
    }
    double t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// hbar = h - c1 hbar ; x += c2 hbar ; [h = v - c3 h when v != nullptr]           (:121-128)
__global__ __launch_bounds__(KK_TPB) void k_lsmr_hx(double* __restrict__ h, double* __restrict__ hbar, double* __restrict__ x,
                                                    const double* __restrict__ v, int64_t ld, int64_t rpb, double c1,
                                                    double c2, double c3) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 hv = ld2(h + i), hb = ld2(hbar + i), xv = ld2(x + i);
        hb.x = fma(-c1, hb.x, hv.x); hb.y = fma(-c1, hb.y, hv.y);
        xv.x = fma(c2, hb.x, xv.x); xv.y = fma(c2, hb.y, xv.y);
        st2(hbar + i, hb); st2(x + i, xv);
        if (v) {
            const d2 vv = ld2(v + i);
            hv.x = fma(-c3, hv.x, vv.x); hv.y = fma(-c3, hv.y, vv.y);
            st2(h + i, hv);
        }
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
__global__ __launch_bounds__(KK_TPB) void k_lowsync_solve(const double* __restrict__ p, const double* __restrict__ g_ride,
                                                          double* L, int cap, int m, int newest,
                                                          const double* __restrict__ a0, double* __restrict__ coef_out,
                                                          double* __restrict__ s_out) {
    __shared__ double rhs[KK_MAX_M];
    const int i = threadIdx.x;
    if (g_ride && i < m - 1) L[(int64_t)newest * cap + i] = g_ride[i];
    if (i < m) rhs[i] = p[i];
    __syncthreads();
    for (int j = 0; j < m - 1; ++j) {
        const double sj = rhs[j];
        if (i > j && i < m) {
            const double lij = (g_ride && i == newest) ? g_ride[j] : L[(int64_t)i * cap + j];
            rhs[i] = fma(-lij, sj, rhs[i]);
        }
        __syncthreads();
    }
    if (i < m) {
        const double s = rhs[i];
        s_out[i] = s;
        coef_out[i] = (a0 && i == m - 1) ? s + *a0 : s;
    }
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

// ------------------------------------------------------------------------------------------
// SpMV.  ELL (column-major, padded to `width`) with 2 rows per lane for regular matrices
// (stencils); CSR with L lanes per row (L = 64 is row-per-wavefront) otherwise.
// Fused epilogue (Lanczos three-term tail, lanczos.jl:297-310):
//   ax  = xs * sum_k val*x[col]            (xs: optional device scalar, e.g. 1/alpha in GKL)
//   y   = a1*ax + a0*x[row] - bprev*vprev[row]
//   dot = <x, ax> (mode 1)  or <x, y> (mode 2)    nrm2 = |y|^2
// Column indices >= n_local address the ghost buffer (row-sharded operators).
// ------------------------------------------------------------------------------------------
// streamed-once matrix entries (SELL): scalar non-temporal loads, so that they do not evict the gathered slice of x
__device__ __forceinline__ int ldc(const int32_t* p) {
#ifndef KK_NO_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ double ldv(const double* p) {
#ifndef KK_NO_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct spmv_epi {
    double a1, a0, bprev;
    const double* xs_dev;
    const double* bprev_dev;
    const double* vprev;
    int dot_mode;
    int want_nrm;
    int64_t n_local;  // < 0: no ghost
    const double* ghost;
    const double* dvec;  // dot_mode 3: <dvec, y>
    int acc;             // column-tiled apply: 0 = whole matrix, 1 = first tile (y = raw sums), 2 = middle tile
                         // (y += raw sums), 3 = last tile (sum = y + raw, then the epilogue)
};

__device__ __forceinline__ double xload(const double* __restrict__ x, const spmv_epi& e, int c) {
    if (e.n_local >= 0 && c >= e.n_local) return e.ghost[c - e.n_local];
    return x[c];
}

__global__ __launch_bounds__(KK_TPB) void k_spmv_ell(const int32_t* __restrict__ ecol, const double* __restrict__ eval,
                                                     int64_t ell_ld, int width, int64_t nrows,
                                                     const double* __restrict__ x, double* __restrict__ y, spmv_epi e,
                                                     int nb_logical, double* __restrict__ part_dot,
                                                     double* __restrict__ part_nrm) {
    __shared__ double sm[4];
    // XCD banding: block b runs on XCD b & 7 and walks the band [xcd*per, (xcd+1)*per) of logical
    // 512-row chunks with stride nbx, so the blocks resident on one XCD sweep a contiguous row
    // window together and stencil neighbours (+-nx rows) are L2 hits of the same XCD.
    const int per = (nb_logical + 7) >> 3;
    const int nbx = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    double dacc = 0, nacc = 0;
    const double xs = e.xs_dev ? *e.xs_dev : 1.0;
    const double bp = e.vprev ? (e.bprev_dev ? *e.bprev_dev : e.bprev) : 0.0;
    for (int c = blockIdx.x >> 3; c < per; c += nbx) {
        const int lb = xcd * per + c;
        if (lb >= nb_logical) break;
        const int64_t row = ((int64_t)lb * KK_TPB + threadIdx.x) * 2;
        if (row < nrows) {  // ell_ld is even and >= nrows; pad entries have val 0, col 0
            double s0 = 0, s1 = 0;
            const int32_t* cp = ecol + row;
            const double* vp = eval + row;
            for (int k = 0; k < width; ++k) {
                const int2 cc = ldi2s(cp + (int64_t)k * ell_ld);
                const d2 v = ld2s(vp + (int64_t)k * ell_ld);
                s0 = fma(v.x, xload(x, e, cc.x), s0);
                s1 = fma(v.y, xload(x, e, cc.y), s1);
            }
            s0 *= xs; s1 *= xs;
            d2 out{e.a1 * s0, e.a1 * s1};
            d2 xv{0.0, 0.0};
            if (e.a0 != 0.0 || e.dot_mode == 1 || e.dot_mode == 2) {
                xv = ld2(x + row);
                xv.x *= xs; xv.y *= xs;
            }
            if (e.a0 != 0.0) { out.x = fma(e.a0, xv.x, out.x); out.y = fma(e.a0, xv.y, out.y); }
            if (e.dot_mode == 1) { dacc = fma(xv.x, out.x, dacc); dacc = fma(xv.y, out.y, dacc); }
            if (e.vprev) {
                const d2 p = ld2(e.vprev + row);
                out.x = fma(-bp, p.x, out.x); out.y = fma(-bp, p.y, out.y);
            }
            if (row + 1 >= nrows) out.y = 0.0;  // odd nrows: keep the pad row zero
            if (e.dot_mode == 2) { dacc = fma(xv.x, out.x, dacc); dacc = fma(xv.y, out.y, dacc); }
            if (e.dot_mode == 3) { const d2 z = ld2(e.dvec + row); dacc = fma(z.x, out.x, dacc); dacc = fma(z.y, out.y, dacc); }
            if (e.want_nrm) { nacc = fma(out.x, out.x, nacc); nacc = fma(out.y, out.y, nacc); }
            st2(y + row, out);
        }
    }
    if (e.dot_mode) {
        double t = block_sum(dacc, sm);
        if (threadIdx.x == 0) part_dot[blockIdx.x] = t;
    }
    if (e.want_nrm) {
        double t = block_sum(nacc, sm);
        if (threadIdx.x == 0) part_nrm[blockIdx.x] = t;
    }
}

template <int L>
__global__ __launch_bounds__(KK_TPB) void k_spmv_csr(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                     const double* __restrict__ val, int64_t nrows,
                                                     const double* __restrict__ x, double* __restrict__ y, spmv_epi e,
                                                     double* __restrict__ part_dot, double* __restrict__ part_nrm) {
    __shared__ double sm[4];
    constexpr int RPB = KK_TPB / L;  // rows per block iteration
    const int sub = threadIdx.x % L, rl = threadIdx.x / L;
    double dacc = 0, nacc = 0;
    const double xs = e.xs_dev ? *e.xs_dev : 1.0;
    for (int64_t row = (int64_t)blockIdx.x * RPB + rl; row < nrows; row += (int64_t)gridDim.x * RPB) {
        const int b = rowptr[row], en = rowptr[row + 1];
        double s = 0;
        for (int k = b + sub; k < en; k += L) s = fma(val[k], xload(x, e, colind[k]), s);
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
        if (sub == 0) {
            s *= xs;
            double out = e.a1 * s;
            double xv = 0;
            if (e.a0 != 0.0 || e.dot_mode == 1 || e.dot_mode == 2) xv = x[row] * xs;
            if (e.a0 != 0.0) out = fma(e.a0, xv, out);
            if (e.dot_mode == 1) dacc = fma(xv, out, dacc);
            if (e.vprev) {
                const double bp = e.bprev_dev ? *e.bprev_dev : e.bprev;
                out = fma(-bp, e.vprev[row], out);
            }
            if (e.dot_mode == 2) dacc = fma(xv, out, dacc);
            if (e.dot_mode == 3) dacc = fma(e.dvec[row], out, dacc);
            if (e.want_nrm) nacc = fma(out, out, nacc);
            y[row] = out;
        }
    }
    if (e.dot_mode) {
        double t = block_sum(dacc, sm);
        if (threadIdx.x == 0) part_dot[blockIdx.x] = t;
    }
    if (e.want_nrm) {
        double t = block_sum(nacc, sm);
        if (threadIdx.x == 0) part_nrm[blockIdx.x] = t;
    }
}

// SELL-64-sigma SpMV for irregular matrices (e.g. A' of the rectangular GKL map): one wavefront per
// chunk of 64 rows of similar length (rows are sorted by length inside windows of sigma rows on the
// host), data stored [chunk][k][lane] so every load is one contiguous 512 B (values) / 256 B
// (columns) wave transaction and padding is limited to the spread inside one chunk.
__global__ __launch_bounds__(KK_TPB) void k_spmv_sell(const int64_t* __restrict__ coff, const int32_t* __restrict__ perm,
                                                      const int32_t* __restrict__ scol, const double* __restrict__ sval,
                                                      int64_t nchunks, const double* __restrict__ x, double* __restrict__ y,
                                                      spmv_epi e, double* __restrict__ part_dot, double* __restrict__ part_nrm) {
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double dacc = 0, nacc = 0;
    const double xs = e.xs_dev ? *e.xs_dev : 1.0;
    const double bp = e.vprev ? (e.bprev_dev ? *e.bprev_dev : e.bprev) : 0.0;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < nchunks; c += (int64_t)gridDim.x * 4) {
        const int64_t off = coff[c];
        const int w = (int)((coff[c + 1] - off) >> 6);
        const int32_t* cp = scol + off + lane;
        const double* vp = sval + off + lane;
        double s0 = 0, s1 = 0;
        int k = 0;
        for (; k + 4 <= w; k += 4) {
            const int c0 = ldc(cp + (k + 0) * 64), c1 = ldc(cp + (k + 1) * 64), c2 = ldc(cp + (k + 2) * 64), c3 = ldc(cp + (k + 3) * 64);
            const double v0 = ldv(vp + (k + 0) * 64), v1 = ldv(vp + (k + 1) * 64), v2 = ldv(vp + (k + 2) * 64), v3 = ldv(vp + (k + 3) * 64);
            s0 = fma(v0, xload(x, e, c0), s0);
            s1 = fma(v1, xload(x, e, c1), s1);
            s0 = fma(v2, xload(x, e, c2), s0);
            s1 = fma(v3, xload(x, e, c3), s1);
        }
        for (; k < w; ++k) s0 = fma(ldv(vp + k * 64), xload(x, e, ldc(cp + k * 64)), s0);
        const int row = perm[c * 64 + lane];
        if (row >= 0) {
            double raw = s0 + s1;
            if (e.acc == 1) { y[row] = raw; continue; }
            if (e.acc == 2) { y[row] += raw; continue; }
            if (e.acc == 3) raw += y[row];
            const double s = raw * xs;
            double out = e.a1 * s;
            double xv = 0;
            if (e.a0 != 0.0 || e.dot_mode == 1 || e.dot_mode == 2) xv = x[row] * xs;
            if (e.a0 != 0.0) out = fma(e.a0, xv, out);
            if (e.dot_mode == 1) dacc = fma(xv, out, dacc);
            if (e.vprev) out = fma(-bp, e.vprev[row], out);
            if (e.dot_mode == 2) dacc = fma(xv, out, dacc);
            if (e.dot_mode == 3) dacc = fma(e.dvec[row], out, dacc);
            if (e.want_nrm) nacc = fma(out, out, nacc);
            y[row] = out;
        }
    }
    if (e.dot_mode) {
        double t = block_sum(dacc, sm);
        if (threadIdx.x == 0) part_dot[blockIdx.x] = t;
    }
    if (e.want_nrm) {
        double t = block_sum(nacc, sm);
        if (threadIdx.x == 0) part_nrm[blockIdx.x] = t;
    }
}

// Window variant of the SELL kernel for sigma = 256 = the rows of one thread block (used by the column tiles, where
// a row has only a handful of entries per tile and the row-sorted result order would turn the y update into
// scattered 8-byte accesses): the four waves compute the raw sums of the four chunks of a 256-row window in the sorted
// order, park them in LDS under the row's position in the window, and after a barrier thread t finishes row
// base + t -- y, x, v_prev and the inner-product operands are all read and written coalesced.
__global__ __launch_bounds__(KK_TPB) void k_spmv_sellw(const int64_t* __restrict__ coff, const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ scol, const double* __restrict__ sval,
                                                       int64_t nchunks, int64_t nrows, const double* __restrict__ x,
                                                       double* __restrict__ y, spmv_epi e, double* __restrict__ part_dot,
                                                       double* __restrict__ part_nrm) {
    __shared__ double res[KK_TPB];
    __shared__ double sm[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double dacc = 0, nacc = 0;
    const double xs = e.xs_dev ? *e.xs_dev : 1.0;
    const double bp = e.vprev ? (e.bprev_dev ? *e.bprev_dev : e.bprev) : 0.0;
    const int64_t nwin = (nchunks + 3) >> 2;
    // the chunk descriptor of the NEXT window is fetched while the current one is processed, and the old y of the
    // accumulating tiles is requested before the gathers: two of the four dependent memory round trips per window go
    int64_t win = blockIdx.x;
    int64_t off = 0, offn = 0;
    int32_t prow = -1;
    if (win < nwin && win * 4 + wave < nchunks) {
        off = coff[win * 4 + wave]; offn = coff[win * 4 + wave + 1];
        prow = perm[(win * 4 + wave) * 64 + lane];
    }
    for (; win < nwin; win += gridDim.x) {
        const int64_t c = win * 4 + wave;
        const int64_t row = win * KK_TPB + tid;
        double yold = 0.0;
        if (e.acc >= 2 && row < nrows) yold = y[row];
        const int64_t wnext = win + gridDim.x;
        int64_t off2 = 0, offn2 = 0;
        int32_t prow2 = -1;
        if (wnext < nwin && wnext * 4 + wave < nchunks) {
            off2 = coff[wnext * 4 + wave]; offn2 = coff[wnext * 4 + wave + 1];
            prow2 = perm[(wnext * 4 + wave) * 64 + lane];
        }
        if (c < nchunks) {
            const int w = (int)((offn - off) >> 6);
            const int32_t* cp = scol + off + lane;
            const double* vp = sval + off + lane;
            double s0 = 0, s1 = 0;
            int k = 0;
            for (; k + 4 <= w; k += 4) {
                const int c0 = ldc(cp + (k + 0) * 64), c1 = ldc(cp + (k + 1) * 64), c2 = ldc(cp + (k + 2) * 64), c3 = ldc(cp + (k + 3) * 64);
                const double v0 = ldv(vp + (k + 0) * 64), v1 = ldv(vp + (k + 1) * 64), v2 = ldv(vp + (k + 2) * 64), v3 = ldv(vp + (k + 3) * 64);
                s0 = fma(v0, xload(x, e, c0), s0);
                s1 = fma(v1, xload(x, e, c1), s1);
                s0 = fma(v2, xload(x, e, c2), s0);
                s1 = fma(v3, xload(x, e, c3), s1);
            }
            for (; k < w; ++k) s0 = fma(ldv(vp + k * 64), xload(x, e, ldc(cp + k * 64)), s0);
            if (prow >= 0) res[prow - win * KK_TPB] = s0 + s1;
        }
        __syncthreads();
        if (row < nrows) {
            double raw = res[tid];
            if (e.acc == 1) y[row] = raw;
            else if (e.acc == 2) y[row] = yold + raw;
            else {
                if (e.acc == 3) raw += yold;
                const double s = raw * xs;
                double out = e.a1 * s;
                double xv = 0;
                if (e.a0 != 0.0 || e.dot_mode == 1 || e.dot_mode == 2) xv = x[row] * xs;
                if (e.a0 != 0.0) out = fma(e.a0, xv, out);
                if (e.dot_mode == 1) dacc = fma(xv, out, dacc);
                if (e.vprev) out = fma(-bp, e.vprev[row], out);
                if (e.dot_mode == 2) dacc = fma(xv, out, dacc);
                if (e.dot_mode == 3) dacc = fma(e.dvec[row], out, dacc);
                if (e.want_nrm) nacc = fma(out, out, nacc);
                y[row] = out;
            }
        }
        __syncthreads();
        off = off2; offn = offn2; prow = prow2;
    }
    if (e.dot_mode) {
        double t = block_sum(dacc, sm);
        if (threadIdx.x == 0) part_dot[blockIdx.x] = t;
    }
    if (e.want_nrm) {
        double t = block_sum(nacc, sm);
        if (threadIdx.x == 0) part_nrm[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// restart-time kernels (per restart, not per iteration)
// ------------------------------------------------------------------------------------------
// basistransform! (orthonormal.jl:291-354): V[:, 0:n] <- V[:, 0:m] * U (m x n, column-major in
// device memory).  Row-local, so it is done in place: a block stages a 64-row x m tile in LDS,
// then each thread produces outputs for (row, 4 columns at a time).
#define BT_ROWS 64
__global__ __launch_bounds__(KK_TPB) void k_basistransform(double* __restrict__ V, int64_t ld, int m, int n,
                                                           const double* __restrict__ U) {
    extern __shared__ __attribute__((aligned(16))) double tile[];  // [m][BT_ROWS + 1]
    const int tid = threadIdx.x;
    const int TS = BT_ROWS + 1;
    for (int64_t rb = (int64_t)blockIdx.x * BT_ROWS; rb < ld; rb += (int64_t)gridDim.x * BT_ROWS) {
        for (int idx = tid; idx < m * BT_ROWS; idx += KK_TPB) {
            const int i = idx / BT_ROWS, r = idx % BT_ROWS;
            tile[i * TS + r] = V[(int64_t)i * ld + rb + r];
        }
        __syncthreads();
        const int r = tid % BT_ROWS, jg = tid / BT_ROWS;  // 4 column groups
        for (int j = jg; j < n; j += 4) {
            const double* Uj = U + (int64_t)j * m;
            double a = 0;
            for (int i = 0; i < m; ++i) a = fma(tile[i * TS + r], Uj[i], a);
            V[(int64_t)j * ld + rb + r] = a;
        }
        __syncthreads();
    }
}

// basistransform! as a tall-skinny GEMM on v_mfma_f64_16x16x4_f64 (thick restart,
// eigsolve/lanczos.jl:109): out[rows, 0:n] = V[rows, 0:m] * U, in place.
//   D[i][j] += sum_k A[i][k] B[k][j];  k = 4 basis columns per MFMA, j = 16 output columns per tile.
//   Lane (i = l&15, kq = l>>4) loads 4 CONSECUTIVE rows R0+4i..R0+4i+3 of column C+kq (32 B);
//   MFMA t uses element t, i.e. its 16 "rows" are R0 + 4i + t.  The result regs of lane l for
//   fixed (tile, r) and t = 0..3 are 4 consecutive rows R0 + 4((l>>4)+4r) + t -> 32 B stores.
// A wave owns 64-row chunks: it reads all m columns of a chunk before writing the n outputs of
// the same rows, so the transform is safely in place.  U sits in LDS (B operand).
template <int NJ>
__global__ __launch_bounds__(KK_TPB) void k_basistransform_mfma(double* V, int64_t ld, int m, int n, int npad,
                                                                const double* __restrict__ U, int64_t rpb) {
    extern __shared__ __attribute__((aligned(16))) double Us[];  // [m4][npad], zero padded
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m4 = (m + 3) & ~3;
    for (int idx = tid; idx < m4 * npad; idx += KK_TPB) {
        const int k = idx / npad, j = idx % npad;
        Us[idx] = (k < m && j < n) ? U[(int64_t)j * m + k] : 0.0;
    }
    __syncthreads();
    const int i = lane & 15, kq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t R0 = r0 + wave * 64; R0 < r1; R0 += 256) {
        v4d acc[NJ][4];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[jt][t] = v4d{0.0, 0.0, 0.0, 0.0};
        const double* vin = V + R0 + 4 * i;
        for (int C = 0; C < m4; C += 4) {
            const int col = C + kq;
            double a[4];
            if (col < m) {
                const d2 x0 = ld2s(vin + (int64_t)col * ld), x1 = ld2s(vin + (int64_t)col * ld + 2);
                a[0] = x0.x; a[1] = x0.y; a[2] = x1.x; a[3] = x1.y;
            } else {
                a[0] = a[1] = a[2] = a[3] = 0.0;
            }
            const double* urow = Us + (C + kq) * npad + i;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const double bv = urow[jt * 16];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[jt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bv, acc[jt][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int col = jt * 16 + i;
            if (col < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* o = V + (int64_t)col * ld + R0 + 4 * (kq + 4 * r);
                    st2(o, d2{acc[jt][0][r], acc[jt][1][r]});
                    st2(o + 2, d2{acc[jt][2][r], acc[jt][3][r]});
                }
            }
        }
    }
}

// rmul!(b, G::Givens) (dense/givens.jl:20-36): (q1,q2) <- (c q1 - s q2, s q1 + c q2)
__global__ __launch_bounds__(KK_TPB) void k_givens(double* __restrict__ q1, double* __restrict__ q2, int64_t ld,
                                                   int64_t rpb, double c, double s) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 a = ld2(q1 + r), b = ld2(q2 + r), o1, o2;
        o1.x = c * a.x - s * b.x; o1.y = c * a.y - s * b.y;
        o2.x = s * a.x + c * b.x; o2.y = s * a.y + c * b.y;
        st2(q1 + r, o1); st2(q2 + r, o2);
    }
}

// rmul!(b, H::Householder) (dense/reflector.jl:143-154), row-local and fused:
//   t = sum_j V[row,j] v[j];  V[row,j] -= beta * t * v[j]
// two sweeps over the m columns of the row tile; the second sweep hits L2.
__global__ __launch_bounds__(KK_TPB) void k_householder(double* __restrict__ V, int64_t ld, int m, kk_coef hv,
                                                        double beta, int64_t rpb) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 t{0.0, 0.0};
        for (int j = 0; j < m; ++j) {
            const d2 x = ld2(V + (int64_t)j * ld + r);
            t.x = fma(x.x, hv.v[j], t.x); t.y = fma(x.y, hv.v[j], t.y);
        }
        t.x *= beta; t.y *= beta;
        for (int j = 0; j < m; ++j) {
            d2 x = ld2(V + (int64_t)j * ld + r);
            x.x = fma(-t.x, hv.v[j], x.x); x.y = fma(-t.y, hv.v[j], x.y);
            st2(V + (int64_t)j * ld + r, x);
        }
    }
}

// rank1update! (orthonormal.jl:210-275): V_j = beta*V_j + alpha * y * x[j]
__global__ __launch_bounds__(KK_TPB) void k_rank1(double* __restrict__ V, int64_t ld, int m,
                                                  const double* __restrict__ y, kk_coef xc, double alpha, double beta,
                                                  int64_t rpb) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        const d2 yv = ld2(y + r);
        for (int j = 0; j < m; ++j) {
            const double a = alpha * xc.v[j];
            d2 x;
            if (beta == 0.0) { x.x = a * yv.x; x.y = a * yv.y; }
            else {
                x = ld2(V + (int64_t)j * ld + r);
                x.x = fma(a, yv.x, beta * x.x); x.y = fma(a, yv.y, beta * x.y);
            }
            st2(V + (int64_t)j * ld + r, x);
        }
    }
}

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

// SpMM on ELL: Y[:, j] = A X[:, j], j < nb <= NB (apply(f, ::Block), blocklanczos.jl:39): the matrix
// is streamed once for the whole block instead of once per vector.
template <int NB>
__global__ __launch_bounds__(KK_TPB) void k_spmm_ell(const int32_t* __restrict__ ecol, const double* __restrict__ eval,
                                                     int64_t ell_ld, int width, int64_t nrows,
                                                     const double* __restrict__ X, int64_t ldx, double* __restrict__ Y,
                                                     int64_t ldy, int nb, int nb_logical) {
    const int per = (nb_logical + 7) >> 3;
    const int nbx = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    for (int cblk = blockIdx.x >> 3; cblk < per; cblk += nbx) {
        const int lb = xcd * per + cblk;
        if (lb >= nb_logical) break;
        const int64_t row = ((int64_t)lb * KK_TPB + threadIdx.x) * 2;
        if (row >= nrows) continue;
        d2 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = d2{0.0, 0.0};
        for (int k = 0; k < width; ++k) {
            const int2 cc = ldi2s(ecol + (int64_t)k * ell_ld + row);
            const d2 v = ld2s(eval + (int64_t)k * ell_ld + row);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nb) {
                    acc[j].x = fma(v.x, X[(int64_t)j * ldx + cc.x], acc[j].x);
                    acc[j].y = fma(v.y, X[(int64_t)j * ldx + cc.y], acc[j].y);
                }
            }
        }
        const bool last_odd = (row + 1 >= nrows);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < nb) {
                if (last_odd) acc[j].y = 0.0;
                st2(Y + (int64_t)j * ldy + row, acc[j]);
            }
        }
    }
}

// ==========================================================================================
// host-side launchers
// ==========================================================================================
static inline double* part_row(kk_ctx ctx, int row) { return ctx->partials + (int64_t)row * KK_MAX_BLOCKS; }
#define PART_SCAL_A (2 * KK_MAX_M)      // partial rows used by scalar reductions
#define PART_SCAL_B (2 * KK_MAX_M + 1)

static int finalize_scalar(kk_ctx ctx, int part_row_idx, int n, double* out, bool with_sqrt) {
    const bool sharded = ctx->allreduce != nullptr;
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

int kk_launch_spmv(kk_ctx ctx, const kk_sparse_dev& M, const double* x, double* y, int64_t ld_y_rows,
                   const kk_spmv_fuse& f) {
    (void)ld_y_rows;
    if (M.halo) {  // row-sharded operator: let the caller fill the ghost buffer from x (P2P on this stream)
        const int st = M.halo(M.halo_user, x);
        if (st != 0) { kk_set_error("halo hook failed with status %d", st); return KK_ERR_INVALID; }
    }
    spmv_epi e;
    e.a1 = f.a1; e.a0 = f.a0; e.bprev = f.bprev;
    e.xs_dev = f.xscale_dev; e.bprev_dev = f.bprev_dev; e.vprev = f.vprev;
    e.dot_mode = f.dot_mode; e.want_nrm = f.nrm_out ? 1 : 0;
    e.n_local = M.n_ghost > 0 ? M.n_local : -1;
    e.ghost = M.ghost;
    e.dvec = f.dot_vec;
    e.acc = 0;
    double* pd = part_row(ctx, PART_SCAL_A);
    double* pn = part_row(ctx, PART_SCAL_B);
    int nblk = 0;
    std::unique_ptr<kk_prof_scope> ps(new kk_prof_scope(ctx, M.format == 0 ? "k_spmv_ell" : (M.format >= 2 ? "k_spmv_sell" : "k_spmv_csr")));
    if (M.format == 2) {
        nblk = (int)std::min<int64_t>((M.sell_nchunks + 3) / 4, (int64_t)ctx->num_cus * 16);
        if (nblk < 1) nblk = 1;
        hipLaunchKernelGGL(k_spmv_sell, dim3(nblk), dim3(KK_TPB), 0, ctx->stream, M.sell_off, M.sell_perm, M.sell_col, M.sell_val,
                           M.sell_nchunks, x, y, e, pd, pn);
    } else if (M.format == 3) {
        // column tiles one after the other: tile t gathers from the L2-resident slice [t, t+1) * tile_cols of x and
        // accumulates into y; the epilogue (scaling, a0 x, - beta v_prev, inner products) runs with the last tile
        if (f.vprev == y) { ps.reset(); kk_set_error("tiled spmv: v_prev must not alias y"); return KK_ERR_INVALID; }
        for (int t = 0; t < M.ntiles; ++t) {
            const kk_sparse_dev& S = M.tiles[t];
            spmv_epi et = e;
            et.acc = M.ntiles == 1 ? 0 : (t == 0 ? 1 : (t == M.ntiles - 1 ? 3 : 2));
            if (et.acc == 1 || et.acc == 2) { et.dot_mode = 0; et.want_nrm = 0; }
            nblk = (int)std::min<int64_t>((S.sell_nchunks + 3) / 4, (int64_t)ctx->num_cus * 16);
            if (nblk < 1) nblk = 1;
            hipLaunchKernelGGL(k_spmv_sellw, dim3(nblk), dim3(KK_TPB), 0, ctx->stream, S.sell_off, S.sell_perm, S.sell_col, S.sell_val,
                               S.sell_nchunks, S.nrows, x, y, et, pd, pn);
        }
    } else if (M.format == 0) {
        const int nb_logical = (int)((M.nrows + 2 * KK_TPB - 1) / (2 * KK_TPB));
        const int per = (nb_logical + 7) / 8;
        const int nbx = std::min(per, KK_MAX_BLOCKS / 8);
        nblk = nbx * 8;
        hipLaunchKernelGGL(k_spmv_ell, dim3(nblk), dim3(KK_TPB), 0, ctx->stream, M.ell_col, M.ell_val, M.ell_ld, M.width,
                           M.nrows, x, y, e, nb_logical, pd, pn);
    } else {
        const int L = M.lanes_per_row;
        const int rpb = KK_TPB / L;
        int64_t want = (M.nrows + rpb - 1) / rpb;
        nblk = (int)std::min<int64_t>(want, (int64_t)ctx->num_cus * 16);
        if (nblk < 1) nblk = 1;
        dim3 g(nblk), b(KK_TPB);
#define CSR_CASE(LL) case LL: hipLaunchKernelGGL((k_spmv_csr<LL>), g, b, 0, ctx->stream, M.rowptr, M.colind, M.val, M.nrows, x, y, e, pd, pn); break;
        switch (L) {
            CSR_CASE(2) CSR_CASE(4) CSR_CASE(8) CSR_CASE(16) CSR_CASE(32) CSR_CASE(64)
            default: ps.reset(); kk_set_error("bad lanes_per_row %d", L); return KK_ERR_INVALID;
        }
#undef CSR_CASE
    }
    ps.reset();
    KK_HIP(hipGetLastError());
    if (nblk > KK_MAX_BLOCKS && (f.dot_mode || f.nrm_out)) {
        kk_set_error("spmv grid %d exceeds partial buffer", nblk);
        return KK_ERR_INVALID;
    }
    if (f.dot_mode) KK_TRY(finalize_scalar(ctx, PART_SCAL_A, nblk, f.dot_out, false));
    if (f.nrm_out) KK_TRY(finalize_scalar(ctx, PART_SCAL_B, nblk, f.nrm_out, true));
    return KK_OK;
}

int kk_launch_basistransform(kk_ctx ctx, double* V, int64_t ld, int m, int n, const double* U_dev) {
    kk_prof_scope ps(ctx, "k_basistransform");
    const int nj = (n + 15) / 16;
    if (nj <= 6 && !getenv("KK_BASISTRANSFORM_LDS")) {
        int npad = (n + 15) / 16 * 16;
        while (npad % 32 != 16) npad += 16;   // B-operand rows land on disjoint LDS banks
        const int m4 = (m + 3) & ~3;
        const size_t shm = (size_t)m4 * npad * sizeof(double);
        kk_part p = kk_partition(ctx, ld);
        dim3 g(p.nblk), b(KK_TPB);
#define BT_CASE(NJT)                                                                                                   \
        {                                                                                                              \
            KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_basistransform_mfma<NJT>),                      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                         \
            hipLaunchKernelGGL((k_basistransform_mfma<NJT>), g, b, shm, ctx->stream, V, ld, m, n, npad, U_dev, p.rpb);  \
        }
        if (shm <= 160 * 1024 - 256) {
            switch (nj) {
                case 1: BT_CASE(1) break;
                case 2: BT_CASE(2) break;
                case 3: BT_CASE(3) break;
                case 4: BT_CASE(4) break;
                case 5: BT_CASE(5) break;
                default: BT_CASE(6) break;
            }
            KK_HIP(hipGetLastError());
            return KK_OK;
        }
#undef BT_CASE
    }
    const size_t shm = (size_t)m * (BT_ROWS + 1) * sizeof(double);
    int nb = (int)std::min<int64_t>(ld / BT_ROWS, (int64_t)ctx->num_cus * 8);
    if (nb < 1) nb = 1;
    if (shm > 64 * 1024) {
        KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_basistransform),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    }
    hipLaunchKernelGGL(k_basistransform, dim3(nb), dim3(KK_TPB), shm, ctx->stream, V, ld, m, n, U_dev);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_givens(kk_ctx ctx, double* q1, double* q2, int64_t ld, double c, double s) {
    kk_prof_scope ps(ctx, "k_givens");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_givens, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, q1, q2, ld, p.rpb, c, s);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_householder(kk_ctx ctx, double* V, int64_t ld, int m, const kk_coef* v, double beta) {
    kk_prof_scope ps(ctx, "k_householder");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_householder, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, V, ld, m, *v, beta, p.rpb);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_rank1(kk_ctx ctx, double* V, int64_t ld, int m, const double* y, const kk_coef* x, double alpha,
                    double beta) {
    kk_prof_scope ps(ctx, "k_rank1");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_rank1, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, V, ld, m, y, *x, alpha, beta, p.rpb);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

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
        hipLaunchKernelGGL(k_finalize_project, dim3((nb + 3) / 4), dim3(KK_TPB), 0, ctx->stream, part, p.nblk, nb, norms2_dev,
                           (double*)nullptr);
        KK_HIP(hipGetLastError());
        KK_TRY(kk_allreduce(ctx, norms2_dev, nb));
    }
    return KK_OK;
}

// Y[:, j] = A X[:, j], j < nb (any nb: processed 16 / 8 / 4 columns at a time)
int kk_launch_spmm(kk_ctx ctx, const kk_sparse_dev& M, const double* X, int64_t ldx, double* Y, int64_t ldy, int nb) {
    if (M.format != 0 || M.n_ghost > 0 || M.halo) {  // CSR / ghosted operators: one SpMV per column
        for (int j = 0; j < nb; ++j) {
            kk_spmv_fuse f;
            KK_TRY(kk_launch_spmv(ctx, M, X + (int64_t)j * ldx, Y + (int64_t)j * ldy, ldy, f));
        }
        return KK_OK;
    }
    const int nb_logical = (int)((M.nrows + 2 * KK_TPB - 1) / (2 * KK_TPB));
    const int per = (nb_logical + 7) / 8;
    const int nbx = std::min(per, KK_MAX_BLOCKS / 8);
    dim3 g(nbx * 8), b(KK_TPB);
    int j0 = 0;
    while (j0 < nb) {
        const int rem = nb - j0;
        const double* x = X + (int64_t)j0 * ldx;
        double* y = Y + (int64_t)j0 * ldy;
        kk_prof_scope ps(ctx, "k_spmm_ell");
        if (rem > 8) {
            const int n = std::min(rem, 16);
            hipLaunchKernelGGL((k_spmm_ell<16>), g, b, 0, ctx->stream, M.ell_col, M.ell_val, M.ell_ld, M.width, M.nrows, x, ldx, y, ldy, n, nb_logical);
            j0 += n;
        } else if (rem > 4) {
            hipLaunchKernelGGL((k_spmm_ell<8>), g, b, 0, ctx->stream, M.ell_col, M.ell_val, M.ell_ld, M.width, M.nrows, x, ldx, y, ldy, rem, nb_logical);
            j0 += rem;
        } else {
            hipLaunchKernelGGL((k_spmm_ell<4>), g, b, 0, ctx->stream, M.ell_col, M.ell_val, M.ell_ld, M.width, M.nrows, x, ldx, y, ldy, rem, nb_logical);
            j0 += rem;
        }
    }
    KK_HIP(hipGetLastError());
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

int kk_launch_cg_update(kk_ctx ctx, double* x, const double* p, double* r, const double* q, int64_t ld, double alpha,
                        const double* pq_dev, double* nrm_out3) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_cg_update");
        hipLaunchKernelGGL(k_cg_update, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, x, p, r, q, ld, pt.rpb, alpha, pq_dev,
                           part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true);
}

int kk_launch_bicg_p(kk_ctx ctx, double* p_out, const double* p, const double* r, const double* v, int64_t ld,
                     const double* sc) {
    kk_part pt = kk_partition(ctx, ld);
    kk_prof_scope ps(ctx, "k_bicg_p");
    hipLaunchKernelGGL(k_bicg_p, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, p_out, p, r, v, ld, pt.rpb, sc);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_set_scalar(kk_ctx ctx, double* dst, double v) {
    hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, ctx->stream, dst, v);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_bicg_s(kk_ctx ctx, double* s, const double* r, const double* v, int64_t ld, double* sc, double* nrm_out3) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_bicg_s");
        hipLaunchKernelGGL(k_bicg_s, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, s, r, v, ld, pt.rpb, sc,
                           part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true);
}
int kk_launch_bicg_xr(kk_ctx ctx, double* x, const double* p, const double* s, const double* t, double* r,
                      const double* rs, int64_t ld, double* sc, double* nrm_out3, double* rho_out) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_bicg_xr");
        hipLaunchKernelGGL(k_bicg_xr, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, x, p, s, t, r, rs, ld, pt.rpb, sc,
                           part_row(ctx, PART_SCAL_A), part_row(ctx, PART_SCAL_B));
    }
    KK_HIP(hipGetLastError());
    KK_TRY(finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true));
    return finalize_scalar(ctx, PART_SCAL_B, pt.nblk, rho_out, false);
}

int kk_launch_lsmr_u(kk_ctx ctx, const double* av, double* ah, double* u, int64_t ld, double c, double alpha,
                     double* nrm_out3) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_lsmr_u");
        hipLaunchKernelGGL(k_lsmr_u, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, av, ah, u, ld, pt.rpb, c, alpha,
                           part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true);
}
int kk_launch_lsmr_hx(kk_ctx ctx, double* h, double* hbar, double* x, const double* v, int64_t ld, double c1, double c2,
                      double c3) {
    kk_part pt = kk_partition(ctx, ld);
    kk_prof_scope ps(ctx, "k_lsmr_hx");
    hipLaunchKernelGGL(k_lsmr_hx, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, h, hbar, x, v, ld, pt.rpb, c1, c2, c3);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_lowsync_solve(kk_ctx ctx, const double* p, const double* g_ride, double* L, int cap, int m, int newest,
                            const double* a0_dev, double* coef_out, double* s_out) {
    hipLaunchKernelGGL(k_lowsync_solve, dim3(1), dim3(KK_TPB), 0, ctx->stream, p, g_ride, L, cap, m, newest, a0_dev, coef_out,
                       s_out);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
