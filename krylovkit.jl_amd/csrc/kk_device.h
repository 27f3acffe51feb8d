// Device-side helpers shared by the kernel translation units of libkrylov_hip.so (16-byte loads / stores, non-temporal
// variants, wave64 DPP reductions, block reductions) and the host-side helpers of their launchers.
#pragma once
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

// ---- launcher helpers (host)
static inline double* part_row(kk_ctx ctx, int row) { return ctx->partials + (int64_t)row * KK_MAX_BLOCKS; }
#define PART_SCAL_A (2 * KK_MAX_M)      // partial rows used by scalar reductions
#define PART_SCAL_B (2 * KK_MAX_M + 1)
// out = sum of n per-block partials (row part_row_idx); with_sqrt: out[0..2] = {sum, sqrt, 1/sqrt}.  On a row-sharded
// context the sum is all-reduced across ranks through the context's hook before the square root is taken.
int finalize_scalar(kk_ctx ctx, int part_row_idx, int n, double* out, bool with_sqrt);
// per-column sums of `m` partial rows (the reduction tail of project-type kernels)
int finalize_rows(kk_ctx ctx, const double* part, int nblk, int m, double* ws_a, double* ws_b);
