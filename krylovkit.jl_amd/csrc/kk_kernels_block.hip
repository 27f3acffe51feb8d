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

template <int NG, bool SAME>  // NG groups of 16 X-columns; SAME: Y is X (p <= 16), one load stream feeds both operands
__global__ __launch_bounds__(KK_TPB) void k_block_gram(const double* __restrict__ X, int64_t ldx, int p,
                                                       const double* __restrict__ Y, int64_t ldy, int q, int64_t ld,
                                                       int64_t rpb, double* __restrict__ part, int nt_x) {
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [NG][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    v4d acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = v4d{0.0, 0.0, 0.0, 0.0};
    const bool yok = c < q;
    const bool xnt = (X != Y) && nt_x;
    for (int64_t rc = r0 + wave * BG_CHUNK; rc < r1; rc += 4 * BG_CHUNK) {
        // rows of lane (c, kq): {rc + 4t + 2kq, +1 : t = 0,2,4,6}  -- the four lanes of one column read 64
        // contiguous bytes per load instruction (same row -> k-slot map for X and Y, so the contraction is unchanged)
        const int64_t row = rc + kq * 2;
        double yv[BG_T];
        if (SAME) {
        } else if (yok) {
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
                if (xnt) {   // X is streamed once: non-temporal, as in the single-vector kernels
#pragma unroll
                    for (int t = 0; t < BG_T; t += 2) { d2 v = ld2s(xp + 4 * t); xv[t] = v.x; xv[t + 1] = v.y; }
                } else {     // X == Y panels re-hit L2
#pragma unroll
                    for (int t = 0; t < BG_T; t += 2) { d2 v = ld2(xp + 4 * t); xv[t] = v.x; xv[t + 1] = v.y; }
                }
            } else {
#pragma unroll
                for (int t = 0; t < BG_T; ++t) xv[t] = 0.0;
            }
#pragma unroll
            for (int t = 0; t < BG_T; ++t)
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[t], SAME ? xv[t] : yv[t], acc[g], 0, 0, 0);
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

// Two panels in one pass over X:  C = X' Y  and  C2 = X' Y2  (q, q2 <= 16).  Used by the one-pass block step: Y = A Xnew and
// Y2 = Xnew, the newest basis block, whose Gram rows against the whole basis ride along (the block analogue of the Gram
// row that rides along in k_project for the low-synchronisation MGS).  When Y2 is group gx of X itself (block start a
// multiple of 16) its registers serve as the A operand of that group too: the ride-along then costs no memory traffic.
template <int NG>
__global__ __launch_bounds__(KK_TPB) void k_block_gram2(const double* __restrict__ X, int64_t ldx, int p,
                                                        const double* __restrict__ Y, int64_t ldy, int q,
                                                        const double* __restrict__ Y2, int64_t ldy2, int q2, int gx, int64_t ld,
                                                        int64_t rpb, double* __restrict__ part, double* __restrict__ part2,
                                                        double* __restrict__ part3) {
    // part3 (optional): Y'Y as well -- Y's registers serve both MFMA operands, no memory traffic.  The one-pass block step
    // gets |A X|-Gram this way and from it the Gram matrix of the NEXT residual block without reading that block
    // (W'W = (AX)'(AX) - P'Pc to first order), i.e. the first CholQR2 Gram pass of the next step.
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [NG][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    v4d acc[NG], acc2[NG], acc3 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < NG; ++g) { acc[g] = v4d{0.0, 0.0, 0.0, 0.0}; acc2[g] = v4d{0.0, 0.0, 0.0, 0.0}; }
    for (int64_t rc = r0 + wave * BG_CHUNK; rc < r1; rc += 4 * BG_CHUNK) {
        const int64_t row = rc + kq * 2;
        double yv[BG_T], zv[BG_T];
#pragma unroll
        for (int t = 0; t < BG_T; ++t) { yv[t] = 0.0; zv[t] = 0.0; }
        if (c < q) {
            const double* yp = Y + (int64_t)c * ldy + row;
#pragma unroll
            for (int t = 0; t < BG_T; t += 2) { d2 v = ld2(yp + 4 * t); yv[t] = v.x; yv[t + 1] = v.y; }
        }
        if (c < q2) {
            const double* zp = Y2 + (int64_t)c * ldy2 + row;
#pragma unroll
            for (int t = 0; t < BG_T; t += 2) { d2 v = ld2(zp + 4 * t); zv[t] = v.x; zv[t + 1] = v.y; }
        }
        if (part3) {
#pragma unroll
            for (int t = 0; t < BG_T; ++t) acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(yv[t], yv[t], acc3, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int col = g * 16 + c;
            double xv[BG_T];
            if (g == gx) {        // uniform: this group IS the ride-along block (q2 = 16 columns, all inside p)
#pragma unroll
                for (int t = 0; t < BG_T; ++t) xv[t] = zv[t];
            } else if (col < p) {
                const double* xp = X + (int64_t)col * ldx + row;
#pragma unroll
                for (int t = 0; t < BG_T; t += 2) { d2 v = ld2(xp + 4 * t); xv[t] = v.x; xv[t + 1] = v.y; }
            } else {
#pragma unroll
                for (int t = 0; t < BG_T; ++t) xv[t] = 0.0;
            }
#pragma unroll
            for (int t = 0; t < BG_T; ++t) {
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[t], yv[t], acc[g], 0, 0, 0);
                acc2[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[t], zv[t], acc2[g], 0, 0, 0);
            }
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double* a = &lds[(g * 4 + r) * 64 + lane];
                        const double v = pass ? acc2[g][r] : acc[g][r];
                        *a = (w == 0) ? v : (*a + v);
                    }
            }
            __syncthreads();
        }
        double* dst = (pass ? part2 : part) + (int64_t)blockIdx.x * (NG * 256);
        for (int e = tid; e < NG * 256; e += KK_TPB) dst[e] = lds[e];
        __syncthreads();
    }
    if (part3) {
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* a = &lds[r * 64 + lane];
                    *a = (w == 0) ? acc3[r] : (*a + acc3[r]);
                }
            }
            __syncthreads();
        }
        part3[(int64_t)blockIdx.x * 256 + tid] = lds[tid];
    }
}

// Software-pipelined form of k_block_gram2 (round 4).  The kernel above issues the loads of a 32-row chunk, waits, and then
// feeds the matrix pipe for (16 NG + 8) x 64 cycles before it can request the next chunk: with two waves per SIMD (228 VGPRs)
// memory time and MFMA time ADD (NG = 3 at 10M rows: 0.73 ms of stream + 0.46 ms of MFMA = the measured 1.19 ms).  Here every
// wave keeps two chunks in registers (ping-pong, unified 512-register file at one or two waves per SIMD) and requests chunk
// i+1 before the first MFMA of chunk i, so a single wave overlaps its own stream with its own matrix work.  Lanes whose column
// lies outside the panel read a clamped (valid) column instead of zeros: an A-operand column only reaches its own row of the
// C tile, a B-operand column only its own column, and the finalize kernels write i < p, j < q only.
template <int NG>
struct g2_chunk {
    d2 x[NG][BG_T / 2];
    d2 y[BG_T / 2], z[BG_T / 2];
};
typedef unsigned v4u_blk __attribute__((ext_vector_type(4)));
// Lane -> address map of the panel streams (tools/column_panel_read.hip, profiles/r04_column_panel_read.jsonl):
//   0  the MFMA operand layout itself: lane (c = l & 15, kq = l >> 4) reads 16 B of column c, one instruction = 16 columns x
//      64 B.  Ceiling 6.2 TB/s with plain loads; non-temporal loads are SLOWER (5.3: the two halves of a line arrive in
//      different instructions);
//   1  whole cache lines: lane (c8 = l & 7, h = l >> 3) reads 16 B of column c8 (c8 + 8 in the odd instructions), one
//      instruction = 8 columns x 128 B, non-temporal: 6.7-6.8 TB/s.  The operand layout is restored in registers: lanes l and
//      l ^ 8 exchange one of their two 16-byte values (8 DPP moves per pair of loads, row_ror:8 with a bank mask).
#ifndef KK_G2P_LINE
#define KK_G2P_LINE 1
#endif
#define KK_G2P_AUX (KK_G2P_LINE ? 2 : 0)
// 16-byte load through a buffer descriptor: the address is descriptor base + lane offset (32 bit) + scalar offset + an
// immediate -- no 64-bit address registers per stream -- and, being an opaque intrinsic, it stays where it is written
// relative to the compiler barrier that follows every request (a plain load of a `const __restrict__` argument is free to
// sink to its first use, which is exactly what un-pipelines the loop: seen in the ISA of the first version)
template <int IMM>
__device__ __forceinline__ v4u_blk g2_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff + IMM, soff, KK_G2P_AUX);
}
__device__ __forceinline__ d2 g2_d2(v4u_blk t) {
    d2 o;
    o.x = __longlong_as_double((long long)(((unsigned long long)t.y << 32) | t.x));
    o.y = __longlong_as_double((long long)(((unsigned long long)t.w << 32) | t.z));
    return o;
}
// whole-line map: a = this lane's 16 B of column c8, b = of column c8 + 8, both at row pair h = l >> 3 of a 16-row group.
// Lane l = c8 + 8 b3 + 16 kq must end up with column (l & 15) at the row pairs 2 kq and 2 kq + 1:
//   first  = b3 ? partner's b : own a        second = b3 ? own b : partner's a        (partner = l ^ 8, same 16-lane row)
__device__ __forceinline__ void g2_line_fix(v4u_blk a, v4u_blk b, d2& first, d2& second) {
    v4u_blk f, s;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = (unsigned)__builtin_amdgcn_update_dpp((int)a[e], (int)b[e], 0x128 /* row_ror:8 */, 0xF, 0xC /* lanes 8-15 of a row */, false);
        s[e] = (unsigned)__builtin_amdgcn_update_dpp((int)b[e], (int)a[e], 0x128, 0xF, 0x3 /* lanes 0-7 */, false);
    }
    first = g2_d2(f);
    second = g2_d2(s);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t g2_rsrc(const double* p, int64_t bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
template <int NG>
struct g2_streams {
    __amdgpu_buffer_rsrc_t x[NG], y, z;
    unsigned vx[NG], vy, vz;        // lane part of the byte offset: clamped column * ld * 8 + row part
    unsigned vx2[NG], vy2, vz2;     // whole-line map: the same for the second column of the lane (c8 + 8)
};
// one 32-row tile of one 16-column stream -> four d2 per lane in the operand layout (t: rows 8t + 2kq .. in map 0; map 1: see g2_line_fix)
__device__ __forceinline__ void g2_load_tile(d2 (&o)[BG_T / 2], __amdgpu_buffer_rsrc_t r, unsigned v, unsigned v2, unsigned soff) {
#if KK_G2P_LINE
    const v4u_blk a0 = g2_bload<0>(r, v, soff), b0 = g2_bload<0>(r, v2, soff);
    const v4u_blk a1 = g2_bload<128>(r, v, soff), b1 = g2_bload<128>(r, v2, soff);
    g2_line_fix(a0, b0, o[0], o[1]);
    g2_line_fix(a1, b1, o[2], o[3]);
#else
    o[0] = g2_d2(g2_bload<0>(r, v, soff)); o[1] = g2_d2(g2_bload<64>(r, v, soff));
    o[2] = g2_d2(g2_bload<128>(r, v, soff)); o[3] = g2_d2(g2_bload<192>(r, v, soff));
#endif
}
// GXL: the ride-along block Y2 is the LAST group of this X panel -- its tile is x[NG-1], nothing extra is read
template <int NG, bool GXL>
__device__ __forceinline__ void g2_request(g2_chunk<NG>& T, const g2_streams<NG>& S, unsigned soff) {
    g2_load_tile(T.y, S.y, S.vy, S.vy2, soff);
    if (!GXL) g2_load_tile(T.z, S.z, S.vz, S.vz2, soff);
#pragma unroll
    for (int g = 0; g < NG; ++g) g2_load_tile(T.x[g], S.x[g], S.vx[g], S.vx2[g], soff);
    asm volatile("" ::: "memory");
}
template <int NG, bool P3, bool GXL>
__device__ __forceinline__ void g2_consume(const g2_chunk<NG>& T, v4d (&acc)[NG], v4d (&acc2)[NG], v4d& acc3) {
    if (P3) {
#pragma unroll
        for (int t = 0; t < BG_T / 2; ++t) {
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(T.y[t].x, T.y[t].x, acc3, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(T.y[t].y, T.y[t].y, acc3, 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int t = 0; t < BG_T / 2; ++t) {
            const d2 xv = T.x[g][t];
            const d2 zv = GXL ? T.x[NG - 1][t] : T.z[t];
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv.x, T.y[t].x, acc[g], 0, 0, 0);
            acc2[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv.x, zv.x, acc2[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv.y, T.y[t].y, acc[g], 0, 0, 0);
            acc2[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv.y, zv.y, acc2[g], 0, 0, 0);
        }
    }
    asm volatile("" ::: "memory");
}
// requires 16 * ldx * 8 < 2^31 (one descriptor spans a 16-column group); the launcher checks
template <int NG, bool P3, bool GXL>
__global__ __launch_bounds__(KK_TPB) void k_block_gram2p(const double* __restrict__ X, int64_t ldx, int p,
                                                         const double* __restrict__ Y, int64_t ldy, int q,
                                                         const double* __restrict__ Y2, int64_t ldy2, int q2, int64_t ld,
                                                         int64_t rpb, double* __restrict__ part, double* __restrict__ part2,
                                                         double* __restrict__ part3) {
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [NG][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    v4d acc[NG], acc2[NG], acc3 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < NG; ++g) { acc[g] = v4d{0.0, 0.0, 0.0, 0.0}; acc2[g] = v4d{0.0, 0.0, 0.0, 0.0}; }
    // one descriptor per 16-column group, columns clamped into the panel (see above); rows beyond ld read as zero
    g2_streams<NG> S;
    (void)c; (void)kq;
#if KK_G2P_LINE
    const int c1 = lane & 7, c2 = c1 + 8;
    const unsigned rowpart = (unsigned)(lane >> 3) * 16;
#else
    const int c1 = c, c2 = c;
    const unsigned rowpart = (unsigned)kq * 16;
#endif
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ncol = imin(16, p - g * 16) > 0 ? imin(16, p - g * 16) : 1;
        const int g0 = imin(g * 16, p - 1);
        S.x[g] = g2_rsrc(X + (int64_t)g0 * ldx, ((int64_t)(ncol - 1) * ldx + ld) * 8);
        S.vx[g] = (unsigned)((int64_t)imin(c1, ncol - 1) * ldx * 8) + rowpart;
        S.vx2[g] = (unsigned)((int64_t)imin(c2, ncol - 1) * ldx * 8) + rowpart;
    }
    S.y = g2_rsrc(Y, ((int64_t)(q - 1) * ldy + ld) * 8);
    S.vy = (unsigned)((int64_t)imin(c1, q - 1) * ldy * 8) + rowpart;
    S.vy2 = (unsigned)((int64_t)imin(c2, q - 1) * ldy * 8) + rowpart;
    S.z = g2_rsrc(Y2, ((int64_t)(q2 - 1) * ldy2 + ld) * 8);
    S.vz = (unsigned)((int64_t)imin(c1, q2 - 1) * ldy2 * 8) + rowpart;
    S.vz2 = (unsigned)((int64_t)imin(c2, q2 - 1) * ldy2 * 8) + rowpart;
    // rpb and ld are multiples of KK_SUB = 512 rows = 4 chunks per wave: every wave has an even number of chunks, so the loop
    // below is branch-free between a request and its use
    const unsigned step = 4 * BG_CHUNK * 8;                       // bytes between two chunks of one wave
    const int nch = (int)((r1 - r0) / (4 * BG_CHUNK));
    unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((r0 + wave * BG_CHUNK) * 8));
    g2_chunk<NG> A, B;
    if (nch > 0) g2_request<NG, GXL>(A, S, so);
    for (int i = 0; i < nch; i += 2) {
        g2_request<NG, GXL>(B, S, so + step);
        g2_consume<NG, P3, GXL>(A, acc, acc2, acc3);
        g2_request<NG, GXL>(A, S, i + 2 < nch ? so + 2 * step : so);   // past the end: a harmless re-read
        g2_consume<NG, P3, GXL>(B, acc, acc2, acc3);
        so += 2 * step;
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double* a = &lds[(g * 4 + r) * 64 + lane];
                        const double v = pass ? acc2[g][r] : acc[g][r];
                        *a = (w == 0) ? v : (*a + v);
                    }
            }
            __syncthreads();
        }
        double* dst = (pass ? part2 : part) + (int64_t)blockIdx.x * (NG * 256);
        for (int e = tid; e < NG * 256; e += KK_TPB) dst[e] = lds[e];
        __syncthreads();
    }
    if (P3) {
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* a = &lds[r * 64 + lane];
                    *a = (w == 0) ? acc3[r] : (*a + acc3[r]);
                }
            }
            __syncthreads();
        }
        part3[(int64_t)blockIdx.x * 256 + tid] = lds[tid];
    }
}

// Gram panel against a tile that is COMPUTED on the fly instead of read:
//     T(rows, q) = beta * Yin + alpha * Z(rows, nz) S(nz, q)            C = X' T
//   * three-term update + re-orthogonalisation panel of block_lanczosrecurrence in one pass (blocklanczos.jl:253-260 then
//     :277-284):  T = AX - [Xprev X][B'; M]  is never written -- the update that follows subtracts V (P + [0; B'; M]) from
//     the ORIGINAL AX -- which removes one read and one write of the 16-column block per step (and with the write the
//     read/write mixing penalty: tools/stream_tile.hip, 6.9 TB/s read-only vs 5.1 TB/s with 16 output columns);
//   * second round of CholQR2:  T = B R1^-1 is stored (STORE) and its Gram matrix T'T accumulated in the same pass (XT).
// Per 32-row chunk a wave first forms its tile with plain FMAs in row-owner layout (lane = row, 8 columns each, Z read with
// 256-byte contiguous 8-byte loads, coefficients broadcast from LDS), parks it in a wave-private LDS slab [16][36] and reads
// it back in the column-owner layout of the MFMA B operand; X is streamed exactly as in k_block_gram.
#define BGT_LD 36                    // row stride (doubles) of the LDS tile: 16-byte aligned, spreads the columns over the banks
template <int NG, bool XT, bool STORE, bool HASY>
__global__ __launch_bounds__(KK_TPB) void k_block_gram_tile(const double* __restrict__ X, int64_t ldx, int p,
                                                            const double* __restrict__ Yin, int64_t ldy,
                                                            const double* __restrict__ Z, int64_t ldz, int nz,
                                                            const double* __restrict__ S, int st, double alpha, double beta,
                                                            double* __restrict__ Yout, int64_t ldyo, int q, int64_t ld,
                                                            int64_t rpb, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [NG*256] reduction | [nz*16] coefficients | [4][16*BGT_LD] tiles
    double* csm = lds + NG * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* tile = csm + 32 * 16 + wave * (16 * BGT_LD);
    for (int e = tid; e < nz * 16; e += KK_TPB) { const int z = e >> 4, j = e & 15; csm[e] = j < q ? S[(int64_t)z * st + j] : 0.0; }
    __syncthreads();
    const int c = lane & 15, kq = lane >> 4;
    // row-owner layout of phase 1: lane = (row rl of the chunk, half h); half h multiplies the Z columns [h*nzh, (h+1)*nzh)
    // into all 16 tile columns (no Z element is loaded twice) and brings in the Y columns 8h .. 8h+7; the two partial
    // tiles are added in the LDS slab.  The loads of the NEXT chunk are issued before the MFMA phase of the current one.
    const int rl = lane & 31, h = lane >> 5;
    const int nzh = (nz + 1) >> 1, z0 = h * nzh, zn = (z0 + nzh <= nz) ? nzh : (nz > z0 ? nz - z0 : 0);
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    v4d acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = v4d{0.0, 0.0, 0.0, 0.0};
    double zr[16], yr[8];
    auto fetch = [&](int64_t rc) {
        const int64_t row = rc + rl;
#pragma unroll
        for (int u = 0; u < 16; ++u) zr[u] = (u < zn) ? Z[(int64_t)(z0 + u) * ldz + row] : 0.0;
        if (HASY) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) yr[jj] = (8 * h + jj < q) ? Yin[(int64_t)(8 * h + jj) * ldy + row] : 0.0;
        }
    };
    int64_t rc = r0 + wave * BG_CHUNK;
    if (rc < r1) fetch(rc);
    for (; rc < r1; rc += 4 * BG_CHUNK) {
        // ---- phase 1: the tile
        {
            double t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = 0.0;
            if (HASY) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    if (h == 0) t[jj] = beta * yr[jj];
                    else t[8 + jj] = beta * yr[jj];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u < zn) {   // wave-uniform per half except for the odd-nz remainder
                    const double zv = alpha * zr[u];
                    const d2* sc = reinterpret_cast<const d2*>(csm + (z0 + u) * 16);
#pragma unroll
                    for (int j2 = 0; j2 < 8; ++j2) {
                        const d2 sv = sc[j2];
                        t[2 * j2] = fma(zv, sv.x, t[2 * j2]);
                        t[2 * j2 + 1] = fma(zv, sv.y, t[2 * j2 + 1]);
                    }
                }
            }
            if (h == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) tile[j * BGT_LD + rl] = t[j];
            }
            __builtin_amdgcn_wave_barrier();
            if (h == 1) {
#pragma unroll
                for (int j = 0; j < 16; ++j) tile[j * BGT_LD + rl] += t[j];
            }
            __builtin_amdgcn_wave_barrier();
            if (STORE) {
                const int64_t row = rc + rl;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    if (8 * h + jj < q) Yout[(int64_t)(8 * h + jj) * ldyo + row] = tile[(8 * h + jj) * BGT_LD + rl];
            }
        }
        if (rc + 4 * BG_CHUNK < r1) fetch(rc + 4 * BG_CHUNK);   // in flight during the MFMA phase below
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab is wave-private: LDS is in order per wave
        // ---- phase 2: Gram of the tile against X (column-owner layout, as k_block_gram)
        double yv[BG_T];
#pragma unroll
        for (int t = 0; t < BG_T; t += 2) {
            const d2 v = *reinterpret_cast<const d2*>(tile + c * BGT_LD + 4 * t + 2 * kq);
            yv[t] = v.x; yv[t + 1] = v.y;
        }
        const int64_t row = rc + kq * 2;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double xv[BG_T];
            if (XT) {
#pragma unroll
                for (int t = 0; t < BG_T; ++t) xv[t] = yv[t];
            } else {
                const int col = g * 16 + c;
                if (col < p) {
                    const double* xp = X + (int64_t)col * ldx + row;
#pragma unroll
                    for (int t = 0; t < BG_T; t += 2) { d2 v = ld2s(xp + 4 * t); xv[t] = v.x; xv[t + 1] = v.y; }
                } else {
#pragma unroll
                    for (int t = 0; t < BG_T; ++t) xv[t] = 0.0;
                }
            }
#pragma unroll
            for (int t = 0; t < BG_T; ++t) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[t], yv[t], acc[g], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");   // the next chunk overwrites the slab only after these reads
    }
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

// C[i*rs + j*cs] = sum_b part[b][e(i,j)]: rs = 1, cs = ldc is the column-major panel; rs = row stride, cs = 1 writes the panel
// row-major, i.e. directly in the coefficient layout k_block_update reads.  A block reduces 16 consecutive tile elements:
// thread (el = tid & 15, sl = tid >> 4) adds the partial tiles sl, sl + 16, ... (128-byte coalesced loads), the 16 slices are
// combined through LDS in a fixed order -- deterministic, and the reduction over up to 2048 partial tiles no longer runs
// as one serial chain per output entry (306 us -> a few us per Gram panel at 8 blocks per CU).
__device__ __forceinline__ void finalize_gram_body(const double* __restrict__ part, int nblk, int ng, int p, int q,
                                                   double* __restrict__ C, int rs, int cs, int blk) {
    __shared__ double sm[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e = blk * 16 + el;                         // tile element: e = (g*4 + r)*64 + lane
    const int64_t stride = (int64_t)ng * 256;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int b = sl;
    for (; b + 48 < nblk; b += 64) {
        a0 += part[(int64_t)b * stride + e];
        a1 += part[(int64_t)(b + 16) * stride + e];
        a2 += part[(int64_t)(b + 32) * stride + e];
        a3 += part[(int64_t)(b + 48) * stride + e];
    }
    for (; b < nblk; b += 16) a0 += part[(int64_t)b * stride + e];
    sm[sl][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0) {
        double a = 0;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) a += sm[s2][el];
        const int g = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        const int j = lane & 15, i = g * 16 + (lane >> 4) + 4 * r;
        if (i < p && j < q) C[(int64_t)i * rs + (int64_t)j * cs] = a;
    }
}
__global__ __launch_bounds__(KK_TPB) void k_finalize_gram(const double* __restrict__ part, int nblk, int ng, int p, int q,
                                                          double* __restrict__ C, int rs, int cs) {
    finalize_gram_body(part, nblk, ng, p, q, C, rs, cs, blockIdx.x);
}
// the three panels of one k_block_gram2 launch in ONE finalize launch: blocks [0, ng*16) -> C (X'Y), [ng*16, 2 ng*16) -> C2
// (X'Y2), the last 16 -> C3 (Y'Y, one tile, column-major ld 16) when present
__global__ __launch_bounds__(KK_TPB) void k_finalize_gram3(const double* __restrict__ part, const double* __restrict__ part2,
                                                           const double* __restrict__ part3, int nblk, int ng, int p, int q, int q2,
                                                           double* __restrict__ C, int rs, double* __restrict__ C2, int rs2,
                                                           double* __restrict__ C3) {
    const int nb1 = ng * 16;
    if ((int)blockIdx.x < nb1) finalize_gram_body(part, nblk, ng, p, q, C, rs, 1, blockIdx.x);
    else if ((int)blockIdx.x < 2 * nb1) finalize_gram_body(part2, nblk, ng, p, q2, C2, rs2, 1, blockIdx.x - nb1);
    else finalize_gram_body(part3, nblk, 1, q, q, C3, 1, 16, blockIdx.x - 2 * nb1);
}

// ---- one-block dense helpers of the asynchronous block step (p <= 16): the Cholesky factors, their inverses and the
// coefficient panels are formed on the device so that a whole BlockLanczos expand! is enqueued without a host round trip;
// the host reads the flag afterwards and repeats the step on the synchronous route if a safety test failed.
// upper Cholesky G = R'R in LDS; returns false (uniformly) if a pivot is not safely positive:
// pivot^2 must exceed rel^2 * G_jj and abs_min^2   (same test as the host-side chol_upper_safe)
// One wave (64 lanes, launch bounds 64): right-looking factorisation in LDS, every step a handful of LDS round trips
// (thread-0-serial versions of these helpers took ~100 us per launch, 0.4 ms per block step).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__device__ bool chol16(const double* G, int p, double (*R)[17], double rel, double abs_min) {
    __shared__ double A[16][17], dg[16];
    __shared__ int ok_s;
    const int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) {
        const int i = e >> 4, j = e & 15;
        const double g = (i < p && j < p) ? G[i + p * j] : 0.0;
        A[i][j] = g; R[i][j] = 0.0;
        if (i == j) dg[i] = g;
    }
    if (lane == 0) ok_s = 1;
    wave_sync();
    for (int j = 0; j < p; ++j) {
        const double d2 = A[j][j], gjj = dg[j];   // A[j][j] = G_jj - sum_k R[k][j]^2, subtracted for k ascending
        if (!(d2 > rel * rel * gjj) || !(d2 > abs_min * abs_min) || !(d2 < 1e300)) { if (lane == 0) ok_s = 0; break; }   // uniform
        const double rjj = sqrt(d2);
        if (lane < 16) R[j][lane] = (lane == j) ? rjj : (lane > j && lane < p ? A[j][lane] / rjj : 0.0);
        wave_sync();
        for (int e = lane; e < 256; e += 64) {
            const int a = e >> 4, b = e & 15;
            if (a > j && b >= a && b < p) A[a][b] -= R[j][a] * R[j][b];
        }
        wave_sync();
    }
    wave_sync();
    return ok_s != 0;
}
// Ri = R^-1 (upper triangular): lane j owns column j, back substitution out of LDS with the column in registers
__device__ void triu_inv16(double (*R)[17], int p, double (*Ri)[17]) {
    const int j = threadIdx.x;
    if (j < 16) {
        double ri[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ri[i] = 0.0;
        if (j < p) {
#pragma unroll
            for (int i = 15; i >= 0; --i) {
                if (i == j) ri[i] = 1.0 / R[j][j];
                else if (i < j) {
                    double t = 0;
#pragma unroll
                    for (int k = 15; k > 0; --k)
                        if (k > i && k <= j) t += R[i][k] * ri[k];
                    ri[i] = -t / R[i][i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Ri[i][j] = ri[i];
    }
    wave_sync();
}
// CholQR2, first factor: G (p x p col-major) -> R1 (p x p col-major, ld p) and the staged panel of R1^-1 (row-major, stride st)
__global__ __launch_bounds__(64) void k_blk_chol1(const double* __restrict__ G, int p, double abs_min, double* __restrict__ R1out,
                                                  double* __restrict__ S1, int st, double* __restrict__ flag) {
    __shared__ double R[16][17], Ri[16][17];
    const bool ok = chol16(G, p, R, 1e-5, abs_min);
    if (!ok) { if (threadIdx.x == 0) flag[0] = 1.0; return; }
    triu_inv16(R, p, Ri);
    for (int e = threadIdx.x; e < p * st; e += 64) { const int i = e / st, j = e % st; S1[e] = j < p ? Ri[i][j] : 0.0; }
    for (int e = threadIdx.x; e < p * p; e += 64) R1out[e] = R[e % p][e / p];
}
// second factor: G2 = Q1'Q1 must be close to I; R2 = chol(G2); staged panel of R2^-1; B = R2 R1 (p x p col-major, ld ldb);
// rows 0..p-1 of the three-term panel S3 (row-major, stride st): S3[i][j] = B[j][i]
__global__ __launch_bounds__(64) void k_blk_chol2(const double* __restrict__ G2, int p, const double* __restrict__ R1,
                                                  double* __restrict__ Bout, int ldb, double* __restrict__ S2, double* __restrict__ S3,
                                                  int st, double* __restrict__ flag, double skip_tol) {
    __shared__ double R[16][17], Ri[16][17], Bm[16][17];
    __shared__ double devs[64];
    {
        double dev = 0.0;
        for (int e = threadIdx.x; e < p * p; e += 64) {
            const int i = e % p, j = e / p;
            const double d = fabs(G2[e] - (i == j ? 1.0 : 0.0));
            dev = (d > dev || !(d == d)) ? (d == d ? d : 1e300) : dev;   // NaN -> huge -> flagged
        }
        devs[threadIdx.x] = dev;
    }
    wave_sync();
    double dev = 0.0;
    for (int t = 0; t < 64; ++t) dev = fmax(dev, devs[t]);   // every lane: the same value
    // every failure also raises flag[1]: the back-substitution that follows must not touch the block (after a normalised
    // commit it is the only copy of the residual block, T = W R1^-1, and the step is repeated from it)
    if (flag[0] != 0.0) { if (threadIdx.x == 0) flag[1] = 1.0; return; }   // first factor already failed
    if (threadIdx.x == 0) flag[2] = dev;   // |Q1'Q1 - I|_max of this step (diagnostics)
    if (!(dev < 1e-3)) { if (threadIdx.x == 0) { flag[0] = 2.0; flag[1] = 1.0; } return; }
    if (dev <= skip_tol) {
        // Q1 = W R1^-1 is orthonormal to skip_tol already (a well-conditioned block: cond(W)^2 eps): the second round would
        // be the identity up to that level.  B = R1, the back-substitution panel is the identity and flag[1] tells the
        // update kernel that follows not to run at all (one read and one write of the block saved).
        for (int e = threadIdx.x; e < p * p; e += 64) { const int i = e % p, j = e / p; Bm[i][j] = (i <= j) ? R1[i + p * j] : 0.0; }
        wave_sync();
        for (int e = threadIdx.x; e < p * p; e += 64) Bout[(e % p) + ldb * (e / p)] = Bm[e % p][e / p];
        for (int e = threadIdx.x; e < p * st; e += 64) {
            const int i = e / st, j = e % st;
            S2[e] = (j == i) ? 1.0 : 0.0;
            S3[e] = j < p ? Bm[j][i] : 0.0;
        }
        if (threadIdx.x == 0) flag[1] = 1.0;
        return;
    }
    const bool ok = chol16(G2, p, R, 1e-2, 0.0);
    if (!ok) { if (threadIdx.x == 0) { flag[0] = 2.0; flag[1] = 1.0; } return; }
    triu_inv16(R, p, Ri);
    for (int e = threadIdx.x; e < p * p; e += 64) {   // B = R2 R1 (upper triangular)
        const int i = e % p, j = e / p;
        double t = 0;
        for (int k = i; k <= j; ++k) t += R[i][k] * R1[k + p * j];
        Bm[i][j] = (i <= j) ? t : 0.0;
    }
    wave_sync();
    for (int e = threadIdx.x; e < p * p; e += 64) Bout[(e % p) + ldb * (e / p)] = Bm[e % p][e / p];
    for (int e = threadIdx.x; e < p * st; e += 64) {
        const int i = e / st, j = e % st;
        S2[e] = j < p ? Ri[i][j] : 0.0;
        S3[e] = j < p ? Bm[j][i] : 0.0;
    }
}
// C = P + [0 ; S3]: the re-orthogonalisation panel with the three-term coefficients added to its last nz rows (row-major, stride st)
__global__ __launch_bounds__(KK_TPB) void k_blk_combine(double* __restrict__ P, const double* __restrict__ S3, int kn, int nz, int st) {
    for (int e = threadIdx.x; e < nz * st; e += KK_TPB) P[(int64_t)(kn - nz) * st + e] += S3[e];
}
// One-pass block step (block_fuse bit 4): P = V'(A X) over the WHOLE basis is the coefficient panel of a single update
// AX - V P (the three-term part [B'; M] is its rows k-p .. k+p-1); M = X'(A X) is rows k .. k+p-1 of the panel.
__global__ __launch_bounds__(KK_TPB) void k_blk_panel_m(const double* __restrict__ P, int st, int k, int p, double* __restrict__ M, int ldm) {
    for (int e = threadIdx.x; e < p * p; e += KK_TPB) { const int i = e % p, j = e / p; M[i + ldm * j] = P[(int64_t)(k + i) * st + j]; }
}
// Safety test of the one-pass step: column j lost more than a factor 1/eta of its norm in the single projection
// (|w_j|^2 < eta^2 (|w_j|^2 + |P_j|^2), |A x_j|^2 = |w_j|^2 + |P_j|^2 for an orthonormal basis) -> flag 3: the caller repeats
// the step on the two-pass route (three-term recurrence, then block_reorthogonalize!), as the reference orders it.
__global__ __launch_bounds__(KK_TPB) void k_blk_onepass_check(const double* __restrict__ P, int st, int kn, int p,
                                                              const double* __restrict__ nrm2, double eta2, double* __restrict__ flag) {
    __shared__ double sm[16][17];
    const int j = threadIdx.x & 15, sl = threadIdx.x >> 4;
    double s = 0;
    if (j < p)
        for (int i = sl; i < kn; i += 16) { const double v = P[(int64_t)i * st + j]; s = fma(v, v, s); }
    sm[sl][j] = s;
    __syncthreads();
    if (sl == 0 && j < p) {
        double t = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += sm[u][j];
        const double w2 = nrm2[j];
        if (!(w2 >= eta2 * (w2 + t)) && flag[0] == 0.0) flag[0] = 3.0;   // NaN -> flagged; all writers store the same value
    }
}
// Gram rows of the newest block from the ride-along panel G2[j][i] = <v_j, x_i> (row-major, stride st):
//   gram[(k+i)*cap + j] = G2[j][i]  for j < k+i      (strictly-lower storage of kk_orth.hip, device mirror)
// gdiag (optional): |x_i|^2 - 1 of the newest block's vectors, the DIAGONAL of E = V'V - I, which the strictly-lower storage has no place for
__global__ __launch_bounds__(KK_TPB) void k_blk_gram_rows(const double* __restrict__ G2, int st, int k, int p, double* __restrict__ gram, int cap,
                                                          double* __restrict__ gdiag) {
    const int kn = k + p;
    for (int e = threadIdx.x + blockIdx.x * KK_TPB; e < kn * p; e += KK_TPB * gridDim.x) {
        const int j = e / p, i = e % p;
        if (j < k + i) gram[(int64_t)(k + i) * cap + j] = G2[(int64_t)j * st + i];
        else if (j == k + i && gdiag) gdiag[k + i] = G2[(int64_t)j * st + i] - 1.0;
    }
}
// One-pass projection with a non-orthonormal basis to first order: the coefficients of the projector onto span(V) are
// (V'V)^-1 V'y ~ (I - E) V'y, E = V'V - I (off-diagonal part from the strictly-lower Gram rows, diagonal from gdiag):
//   Pc[i][j] = P[i][j] - sum_{l != i} G(i, l) P[l][j] - (|v_i|^2 - 1) P[i][j]
// (the diagonal matters since round 4: a block whose second CholQR2 round was skipped at |Q1'Q1 - I| <= 2e-14 carries that
// much in its norms; left out, V'w keeps (|v_i|^2 - 1) P_i and the predicted Gram matrix of the residual block -- the normalised
// commit factors it -- is off by P'DP, 1e-13 relative on config 5 instead of 1e-15)
// Without it the error E of the basis re-enters every new block multiplied by |P| / |w| > 1 and grows geometrically (one
// classical Gram-Schmidt pass is not enough); with it V'w = O(E^2 |P|) + the rounding of the panel itself.
__global__ __launch_bounds__(KK_TPB) void k_blk_panel_correct(const double* __restrict__ P, int st, int kn, int p,
                                                              const double* __restrict__ gram, int cap, double* __restrict__ Pc,
                                                              const double* __restrict__ gdiag) {
    extern __shared__ double psm[];   // P staged: kn * st (every block stages the whole panel and computes its 256 entries)
    for (int e = threadIdx.x; e < kn * st; e += KK_TPB) psm[e] = P[e];
    __syncthreads();
    for (int e = blockIdx.x * KK_TPB + threadIdx.x; e < kn * st; e += gridDim.x * KK_TPB) {
        const int i = e / st, j = e % st;
        double a = 0;
        if (j < p) {
            for (int l = 0; l < i; ++l) a = fma(gram[(int64_t)i * cap + l], psm[l * st + j], a);
            for (int l = i + 1; l < kn; ++l) a = fma(gram[(int64_t)l * cap + i], psm[l * st + j], a);
            if (gdiag) a = fma(gdiag[i], psm[i * st + j], a);
        }
        Pc[e] = psm[e] - a;
    }
}
// Gram matrix of the new residual block W = AX - V Pc without reading it:  W'W = (AX)'(AX) - P'Pc  (first order in
// E = V'V - I, which Pc = (I - E)P carries), P / Pc row-major kn x st, GYY = (AX)'(AX) column-major ld 16.  The diagonal is
// replaced by the column norms the update kernel measured on the actual block (exact); the result is symmetrised and
// written column-major with leading dimension p -- the layout k_blk_chol1 reads.  The subtraction loses
// log2(|AX|^2 / |W|^2) bits; whatever that costs in orthonormality of Q1 = W R1^-1 is measured by the fused second round
// (|Q1'Q1 - I|), which then runs its back-substitution or not.
__global__ __launch_bounds__(KK_TPB) void k_blk_resid_gram(const double* __restrict__ P, const double* __restrict__ Pc, int st, int kn,
                                                           int p, const double* __restrict__ GYY, const double* __restrict__ nrm2,
                                                           double* __restrict__ GW) {
    __shared__ double g[16][17];
    extern __shared__ double pp[];   // both panels staged (2 * kn * st doubles): the kn-long sums then run out of LDS
    double* ps = pp;
    double* pcs = pp + (size_t)kn * st;
    for (int e = threadIdx.x; e < kn * st; e += KK_TPB) { ps[e] = P[e]; pcs[e] = Pc[e]; }
    __syncthreads();
    const int i = threadIdx.x & 15, j = threadIdx.x >> 4;
    double a = 0;
    if (i < p && j < p)
        for (int l = 0; l < kn; ++l) a = fma(ps[l * st + i], pcs[l * st + j], a);
    g[i][j] = (i < p && j < p) ? GYY[i + 16 * j] - a : 0.0;
    __syncthreads();
    if (i < p && j < p) GW[i + p * j] = (i == j) ? (nrm2 ? nrm2[i] : g[i][i]) : 0.5 * (g[i][j] + g[j][i]);
}
// Normalised commit of the residual block (round 4): the first CholQR2 factor of the NEXT step from the Gram matrix the
// one-pass panel predicts (GWE = (AX)'(AX) - P'Pc, all entries estimated), formed BEFORE the residual update runs, so that
// the update can write T = W R1^-1 straight into the next basis slot.  cflag[0] = 0: commit; anything else: the update
// writes the plain residual block.  The commit needs the same pivot margins as k_blk_chol1 and, because every entry of GWE
// carries eps |A x_j|^2, a residual column that kept at least `keep` of its squared norm in the projection.
__global__ __launch_bounds__(64) void k_blk_commit_prep(const double* __restrict__ GWE, const double* __restrict__ GYY, int p, double abs_min,
                                                        double keep, double* __restrict__ R1out, double* __restrict__ S1, int st,
                                                        double* __restrict__ cflag) {
    __shared__ double R[16][17], Ri[16][17];
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    wave_sync();
    if ((int)threadIdx.x < p) {
        const double gw = GWE[threadIdx.x + p * threadIdx.x], gy = GYY[threadIdx.x + 16 * threadIdx.x];
        if (!(gw > keep * gy) || !(gw < 1e300)) bad = 1;
    }
    wave_sync();
    if (bad) { if (threadIdx.x == 0) cflag[0] = 2.0; return; }
    const bool ok = chol16(GWE, p, R, 1e-5, abs_min);
    if (!ok) { if (threadIdx.x == 0) cflag[0] = 1.0; return; }
    triu_inv16(R, p, Ri);
    for (int e = threadIdx.x; e < p * st; e += 64) { const int i = e / st, j = e % st; S1[e] = j < p ? Ri[i][j] : 0.0; }
    for (int e = threadIdx.x; e < p * p; e += 64) R1out[e] = R[e % p][e / p];
    if (threadIdx.x == 0) cflag[0] = 0.0;
}
// rows p..2p-1 of the three-term panel: S3[p + i][j] = M[i][j]  (M col-major, ld ldm)
__global__ __launch_bounds__(64) void k_blk_fill_m(const double* __restrict__ M, int ldm, int p, double* __restrict__ S3, int st) {
    for (int e = threadIdx.x; e < p * st; e += 64) {
        const int i = e / st, j = e % st;
        S3[(p + i) * st + j] = j < p ? M[i + ldm * j] : 0.0;
    }
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

// Variant with the coefficient panel in LDS.  The round-1 kernel fetches the 16 coefficients of every basis column with
// scalar loads right before their FMAs; tools/stream_tile.hip shows that a pure read of the same 112 columns WITH the
// same 16 FMAs per double runs at 6.8 TB/s, the update at 5.0: the difference is the exposed scalar-cache latency (five
// lgkmcnt(0) waits per batch of four columns, 64 SGPRs cannot hold a second batch).  Here the block copies the panel
// (m x NB doubles, <= 17 KB) to LDS once and every lane reads the coefficients with broadcast ds_read_b128 (same address
// in all lanes: conflict-free), NB/2 at a time; the column norms are reduced per iteration with wave sums instead of
// per-thread LDS accumulators (32 KB less LDS per block).
#ifndef KK_BUL_NT
#define KK_BUL_NT 1     // non-temporal Win loads and W stores: -0.3 ... -0.8 % on the block step (same-box A/B)
#endif
#if KK_BUL_NT
#define BUL_LD_WIN ld2s
#define BUL_ST st2s
#else
#define BUL_LD_WIN ld2
#define BUL_ST st2
#endif
template <int NB, bool BZERO>
__global__ __launch_bounds__(KK_TPB, 4) void k_block_update_lds(const double* V, int64_t ld, int m, const double* Win,
                                                             double* Wout, int64_t ldw_in, int64_t ldw_out, int nb,
                                                             const double* __restrict__ S, double alpha, double beta,
                                                             int64_t rpb, double* __restrict__ part_nrm,
                                                             const double* __restrict__ skip) {
    // `skip` (optional device flag): the transform is the identity -- decided on the device by an earlier kernel of the
    // same stream (second CholQR2 round of a block that is orthonormal already) -- and the pass is not executed
    if (skip && *skip != 0.0) return;
    extern __shared__ __attribute__((aligned(16))) double ssm[];   // [m][NB] coefficients, then [NB][4] norm slots
    double* nsl = ssm + (size_t)m * NB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int e = tid; e < m * NB; e += KK_TPB) ssm[e] = S[e];
    if (tid < NB * 4) nsl[tid] = 0.0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + tid * 2; r < r1; r += KK_SUB) {
        d2 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = d2{0.0, 0.0};
        int c = 0;
        d2 xn[4];
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
                const d2* Sc = reinterpret_cast<const d2*>(ssm + (size_t)(c + u) * NB);
#pragma unroll
                for (int j2 = 0; j2 < NB / 2; ++j2) {
                    const d2 sv = Sc[j2];
                    acc[2 * j2].x = fma(sv.x, x[u].x, acc[2 * j2].x); acc[2 * j2].y = fma(sv.x, x[u].y, acc[2 * j2].y);
                    acc[2 * j2 + 1].x = fma(sv.y, x[u].x, acc[2 * j2 + 1].x); acc[2 * j2 + 1].y = fma(sv.y, x[u].y, acc[2 * j2 + 1].y);
                }
            }
        }
        for (; c < m; ++c) {
            const d2 x = ld2(V + (int64_t)c * ld + r);
            const d2* Sc = reinterpret_cast<const d2*>(ssm + (size_t)c * NB);
#pragma unroll
            for (int j2 = 0; j2 < NB / 2; ++j2) {
                const d2 sv = Sc[j2];
                acc[2 * j2].x = fma(sv.x, x.x, acc[2 * j2].x); acc[2 * j2].y = fma(sv.x, x.y, acc[2 * j2].y);
                acc[2 * j2 + 1].x = fma(sv.y, x.x, acc[2 * j2 + 1].x); acc[2 * j2 + 1].y = fma(sv.y, x.y, acc[2 * j2 + 1].y);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < nb) {
                d2 w{alpha * acc[j].x, alpha * acc[j].y};
                if (!BZERO) {
                    const d2 wi = BUL_LD_WIN(Win + (int64_t)j * ldw_in + r);
                    w.x = fma(beta, wi.x, w.x); w.y = fma(beta, wi.y, w.y);
                }
                BUL_ST(Wout + (int64_t)j * ldw_out + r, w);
                if (part_nrm) {
                    const double t = wave_sum(fma(w.x, w.x, w.y * w.y));
                    if (lane == 0) nsl[j * 4 + wave] += t;
                }
            }
        }
    }
    if (part_nrm) {
        __syncthreads();
        if (tid < nb) part_nrm[(int64_t)tid * KK_MAX_BLOCKS + blockIdx.x] = (nsl[tid * 4] + nsl[tid * 4 + 1]) + (nsl[tid * 4 + 2] + nsl[tid * 4 + 3]);
    }
}

// MFMA form of the block update ("W -= V P on v_mfma_f64_16x16x4_f64").  The product is taken TRANSPOSED,
// W^T (16 x rows) += P^T (16 x m) V^T (m x rows), so that every operand sits where the data already is:
//   * B operand = V^T: lane (i = lane % 16, k = lane / 16) supplies V[row, column 4q + k] -- ONE 16-byte load per lane covers the
//     two rows 2i, 2i + 1 of a 32-row tile (.x -> the even-row accumulator, .y -> the odd-row one): 16 lanes x 16 B = 256
//     contiguous bytes of one basis column, four columns per load instruction, no 8-byte access, no re-layout;
//   * A operand = P^T: lane supplies P[4q + k][n = i], one double per four MFMAs (two tiles x even / odd), read from an LDS
//     copy of the panel laid out in fragment order (conflict-free 512-byte wave reads; no broadcast traffic: the VALU kernel
//     issues eight 16-byte broadcast reads per basis column and lane);
//   * D = W^T: lane holds the residual columns n = lane / 16 + 4 r, r < 4, of its row pair -- the Win loads and W stores are
//     again 256 contiguous bytes per column.
// gfx950's f64 matrix rate equals its vector rate: the gain is not flops but instruction issue -- per 32 rows x 4 columns one
// load, one LDS read and two MFMAs instead of one load, eight LDS reads and 32 FMAs.  The summation over the basis columns runs
// in the order q = 0, 1, ... with four columns per MFMA: not the bit pattern of the VALU kernel (parity bar: 1e-10 on H).
template <bool BZERO, int TR /* 32-row tiles per wave iteration */, int PF /* column groups of loads in flight */, int MINB = 2>
__global__ __launch_bounds__(KK_TPB, MINB) void k_block_update_mfma(const double* V, int64_t ld, int m, const double* Win, double* Wout, int64_t ldw_in,
                                                                int64_t ldw_out, int nb, const double* __restrict__ S, int sstride, double alpha, double beta,
                                                                int64_t rpb, double* __restrict__ part_nrm, const double* __restrict__ skip) {
    if (skip && *skip != 0.0) return;
    extern __shared__ __attribute__((aligned(16))) double ssm[];   // [MQ][64] coefficient fragments, then [16][4] norm slots
    const int MQ = (m + 3) >> 2;
    double* nsl = ssm + (size_t)MQ * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int e = tid; e < MQ * 64; e += KK_TPB) {
        const int k = 4 * (e >> 6) + ((e & 63) >> 4), n = e & 15;
        ssm[e] = (k < m && n < sstride) ? S[(size_t)k * sstride + n] : 0.0;
    }
    if (tid < 64) nsl[tid] = 0.0;
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double nacc[4] = {0.0, 0.0, 0.0, 0.0};   // squared norms of columns lk + 4 r over this lane's rows
    // a wave owns TR consecutive 32-row tiles per iteration; rpb is a multiple of 512 = 4 waves x 128 rows
    for (int64_t rt = r0 + (int64_t)wave * (32 * TR); rt < r1; rt += 4 * 32 * TR) {
        v4d acc[TR][2];
#pragma unroll
        for (int t = 0; t < TR; ++t) { acc[t][0] = v4d{0.0, 0.0, 0.0, 0.0}; acc[t][1] = v4d{0.0, 0.0, 0.0, 0.0}; }
        const int64_t rbase = rt + 2 * li;
        d2 x[PF][TR];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int c = 4 * u + lk;
            const double* col = V + (int64_t)(c < m ? c : m - 1) * ld + rbase;
#pragma unroll
            for (int t = 0; t < TR; ++t) x[u][t] = ld2s(col + 32 * t);
        }
        for (int q0 = 0; q0 < MQ; q0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int q = q0 + u;
                if (q < MQ) {   // uniform
                    const double a = ssm[q * 64 + lane];
                    d2 xc[TR];
#pragma unroll
                    for (int t = 0; t < TR; ++t) xc[t] = x[u][t];
                    const int qn = q + PF;
                    if (qn < MQ) {
                        const int c = 4 * qn + lk;
                        const double* col = V + (int64_t)(c < m ? c : m - 1) * ld + rbase;
#pragma unroll
                        for (int t = 0; t < TR; ++t) x[u][t] = ld2s(col + 32 * t);
                    }
#pragma unroll
                    for (int t = 0; t < TR; ++t) {
                        acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xc[t].x, acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xc[t].y, acc[t][1], 0, 0, 0);
                    }
                }
            }
        }
        // epilogue: w = beta Win + alpha (V S), columns n = lk + 4 r of the row pairs (2 li, 2 li + 1) of every tile
        // (accumulator element r of lane l is D[l / 16 + 4 r][l % 16]: the layout k_finalize_gram decodes)
#pragma unroll
        for (int t = 0; t < TR; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = lk + 4 * r;
                if (n < nb) {
                    d2 w{alpha * acc[t][0][r], alpha * acc[t][1][r]};
                    if (!BZERO) {
                        const d2 wi = BUL_LD_WIN(Win + (int64_t)n * ldw_in + rbase + 32 * t);
                        w.x = fma(beta, wi.x, w.x); w.y = fma(beta, wi.y, w.y);
                    }
                    BUL_ST(Wout + (int64_t)n * ldw_out + rbase + 32 * t, w);
                    nacc[r] = fma(w.x, w.x, fma(w.y, w.y, nacc[r]));
                }
            }
        }
    }
    if (part_nrm) {
        // column lk + 4 r: sum over the 16 lanes of the group (row reduction inside a DPP row), then over the waves
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = nacc[r];
            v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
            if (li == 0) nsl[(lk + 4 * r) * 4 + wave] = v;
        }
        __syncthreads();
        if (tid < nb) part_nrm[(int64_t)tid * KK_MAX_BLOCKS + blockIdx.x] = (nsl[tid * 4] + nsl[tid * 4 + 1]) + (nsl[tid * 4 + 2] + nsl[tid * 4 + 3]);
    }
}

// Deep-prefetch variant: a ring of PF basis-column loads (16 B each) stays in flight per lane -- as many bytes in flight as
// the single-vector unproject kernel keeps with its 32 x 2 tile -- at 2 blocks per CU (up to 256 registers per lane).
// Same arithmetic, same summation order as k_block_update.
// Residual update of the one-pass block step with the NORMALISED COMMIT (round 4, VERDICT round 3 item 5: "CholQR2
// back-substitution fused into the residual update, one write of the 16-column block per step"):
//     w = Win - V S  (accumulated onto Win, column by column),  column norms of w,  and -- when the device flag allows --
//     t = w R1^-1   written to the NEXT BASIS SLOT instead of w to the residual area,   G2 += t' t   (MFMA).
// The next expand! then starts from T and G2: no Gram pass over the residual block, no Q1 = W R1^-1 pass (one read and one
// write of the block saved per step).  t' t: the wave's 128 x 16 tile goes through a wave-private LDS slab in four
// 32-row quarters (lanes 16q..16q+15 own quarter q) and comes back in the column-owner layout of the MFMA operands.
#ifndef KK_BUC_NT
#define KK_BUC_NT 1     // non-temporal Win loads and T / W stores (every element moves once): block step -1.5 % in a same-box A/B
#endif
#if KK_BUC_NT
#define BUC_LD_WIN ld2s
#define BUC_ST st2s
#else
#define BUC_LD_WIN ld2
#define BUC_ST st2
#endif
#define BUC_LD 34     // row stride (doubles) of a slab column: 32 rows + 2 (16-byte aligned, column starts 4 banks apart)
template <int NB>
__global__ __launch_bounds__(KK_TPB, 4) void k_block_update_commit(const double* V, int64_t ld, int m, const double* Win, double* Wout,
                                                                 int64_t ldw, double* Tout, int64_t ldt, int nb,
                                                                 const double* __restrict__ S, int64_t rpb,
                                                                 double* __restrict__ part_nrm, const double* __restrict__ cflag,
                                                                 const double* __restrict__ S1, double* __restrict__ part_g) {
    extern __shared__ __attribute__((aligned(16))) double ssm[];   // [m][NB] coefficients | [64] norm slots | [NB][NB] R1^-1 | 4 slabs [16][BUC_LD]
    double* nsl = ssm + (size_t)m * NB;
    double* s1 = nsl + 64;
    double* slab_all = s1 + NB * NB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* slab = slab_all + wave * (16 * BUC_LD);
    const bool commit = (cflag[0] == 0.0);
    for (int e = tid; e < m * NB; e += KK_TPB) ssm[e] = -S[e];     // w = Win - V S is accumulated ONTO Win (below)
    if (tid < 64) nsl[tid] = 0.0;
    for (int e = tid; e < NB * NB; e += KK_TPB) s1[e] = commit ? S1[e] : 0.0;
    for (int e = tid; e < 4 * 16 * BUC_LD; e += KK_TPB) slab_all[e] = 0.0;    // columns >= NB stay zero
    __syncthreads();
    v4d gacc = v4d{0.0, 0.0, 0.0, 0.0};
    const int cq = lane & 15, kq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    // the first four basis columns of a tile are requested during the LAST column group of the tile before it: the epilogue
    // below (R1^-1 product, slab transposition, MFMAs) would otherwise leave this wave without a load in flight
    d2 xn[4];
    if (m >= 4 && r0 + tid * 2 < r1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) xn[u] = ld2s(V + (int64_t)u * ld + r0 + tid * 2);
    }
    for (int64_t r = r0 + tid * 2; r < r1; r += KK_SUB) {     // (r1 - r0) is a multiple of KK_SUB: every lane of a wave iterates alike
        // the accumulators START as the rows of Win: its 16 loads travel with the first basis columns instead of standing,
        // one after the other, between the main loop and the epilogue (every accumulator stays live for the R1^-1 product,
        // so there is no register to hoist them into: measured 20 % of the kernel)
        d2 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = (j < nb) ? BUC_LD_WIN(Win + (int64_t)j * ldw + r) : d2{0.0, 0.0};
        int c = 0;
        for (; c + 4 <= m; c += 4) {
            d2 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = xn[u];
            if (c + 8 <= m) {
#pragma unroll
                for (int u = 0; u < 4; ++u) xn[u] = ld2s(V + (int64_t)(c + 4 + u) * ld + r);
            } else if (r + KK_SUB < r1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) xn[u] = ld2s(V + (int64_t)u * ld + r + KK_SUB);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const d2* Sc = reinterpret_cast<const d2*>(ssm + (size_t)(c + u) * NB);
#pragma unroll
                for (int j2 = 0; j2 < NB / 2; ++j2) {
                    const d2 sv = Sc[j2];
                    acc[2 * j2].x = fma(sv.x, x[u].x, acc[2 * j2].x); acc[2 * j2].y = fma(sv.x, x[u].y, acc[2 * j2].y);
                    acc[2 * j2 + 1].x = fma(sv.y, x[u].x, acc[2 * j2 + 1].x); acc[2 * j2 + 1].y = fma(sv.y, x[u].y, acc[2 * j2 + 1].y);
                }
            }
        }
        for (; c < m; ++c) {
            const d2 x = ld2(V + (int64_t)c * ld + r);
            const d2* Sc = reinterpret_cast<const d2*>(ssm + (size_t)c * NB);
#pragma unroll
            for (int j2 = 0; j2 < NB / 2; ++j2) {
                const d2 sv = Sc[j2];
                acc[2 * j2].x = fma(sv.x, x.x, acc[2 * j2].x); acc[2 * j2].y = fma(sv.x, x.y, acc[2 * j2].y);
                acc[2 * j2 + 1].x = fma(sv.y, x.x, acc[2 * j2 + 1].x); acc[2 * j2 + 1].y = fma(sv.y, x.y, acc[2 * j2 + 1].y);
            }
        }
        // acc = w = Win - V S; squared column norms
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < nb) {
                const double t = wave_sum(fma(acc[j].x, acc[j].x, acc[j].y * acc[j].y));
                if (lane == 0) nsl[j * 4 + wave] += t;
            }
        }
        if (!commit) {     // uniform for the launch
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (j < nb) BUC_ST(Wout + (int64_t)j * ldw + r, acc[j]);
            continue;
        }
        // t = w R1^-1 in place, last column first (t_j needs w_0 .. w_j only; R1^-1 is upper triangular, row-major in LDS)
#pragma unroll
        for (int j = NB - 1; j >= 0; --j) {
            double tx = 0.0, ty = 0.0;
#pragma unroll
            for (int i = 0; i <= j; ++i) {
                const double sij = s1[i * NB + j];
                tx = fma(acc[i].x, sij, tx); ty = fma(acc[i].y, sij, ty);
            }
            acc[j].x = tx; acc[j].y = ty;
            if (j < nb) BUC_ST(Tout + (int64_t)j * ldt + r, acc[j]);
            __builtin_amdgcn_sched_barrier(0);      // keep the R1^-1 reads of one column together (else all 136 are hoisted: spills)
        }
        // G2 += t' t: quarter q of the wave's 128 rows is owned by lanes 16q .. 16q+15 (two rows each)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (kq == q4) {
#pragma unroll
                for (int j = 0; j < NB; ++j) *reinterpret_cast<d2*>(slab + j * BUC_LD + 2 * cq) = acc[j];
            }
            wave_sync();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const d2 v = *reinterpret_cast<const d2*>(slab + cq * BUC_LD + kq * 8 + 2 * u);
                gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, v.x, gacc, 0, 0, 0);
                gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, v.y, gacc, 0, 0, 0);
            }
            wave_sync();
        }
    }
    __syncthreads();
    if (tid < nb) part_nrm[(int64_t)tid * KK_MAX_BLOCKS + blockIdx.x] = (nsl[tid * 4] + nsl[tid * 4 + 1]) + (nsl[tid * 4 + 2] + nsl[tid * 4 + 3]);
    if (commit) {      // the four waves' tiles in a fixed order, one 256-double partial tile per block (layout of k_block_gram)
        double* red = slab_all;   // all slabs are idle now (the barrier above)
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    double* a = &red[rr * 64 + lane];
                    *a = (w == 0) ? gacc[rr] : (*a + gacc[rr]);
                }
            }
            __syncthreads();
        }
        part_g[(int64_t)blockIdx.x * 256 + tid] = red[tid];
    }
}

template <int NB, bool BZERO, int PF>
__global__ __launch_bounds__(KK_TPB, 2) void k_block_update_pf(const double* V, int64_t ld, int m, const double* Win,
                                                               double* Wout, int64_t ldw_in, int64_t ldw_out, int nb,
                                                               const double* __restrict__ S, double alpha, double beta,
                                                               int64_t rpb, double* __restrict__ part_nrm) {
    __shared__ double nsm[NB * KK_TPB];
    __shared__ double sm[4];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    if (part_nrm) {
#pragma unroll
        for (int j = 0; j < NB; ++j) nsm[j * KK_TPB + tid] = 0.0;
    }
    for (int64_t r = r0 + tid * 2; r < r1; r += KK_SUB) {
        d2 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = d2{0.0, 0.0};
        d2 x[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (u < m) x[u] = ld2s(V + (int64_t)u * ld + r);
        int c = 0;
        for (; c + PF <= m; c += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const d2 xv = x[u];
                if (c + u + PF < m) x[u] = ld2s(V + (int64_t)(c + u + PF) * ld + r);   // uniform branch
                const double* Sc = S + (int64_t)(c + u) * NB;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const double sv = Sc[j];
                    acc[j].x = fma(sv, xv.x, acc[j].x); acc[j].y = fma(sv, xv.y, acc[j].y);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (c + u < m) {
                const d2 xv = x[u];
                const double* Sc = S + (int64_t)(c + u) * NB;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const double sv = Sc[j];
                    acc[j].x = fma(sv, xv.x, acc[j].x); acc[j].y = fma(sv, xv.y, acc[j].y);
                }
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
    return kk_launch_block_gram_rs(ctx, X, ldx, p, Y, ldy, q, ld, C_dev, 1, ldc);
}
int kk_launch_blk_chol1(kk_ctx ctx, const double* G, int p, double abs_min, double* R1, double* S1, int st, double* flag) {
    hipLaunchKernelGGL(k_blk_chol1, dim3(1), dim3(64), 0, ctx->stream, G, p, abs_min, R1, S1, st, flag);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_chol2(kk_ctx ctx, const double* G2, int p, const double* R1, double* B, int ldb, double* S2, double* S3, int st,
                        double* flag) {
    hipLaunchKernelGGL(k_blk_chol2, dim3(1), dim3(64), 0, ctx->stream, G2, p, R1, B, ldb, S2, S3, st, flag, ctx->qr_skip_tol);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_fill_m(kk_ctx ctx, const double* M, int ldm, int p, double* S3, int st) {
    hipLaunchKernelGGL(k_blk_fill_m, dim3(1), dim3(64), 0, ctx->stream, M, ldm, p, S3, st);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_panel_m(kk_ctx ctx, const double* P, int st, int k, int p, double* M, int ldm) {
    hipLaunchKernelGGL(k_blk_panel_m, dim3(1), dim3(KK_TPB), 0, ctx->stream, P, st, k, p, M, ldm);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_onepass_check(kk_ctx ctx, const double* P, int st, int kn, int p, const double* nrm2, double eta, double* flag) {
    hipLaunchKernelGGL(k_blk_onepass_check, dim3(1), dim3(KK_TPB), 0, ctx->stream, P, st, kn, p, nrm2, eta * eta, flag);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_gram_rows(kk_ctx ctx, const double* G2, int st, int k, int p, double* gram, int cap, double* gdiag) {
    hipLaunchKernelGGL(k_blk_gram_rows, dim3(4), dim3(KK_TPB), 0, ctx->stream, G2, st, k, p, gram, cap, gdiag);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_panel_correct(kk_ctx ctx, const double* P, int st, int kn, int p, const double* gram, int cap, double* Pc,
                                const double* gdiag) {
    hipLaunchKernelGGL(k_blk_panel_correct, dim3((kn * st + KK_TPB - 1) / KK_TPB), dim3(KK_TPB), (size_t)kn * st * sizeof(double), ctx->stream, P, st, kn, p, gram, cap, Pc, gdiag);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_resid_gram(kk_ctx ctx, const double* P, const double* Pc, int st, int kn, int p, const double* GYY,
                             const double* nrm2, double* GW) {
    // both panels in dynamic LDS: 2 kn st doubles (64 KB at kn = 256, st = 16) next to the static 16 x 17 tile -- beyond the
    // 64 KB a kernel gets without asking.  The opt-in is a per-device attribute of the function
    const size_t dyn = (size_t)2 * kn * st * sizeof(double);
    if (dyn + 4096 > 160 * 1024) {
        kk_set_error("kk_launch_blk_resid_gram: panels of %d x %d do not fit the LDS", kn, st);
        return KK_ERR_UNSUPPORTED;
    }
    static bool configured[KK_MAX_DEVICES] = {};
    const int dev = ctx->device;
    if (dyn > 48 * 1024 && (dev < 0 || dev >= KK_MAX_DEVICES || !configured[dev])) {
        KK_HIP(hipSetDevice(ctx->device));
        KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_blk_resid_gram), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        if (dev >= 0 && dev < KK_MAX_DEVICES) configured[dev] = true;
    }
    hipLaunchKernelGGL(k_blk_resid_gram, dim3(1), dim3(KK_TPB), dyn, ctx->stream, P, Pc, st, kn, p, GYY, nrm2, GW);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_blk_combine(kk_ctx ctx, double* P, const double* S3, int kn, int nz, int st) {
    hipLaunchKernelGGL(k_blk_combine, dim3(1), dim3(KK_TPB), 0, ctx->stream, P, S3, kn, nz, st);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
// Grid of a Gram panel kernel.  A wave keeps 4 KB per 16-column group in flight, so narrow panels need many resident waves
// to cover the HBM latency (p = 16 at 2 blocks per CU: 3.1 TB/s); every block leaves an NG*256-double partial tile, which
// bounds the grid from above.
static void gram_grid(kk_ctx ctx, int64_t ld, int NG, int cap_bpc, int* nblk_out, int64_t* rpb_out) {
    kk_part pt = kk_partition(ctx, ld);
    int nblk = pt.nblk;
    int64_t rpb = pt.rpb;
    int bpc = ctx->gram_bpc / NG;
    const int floor_bpc = NG <= 4 ? (ctx->gram_bpc < 4 ? ctx->gram_bpc : 4) : 2;   // NG = 8: 188 VGPRs, two waves per SIMD anyway
    if (bpc < floor_bpc) bpc = floor_bpc;
    if (bpc > cap_bpc) bpc = cap_bpc;
    const int maxb = bpc * ctx->num_cus;
    if (nblk > maxb) {
        const int64_t nsub = ld / KK_SUB;
        const int64_t spb = (nsub + maxb - 1) / maxb;
        rpb = spb * KK_SUB;
        nblk = (int)((nsub + spb - 1) / spb);
    }
    *nblk_out = nblk;
    *rpb_out = rpb;
}
// C (p x q, strides rs / cs) = X' T with the tile T = beta*Yin + alpha * Z S formed on the fly (k_block_gram_tile);
// X == nullptr: C = T' T (p = q).  Yout != nullptr: T is also written there.
int kk_launch_block_gram_tile(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Yin, int64_t ldy, const double* Z,
                              int64_t ldz, int nz, const double* S_dev, int st, double alpha, double beta, double* Yout,
                              int64_t ldyo, int q, int64_t ld, double* C_dev, int rs, int cs) {
    if (p <= 0 || q <= 0) return KK_OK;
    if (p > 128 || q > 16 || nz > 32) { kk_set_error("kk_launch_block_gram_tile: p=%d q=%d nz=%d exceed one launch", p, q, nz); return KK_ERR_INVALID; }
    const int ng = (p + 15) / 16;
    int NG = 1;
    while (NG < ng) NG *= 2;
    int nblk;
    int64_t rpb;
    gram_grid(ctx, ld, NG, 6, &nblk, &rpb);   // 24.5 KB of LDS per block
    const size_t shm = ((size_t)NG * 256 + 32 * 16 + 4 * 16 * BGT_LD) * sizeof(double);
    double* part = ctx->partials;
    const bool xt = (X == nullptr), store = (Yout != nullptr), hasy = (Yin != nullptr && beta != 0.0);
    {
        kk_prof_scope ps(ctx, "k_block_gram");
        dim3 g(nblk), b(KK_TPB);
#define GT_ARGS X, ldx, p, Yin, ldy, Z, ldz, nz, S_dev, st, alpha, beta, Yout, ldyo, q, ld, rpb, part
        if (xt) {
            if (store) hipLaunchKernelGGL((k_block_gram_tile<1, true, true, false>), g, b, shm, ctx->stream, GT_ARGS);
            else hipLaunchKernelGGL((k_block_gram_tile<1, true, false, false>), g, b, shm, ctx->stream, GT_ARGS);
        } else if (hasy && !store) {
            switch (NG) {
                case 1: hipLaunchKernelGGL((k_block_gram_tile<1, false, false, true>), g, b, shm, ctx->stream, GT_ARGS); break;
                case 2: hipLaunchKernelGGL((k_block_gram_tile<2, false, false, true>), g, b, shm, ctx->stream, GT_ARGS); break;
                case 4: hipLaunchKernelGGL((k_block_gram_tile<4, false, false, true>), g, b, shm, ctx->stream, GT_ARGS); break;
                default: hipLaunchKernelGGL((k_block_gram_tile<8, false, false, true>), g, b, shm, ctx->stream, GT_ARGS); break;
            }
        } else {
            kk_set_error("kk_launch_block_gram_tile: unsupported combination");
            return KK_ERR_UNSUPPORTED;
        }
#undef GT_ARGS
    }
    KK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_finalize_gram, dim3(NG * 16), dim3(KK_TPB), 0, ctx->stream, part, nblk, NG, p, q,
                       C_dev, rs, cs);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
// general output strides: C[i*rs + j*cs]
int kk_launch_block_gram_rs(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
                            double* C_dev, int rs, int cs) {
    if (p <= 0 || q <= 0) return KK_OK;
    if (p > 128 || q > 16) { kk_set_error("kk_launch_block_gram: p=%d q=%d exceed one launch (128 x 16)", p, q); return KK_ERR_INVALID; }
    const int ng = (p + 15) / 16;
    int NG = 1;
    while (NG < ng) NG *= 2;
    int nblk;
    int64_t rpb;
    gram_grid(ctx, ld, NG, 8, &nblk, &rpb);
    const size_t shm = (size_t)NG * 256 * sizeof(double);
    double* part = ctx->partials;
    {
        kk_prof_scope ps(ctx, "k_block_gram");
        dim3 g(nblk), b(KK_TPB);
#define BG_ARGS X, ldx, p, Y, ldy, q, ld, rpb, part, ctx->gram_nt
        if (NG == 1 && X == Y && ldx == ldy && p == q) {
            hipLaunchKernelGGL((k_block_gram<1, true>), g, b, shm, ctx->stream, BG_ARGS);
        } else {
            switch (NG) {
                case 1: hipLaunchKernelGGL((k_block_gram<1, false>), g, b, shm, ctx->stream, BG_ARGS); break;
                case 2: hipLaunchKernelGGL((k_block_gram<2, false>), g, b, shm, ctx->stream, BG_ARGS); break;
                case 4: hipLaunchKernelGGL((k_block_gram<4, false>), g, b, shm, ctx->stream, BG_ARGS); break;
                default: hipLaunchKernelGGL((k_block_gram<8, false>), g, b, shm, ctx->stream, BG_ARGS); break;
            }
        }
#undef BG_ARGS
    }
    KK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_finalize_gram, dim3(NG * 16), dim3(KK_TPB), 0, ctx->stream, part, nblk, NG, p, q,
                       C_dev, rs, cs);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

// C = X' Y and C2 = X' Y2 in one pass (both row-major panels: C[i*rs + j], C2[i*rs2 + j])
int kk_launch_block_gram2(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, const double* Y2,
                          int64_t ldy2, int q2, int64_t ld, double* C_dev, int rs, double* C2_dev, int rs2, double* C3_dev) {
    if (p <= 0 || q <= 0 || q2 <= 0) return KK_OK;
    if (p > 128 || q > 16 || q2 > 16) { kk_set_error("kk_launch_block_gram2: p=%d q=%d q2=%d exceed one launch (128 x 16)", p, q, q2); return KK_ERR_INVALID; }
    const int ng = (p + 15) / 16;
    const int NG = ng <= 4 ? ng : (ng == 5 ? 5 : 8);   // 48-column chunks (kn = 96, 112) run three groups, not four with one empty
    int nblk;
    int64_t rpb;
    // ride-along block = a whole group of X?  (same leading dimension, starts on a 16-column boundary of this launch, 16 wide)
    int gx = -1;
    if (ldy2 == ldx && q2 == 16 && Y2 >= X) {
        const int64_t off = Y2 - X;
        if (off % ldx == 0 && (off / ldx) % 16 == 0 && off / ldx + 16 <= p) gx = (int)(off / ldx / 16);
    }
    const size_t shm = (size_t)NG * 256 * sizeof(double);
    // the pipelined kernel addresses every 16-column group through one 32-bit buffer descriptor
    const int64_t lim = (int64_t)1 << 31;
    const bool pipe = ctx->gram2_pipe && NG <= 5 && (15 * ldx + ld) * 8 < lim && ((int64_t)(q - 1) * ldy + ld) * 8 < lim &&
                      ((int64_t)(q2 - 1) * ldy2 + ld) * 8 < lim && ld % KK_SUB == 0;
    const bool gxl = pipe && gx == NG - 1 && p == NG * 16;   // the pipelined kernel aliases the ride-along tile only in the common last-group case
    if (pipe) {
        // the pipelined kernel runs as ONE resident wave of blocks: as many per CU as its register budget admits
        // resident blocks per CU of the instantiation (one 256-thread block = one wave per SIMD; register counts of the
        // gfx950 build: NG = 1: 92-136, 2: 144-184, 3: 196-236, 4: 244-292, 5: 300-344 of the 512 per SIMD lane)
        const int o = NG == 1 ? 3 : (NG <= 3 ? 2 : (NG == 4 && gxl ? 2 : 1));
        int bpc = ctx->gram2_bpc > 0 ? ctx->gram2_bpc : o;
        const int64_t nsub = ld / KK_SUB;
        int64_t maxb = (int64_t)bpc * ctx->num_cus;
        if (maxb > KK_MAX_BLOCKS) maxb = KK_MAX_BLOCKS;
        const int64_t spb = std::max<int64_t>(1, (nsub + maxb - 1) / maxb);
        rpb = spb * KK_SUB;
        nblk = (int)std::max<int64_t>(1, (nsub + spb - 1) / spb);
    } else {
        // NG >= 5: two blocks per CU fit (VGPRs); the partial tiles double (+ one tile per block for Y'Y), which bounds the grid
        gram_grid(ctx, ld, NG == 5 ? 8 : (NG == 3 ? 4 : NG), NG >= 5 ? 2 : (NG >= 3 ? 3 : 8), &nblk, &rpb);
    }
    if ((2 * (int64_t)NG + 1) * nblk * 256 > (int64_t)(2 * KK_MAX_M + 8) * KK_MAX_BLOCKS) { kk_set_error("kk_launch_block_gram2: partial buffer too small"); return KK_ERR_INVALID; }
    double* part = ctx->partials;
    double* part2 = part + (int64_t)nblk * NG * 256;
    double* part3 = C3_dev ? part2 + (int64_t)nblk * NG * 256 : nullptr;
    {
        kk_prof_scope ps(ctx, "k_block_gram");
        dim3 g(nblk), b(KK_TPB);
#define BG2_ARGS X, ldx, p, Y, ldy, q, Y2, ldy2, q2, gx, ld, rpb, part, part2, part3
#define G2P_ARGS X, ldx, p, Y, ldy, q, Y2, ldy2, q2, ld, rpb, part, part2, part3
#define G2P_CASE(N) case N: if (part3) { if (gxl) hipLaunchKernelGGL((k_block_gram2p<N, true, true>), g, b, shm, ctx->stream, G2P_ARGS); \
                                         else hipLaunchKernelGGL((k_block_gram2p<N, true, false>), g, b, shm, ctx->stream, G2P_ARGS); } \
                            else { if (gxl) hipLaunchKernelGGL((k_block_gram2p<N, false, true>), g, b, shm, ctx->stream, G2P_ARGS); \
                                   else hipLaunchKernelGGL((k_block_gram2p<N, false, false>), g, b, shm, ctx->stream, G2P_ARGS); } break;
        if (pipe) switch (NG) { G2P_CASE(1) G2P_CASE(2) G2P_CASE(3) G2P_CASE(4) default: G2P_CASE(5) }
        else switch (NG) {
            case 1: hipLaunchKernelGGL((k_block_gram2<1>), g, b, shm, ctx->stream, BG2_ARGS); break;
            case 2: hipLaunchKernelGGL((k_block_gram2<2>), g, b, shm, ctx->stream, BG2_ARGS); break;
            case 3: hipLaunchKernelGGL((k_block_gram2<3>), g, b, shm, ctx->stream, BG2_ARGS); break;
            case 4: hipLaunchKernelGGL((k_block_gram2<4>), g, b, shm, ctx->stream, BG2_ARGS); break;
            case 5: hipLaunchKernelGGL((k_block_gram2<5>), g, b, shm, ctx->stream, BG2_ARGS); break;
            default: hipLaunchKernelGGL((k_block_gram2<8>), g, b, shm, ctx->stream, BG2_ARGS); break;
        }
#undef G2P_CASE
#undef G2P_ARGS
#undef BG2_ARGS
    }
    KK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_finalize_gram3, dim3(2 * NG * 16 + (C3_dev ? 16 : 0)), dim3(KK_TPB), 0, ctx->stream, part, part2, part3, nblk, NG, p, q, q2,
                       C_dev, rs, C2_dev, rs2, C3_dev);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_blk_commit_prep(kk_ctx ctx, const double* GWE, const double* GYY, int p, double abs_min, double keep, double* R1, double* S1,
                              int st, double* cflag) {
    hipLaunchKernelGGL(k_blk_commit_prep, dim3(1), dim3(64), 0, ctx->stream, GWE, GYY, p, abs_min, keep, R1, S1, st, cflag);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
// residual update of the one-pass block step with the normalised commit (k_block_update_commit): W = Win - V S with its
// squared column norms (norms2_dev, all-reduced); cflag[0] == 0 on the device: T = W R1^-1 -> Tout and G2 = T'T -> G2_dev
// (p x p column-major, NOT all-reduced: the consumer does that), else W -> Wout (may alias Win).
int kk_launch_block_update_commit(kk_ctx ctx, const double* V, int64_t ld, int m, const double* Win, double* Wout, int64_t ldw, double* Tout,
                                  int64_t ldt, int nb, const double* S_dev, double* norms2_dev, const double* cflag_dev,
                                  const double* S1_dev, double* G2_dev) {
    if (nb <= 0 || nb > 16) { kk_set_error("kk_launch_block_update_commit: nb=%d", nb); return KK_ERR_INVALID; }
    const int NB = kk_bu_stride(nb);
    kk_part p = kk_partition(ctx, ld);
    if (ld % KK_SUB != 0 || (int64_t)p.nblk * 256 > (int64_t)(2 * KK_MAX_M + 8 - 16) * KK_MAX_BLOCKS) {
        kk_set_error("kk_launch_block_update_commit: layout not supported"); return KK_ERR_UNSUPPORTED;
    }
    const size_t shm = ((size_t)m * NB + 64 + NB * NB + 4 * 16 * BUC_LD) * sizeof(double);
    if (shm > 64 * 1024) { kk_set_error("kk_launch_block_update_commit: %d x %d panel does not fit", m, NB); return KK_ERR_UNSUPPORTED; }
    double* part = ctx->partials;
    double* part_g = ctx->partials + (int64_t)16 * KK_MAX_BLOCKS;
    dim3 g(p.nblk), b(KK_TPB);
    {
        kk_prof_scope ps(ctx, "k_block_update");
#define BUC_ARGS V, ld, m, Win, Wout, ldw, Tout, ldt, nb, S_dev, p.rpb, part, cflag_dev, S1_dev, part_g
        if (NB == 4) hipLaunchKernelGGL((k_block_update_commit<4>), g, b, shm, ctx->stream, BUC_ARGS);
        else if (NB == 8) hipLaunchKernelGGL((k_block_update_commit<8>), g, b, shm, ctx->stream, BUC_ARGS);
        else hipLaunchKernelGGL((k_block_update_commit<16>), g, b, shm, ctx->stream, BUC_ARGS);
#undef BUC_ARGS
    }
    KK_HIP(hipGetLastError());
    KK_TRY(finalize_rows(ctx, part, p.nblk, nb, norms2_dev, nullptr));
    KK_TRY(kk_allreduce(ctx, norms2_dev, nb));
    // (when the flag said "no commit" the partial tiles were not written: the finalize then sums stale data into G2_dev,
    //  which nobody reads -- the host learns the flag with the step's read-back)
    hipLaunchKernelGGL(k_finalize_gram, dim3(16), dim3(KK_TPB), 0, ctx->stream, part_g, p.nblk, 1, nb, nb, G2_dev, 1, nb);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

// Wout[:, j] = beta*Win[:, j] + alpha * sum_c V[:, c] S_dev[c*nb + j], j < nb <= 16; optional norms2_dev[nb]
int kk_launch_block_update(kk_ctx ctx, const double* V, int64_t ld, int m, const double* Win, double* Wout, int64_t ldw_in,
                           int64_t ldw_out, int nb, const double* S_dev, double alpha, double beta, double* norms2_dev,
                           const double* skip_dev) {
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
#define BU_PF(NBT, PFD) \
        if (bz) hipLaunchKernelGGL((k_block_update_pf<NBT, true, PFD>), g, b, 0, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, alpha, beta, p.rpb, part); \
        else hipLaunchKernelGGL((k_block_update_pf<NBT, false, PFD>), g, b, 0, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, alpha, beta, p.rpb, part);
        const size_t shm = ((size_t)m * kk_bu_stride(nb) + 64) * sizeof(double);
#define BU_LDS(NBT) \
        if (bz) hipLaunchKernelGGL((k_block_update_lds<NBT, true>), g, b, shm, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, alpha, beta, p.rpb, part, skip_dev); \
        else hipLaunchKernelGGL((k_block_update_lds<NBT, false>), g, b, shm, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, alpha, beta, p.rpb, part, skip_dev);
        if (ctx->bu_mfma && m >= 4 && ((size_t)((m + 3) / 4) * 64 + 64) * sizeof(double) <= 60 * 1024) {
            // MFMA form (any nb <= 16: the panel stride is that of the VALU kernels)
            const size_t shm2 = ((size_t)((m + 3) / 4) * 64 + 64) * sizeof(double);
            const int sst = kk_bu_stride(nb);
#define BU_MFMA(TRV, PFV, MB) \
            if (bz) hipLaunchKernelGGL((k_block_update_mfma<true, TRV, PFV, MB>), g, b, shm2, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, sst, alpha, beta, p.rpb, part, skip_dev); \
            else hipLaunchKernelGGL((k_block_update_mfma<false, TRV, PFV, MB>), g, b, shm2, ctx->stream, V, ld, m, Win, Wout, ldw_in, ldw_out, nb, S_dev, sst, alpha, beta, p.rpb, part, skip_dev);
            switch (ctx->bu_mfma) {   // tile shapes (tools/bu_mfma_check.py)
                case 2: { BU_MFMA(2, 8, 2) } break;
                case 3: { BU_MFMA(4, 4, 2) } break;
                case 4: { BU_MFMA(1, 8, 4) } break;
                case 5: { BU_MFMA(2, 4, 4) } break;
                case 6: { BU_MFMA(1, 16, 2) } break;
                default: { BU_MFMA(2, 4, 2) } break;
            }
#undef BU_MFMA
        }
        else if (ctx->bu_prefetch == 1 && m * kk_bu_stride(nb) <= 8192) { if (nb <= 4) { BU_LDS(4) } else if (nb <= 8) { BU_LDS(8) } else { BU_LDS(16) } }
        else if (nb > 8 && ctx->bu_prefetch == 16) { BU_PF(16, 16) }
        else if (nb > 8 && ctx->bu_prefetch == 8) { BU_PF(16, 8) }
        else if (nb > 8 && ctx->bu_prefetch == 24) { BU_PF(16, 24) }
        else if (nb <= 4) { BU_CASE(4) } else if (nb <= 8) { BU_CASE(8) } else { BU_CASE(16) }
#undef BU_PF
#undef BU_LDS
#undef BU_CASE
    }
    KK_HIP(hipGetLastError());
    if (norms2_dev) {
        KK_TRY(finalize_rows(ctx, part, p.nblk, nb, norms2_dev, nullptr));
        KK_TRY(kk_allreduce(ctx, norms2_dev, nb));
    }
    return KK_OK;
}

