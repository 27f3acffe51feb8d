// gfx950 sparse kernels: SpMV on ELL / CSR / SELL-64-sigma / column-tiled SELL with the fused Krylov epilogue, and SpMM
// on ELL for the block path.
#include "kk_device.h"


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
    int nt_store;        // y is written with non-temporal stores: long vectors, whose next reader streams them anyway (the sweep
                         // kernel that follows loses 1.4 % at 10M rows, same-box A/B: headline +1.1 %, general-format leg +1.1 %);
                         // short vectors stay in the L2 for their reader
};

// gathered element of x; column indices >= n_local address the ghost buffer.  BRANCH-FREE (the address is selected, the
// load is unconditional): with `if (ghost) return ghost[..]; return x[c];` every gather sits in its own basic block and
// hipcc waits for it (`s_waitcnt vmcnt(0)`) before the next one is issued -- the gathers of an unrolled group then take
// one memory round trip EACH instead of one together.
__device__ __forceinline__ double xload(const double* __restrict__ x, const spmv_epi& e, int c) {
    const bool g = e.n_local >= 0 && c >= e.n_local;
    const double* p = g ? e.ghost + (c - e.n_local) : x + c;
    return *p;
}

__global__ __launch_bounds__(KK_TPB) void k_spmv_ell(const int32_t* __restrict__ ecol, const double* __restrict__ eval,
                                                     int64_t ell_ld, int width, int64_t nrows,
                                                     const double* __restrict__ x, double* __restrict__ y, spmv_epi e,
                                                     int nb_logical, double* __restrict__ part_dot,
                                                     double* __restrict__ part_nrm, int64_t row_base, int64_t row_end) {
    __shared__ double sm[4];
    // rows [row_base, row_end) of the operator (the whole matrix, or a boundary strip of a row-sharded stencil; row_base even)
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
        const int64_t row = row_base + ((int64_t)lb * KK_TPB + threadIdx.x) * 2;
        if (row < row_end) {  // ell_ld is even and >= nrows; pad entries have val 0, col 0
            double s0 = 0, s1 = 0;
            const int32_t* cp = ecol + row;
            const double* vp = eval + row;
            int k = 0;
            for (; k + 4 <= width; k += 4) {   // four slots at a time: 8 matrix loads, then 8 gathers, then the products (slot order)
                int2 cc[4]; d2 v[4]; double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { cc[u] = ldi2s(cp + (int64_t)(k + u) * ell_ld); v[u] = ld2s(vp + (int64_t)(k + u) * ell_ld); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { xa[u] = xload(x, e, cc[u].x); xb[u] = xload(x, e, cc[u].y); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s0 = fma(v[u].x, xa[u], s0); s1 = fma(v[u].y, xb[u], s1); }
            }
            for (; k < width; ++k) {
                const int2 cc = ldi2s(cp + (int64_t)k * ell_ld);
                const d2 v = ld2s(vp + (int64_t)k * ell_ld);
                const double xa = xload(x, e, cc.x), xb = xload(x, e, cc.y);
                s0 = fma(v.x, xa, s0);
                s1 = fma(v.y, xb, s1);
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
            if (e.nt_store) st2s(y + row, out); else st2(y + row, out);
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

// SpMV on the dense diagonals of a grid stencil (kk_sparse_dev::dia_*): no column indices (40 instead of 60 matrix bytes per
// row for a 5-point operator), the neighbours are shifted contiguous loads instead of gathers; same XCD banding, same fused
// epilogue as k_spmv_ell.
struct dia_offs { int64_t o[9]; };
// One pair x[idx], x[idx+1] of a shifted diagonal read, BRANCH-FREE: an index outside [0, nrows) reads x[0] instead and the
// value is replaced by 0 afterwards.  (Round 2 / 3 tested the range first and loaded inside the branches: hipcc then puts
// `s_waitcnt vmcnt(0)` behind every one of those loads -- five dependent memory round trips per block iteration, which
// capped the constant-coefficient apply at 2.6-2.8 TB/s as soon as the vectors no longer fit the Infinity Cache,
// tools/stencil_shape_sweep.py.)  Every load of an iteration is issued before the first value is used.
__device__ __forceinline__ d2 dia_pair(const double* __restrict__ x, int64_t idx, int64_t nrows) {
    const bool ok0 = (unsigned long long)idx < (unsigned long long)nrows;
    const bool ok1 = (unsigned long long)(idx + 1) < (unsigned long long)nrows;
    const double a = x[ok0 ? idx : 0], b = x[ok1 ? idx + 1 : 0];
    return d2{ok0 ? a : 0.0, ok1 ? b : 0.0};
}
// CONST: constant-coefficient stencil (kk_sparse_dev::dia_const) -- the coefficient of slot q is cst.c[q] wherever the
// neighbour sits on the same grid line, 0 where a +-1 shift would wrap to the next line; no diagonal is read at all.
// (struct dia_cst {c[9], phase, D}: kk_internal.h -- shared with the sweep kernels that apply the stencil themselves)
// ALIGNED (5-point stencils with an EVEN far offset D, vectors below 4 GB): the load count per row pair drops from eight (two
// 16-byte + six 8-byte) to four 16-byte ones --
//   * x[row +- D], x[row + 1 +- D] are ONE aligned pair each (row and D even);
//   * x[row - 1] and x[row + 2] are the neighbouring lanes' centre pairs: a DPP wave shift instead of a load; only lane 0 /
//     lane 63 of a wave fetch theirs, through a buffer descriptor whose out-of-range offsets switch the other 63 lanes' loads off
//     in the address unit (no branch: a load inside a branch is waited for at its end, see dia_pair).
// 8-byte accesses run at 0.54-0.70 of the 16-byte rate (MI355X_MICROARCH.md): the value-free apply of config 2 moved 24 N bytes
// at 5.0 TB/s with them (r4, 0.63 of peak).  Same operands in the same order: bit-identical results.
typedef unsigned dia_v2u __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double dia_wave_shift(double v) {   // 0x138: lane i takes lane i - 1; 0x130: lane i takes lane i + 1 (edge lane keeps its own)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int PTS, int U, bool CONST, bool ALIGNED = false>   // U row pairs per lane: a block covers U consecutive 512-row chunks, all their loads in flight together
__global__ __launch_bounds__(KK_TPB) void k_spmv_dia(const double* __restrict__ dval, int64_t dld, dia_offs offs, int64_t nrows,
                                                     const double* __restrict__ x, double* __restrict__ y, spmv_epi e,
                                                     int nb_logical, double* __restrict__ part_dot,
                                                     double* __restrict__ part_nrm, int64_t row_base, int64_t row_end, dia_cst cst) {
    __shared__ double sm[4];
    // rows [row_base, row_end): the whole operator, or the ghost-free interior of a row-sharded stencil (row_base even)
    const int per = (nb_logical + 7) >> 3;
    const int nbx = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    constexpr int QC = PTS / 2;          // the middle slot is the main diagonal
    double dacc = 0, nacc = 0;
    const double xs = e.xs_dev ? *e.xs_dev : 1.0;
    const double bp = e.vprev ? (e.bprev_dev ? *e.bprev_dev : e.bprev) : 0.0;
    for (int c = blockIdx.x >> 3; c < per; c += nbx) {
        const int lb = xcd * per + c;
        if (lb >= nb_logical) break;
        const int64_t row0 = row_base + ((int64_t)lb * U * KK_TPB + threadIdx.x) * 2;
        // ---- phase 1: every load of this iteration (a lane whose rows lie beyond row_end loads row 0 and stores nothing)
        d2 xv[U][PTS], dv[U][CONST ? 1 : PTS], pv[U], zv[U];
        double el[U], er[U];   // ALIGNED: x[row - 1] of lane 0, x[row + 2] of lane 63
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + (int64_t)u * 2 * KK_TPB;
            // dia_ld, ld are even and >= nrows: the pair (rl, rl + 1) is inside the arrays.  (ALIGNED: clamped by the operator, not by
            // row_end -- a lane beyond the end of a row-sharded interior still holds what its left neighbour needs)
            const int64_t rl = ALIGNED ? (row < nrows ? row : 0) : (row < row_end ? row : 0);
            xv[u][QC] = ld2(x + rl);                         // centre pair: aligned (row is even); pad rows hold zeros
            if (ALIGNED) {
                const int lane = threadIdx.x & 63;
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(nrows * 8), 0x00020000);
                const dia_v2u l2 = __builtin_amdgcn_raw_buffer_load_b64(rx, (lane == 0 && rl >= 1) ? (unsigned)((rl - 1) * 8) : 0xfffffff0u, 0, 0);
                const dia_v2u r2 = __builtin_amdgcn_raw_buffer_load_b64(rx, (lane == 63 && rl + 2 < nrows) ? (unsigned)((rl + 2) * 8) : 0xfffffff0u, 0, 0);
                el[u] = __hiloint2double((int)l2.y, (int)l2.x);
                er[u] = __hiloint2double((int)r2.y, (int)r2.x);
#pragma unroll
                for (int q = 0; q < PTS; q += PTS - 1) {     // the two far slots (PTS == 5: q = 0 and 4)
                    const int64_t idx = rl + offs.o[q];
                    const bool ok = (unsigned long long)idx < (unsigned long long)nrows;
                    const d2 pr = ld2(x + (ok ? idx : 0));
                    xv[u][q] = d2{ok ? pr.x : 0.0, (ok && idx + 1 < nrows) ? pr.y : 0.0};
                }
            }
#pragma unroll
            for (int q = 0; q < PTS; ++q) {
                if (q == QC || ALIGNED) continue;
                // the +-1 neighbours inside the line share one element with the centre pair: one new load each
                if (q == QC - 1) { const bool ok = rl >= 1; const double a = x[ok ? rl - 1 : 0]; xv[u][q] = d2{ok ? a : 0.0, 0.0}; }
                else if (q == QC + 1) { const bool ok = rl + 2 < nrows; const double b = x[ok ? rl + 2 : 0]; xv[u][q] = d2{0.0, ok ? b : 0.0}; }
                else xv[u][q] = dia_pair(x, rl + offs.o[q], nrows);
            }
            if (!CONST) {
#pragma unroll
                for (int q = 0; q < PTS; ++q) dv[u][q] = ld2s(dval + (int64_t)q * dld + rl);
            }
            pv[u] = e.vprev ? ld2(e.vprev + rl) : d2{0.0, 0.0};
            zv[u] = e.dot_mode == 3 ? ld2(e.dvec + rl) : d2{0.0, 0.0};
        }
        // ---- phase 2: products in slot order (the order of round 2: bit-identical results), epilogue, store
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = row0 + (int64_t)u * 2 * KK_TPB;
            if (ALIGNED) {   // the +-1 neighbours inside the line: the centre pairs of the lanes next door
                const int lane = threadIdx.x & 63;
                const int64_t rl = row < nrows ? row : 0;
                const double fl = dia_wave_shift<0x138>(xv[u][QC].y), fr = dia_wave_shift<0x130>(xv[u][QC].x);
                xv[u][QC - 1].x = lane == 0 ? el[u] : fl;                                     // (lane 0 with rl == 0: the switched-off load returned 0)
                xv[u][QC + 1].y = lane == 63 ? er[u] : (rl + 2 < nrows ? fr : 0.0);
            }
            xv[u][QC - 1].y = xv[u][QC].x;                   // x[row] is the right element of the "-1" pair ...
            xv[u][QC + 1].x = xv[u][QC].y;                   // ... and x[row + 1] the left element of the "+1" pair
            if (row + 1 >= nrows) { xv[u][QC + 1].x = 0.0; } // (row + 1 outside the operator: the old range test gave 0; the pad row holds 0 anyway)
            double s0 = 0, s1 = 0;
            // CONST: position of the first row of the pair inside its grid line (rows and D are below 2^31: 32-bit remainder)
            const int64_t i0 = CONST ? (int64_t)((unsigned)(row + cst.phase) % (unsigned)cst.D) : 0;
            const int64_t i1 = (i0 + 1 == cst.D) ? 0 : i0 + 1;
#pragma unroll
            for (int q = 0; q < PTS; ++q) {
                const int bq = PTS == 5 ? (q == 1 ? -1 : (q == 3 ? 1 : 0)) : q % 3 - 1;   // shift inside the grid line
                d2 v;
                if (CONST) {
                    v.x = (bq < 0 && i0 == 0) || (bq > 0 && i0 == cst.D - 1) ? 0.0 : cst.c[q];
                    v.y = (bq < 0 && i1 == 0) || (bq > 0 && i1 == cst.D - 1) ? 0.0 : cst.c[q];
                } else {
                    v = dv[u][q];
                }
                s0 = fma(v.x, xv[u][q].x, s0);
                s1 = fma(v.y, xv[u][q].y, s1);
            }
            if (row < row_end) {
                const double t0 = s0 * xs, t1 = s1 * xs;
                d2 out{e.a1 * t0, e.a1 * t1};
                const d2 xc{xv[u][QC].x * xs, xv[u][QC].y * xs};
                if (e.a0 != 0.0) { out.x = fma(e.a0, xc.x, out.x); out.y = fma(e.a0, xc.y, out.y); }
                if (e.dot_mode == 1) { dacc = fma(xc.x, out.x, dacc); dacc = fma(xc.y, out.y, dacc); }
                if (e.vprev) { out.x = fma(-bp, pv[u].x, out.x); out.y = fma(-bp, pv[u].y, out.y); }
                if (row + 1 >= nrows) out.y = 0.0;  // odd nrows: keep the pad row zero
                if (e.dot_mode == 2) { dacc = fma(xc.x, out.x, dacc); dacc = fma(xc.y, out.y, dacc); }
                if (e.dot_mode == 3) { dacc = fma(zv[u].x, out.x, dacc); dacc = fma(zv[u].y, out.y, dacc); }
                if (e.want_nrm) { nacc = fma(out.x, out.x, nacc); nacc = fma(out.y, out.y, nacc); }
                if (e.nt_store) st2s(y + row, out); else st2(y + row, out);
            }
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
            const double x0 = xload(x, e, c0), x1 = xload(x, e, c1), x2 = xload(x, e, c2), x3 = xload(x, e, c3);
            s0 = fma(v0, x0, s0);
            s1 = fma(v1, x1, s1);
            s0 = fma(v2, x2, s0);
            s1 = fma(v3, x3, s1);
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
// R = 256-row rounds per sorting window (window = R * 256 rows = 4 R chunks; wave w takes chunks w, w + 4, ...): the wider
// the window the less padding -- 21 % of all slots with 256-row windows on the config-4 operator, 6 % with 1024-row windows
// (12 % less matrix memory) -- while the rows of a window are still finished coalesced from LDS.  The time follows only
// weakly (-2.4 %): padding slots re-read x[0] from the L1, and what bounds these applies is the rate of L2 requests, one
// per REAL nonzero (switching the padding lanes off for the gather changed nothing either).
template <int R>
__global__ __launch_bounds__(KK_TPB) void k_spmv_sellw(const int64_t* __restrict__ coff, const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ scol, const double* __restrict__ sval,
                                                       int64_t nchunks, int64_t nrows, const double* __restrict__ x,
                                                       double* __restrict__ y, spmv_epi e, double* __restrict__ part_dot,
                                                       double* __restrict__ part_nrm) {
    constexpr int W = R * KK_TPB;
    __shared__ double res[W];
    __shared__ double sm[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double dacc = 0, nacc = 0;
    const double xs = e.xs_dev ? *e.xs_dev : 1.0;
    const double bp = e.vprev ? (e.bprev_dev ? *e.bprev_dev : e.bprev) : 0.0;
    const int64_t nwin = (nchunks + 4 * R - 1) / (4 * R);
    // the chunk descriptors of the NEXT window are fetched while the current one is processed, and the old y of the
    // accumulating tiles is requested before the gathers: two of the four dependent memory round trips per window go
    int64_t win = blockIdx.x;
    int64_t off[R], offn[R];
    int32_t prow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        off[r] = 0; offn[r] = 0; prow[r] = -1;
        const int64_t c = win * 4 * R + 4 * r + wave;
        if (win < nwin && c < nchunks) { off[r] = coff[c]; offn[r] = coff[c + 1]; prow[r] = perm[c * 64 + lane]; }
    }
    for (; win < nwin; win += gridDim.x) {
        double yold[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = win * W + r * KK_TPB + tid;
            yold[r] = (e.acc >= 2 && row < nrows) ? y[row] : 0.0;
        }
        const int64_t wnext = win + gridDim.x;
        int64_t off2[R], offn2[R];
        int32_t prow2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            off2[r] = 0; offn2[r] = 0; prow2[r] = -1;
            const int64_t c = wnext * 4 * R + 4 * r + wave;
            if (wnext < nwin && c < nchunks) { off2[r] = coff[c]; offn2[r] = coff[c + 1]; prow2[r] = perm[c * 64 + lane]; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t c = win * 4 * R + 4 * r + wave;
            if (c < nchunks) {
                const int w = (int)((offn[r] - off[r]) >> 6);
                const int32_t* cp = scol + off[r] + lane;
                const double* vp = sval + off[r] + lane;
                double s0 = 0, s1 = 0;
                int k = 0;
                for (; k + 4 <= w; k += 4) {
                    const int c0 = ldc(cp + (k + 0) * 64), c1 = ldc(cp + (k + 1) * 64), c2 = ldc(cp + (k + 2) * 64), c3 = ldc(cp + (k + 3) * 64);
                    const double v0 = ldv(vp + (k + 0) * 64), v1 = ldv(vp + (k + 1) * 64), v2 = ldv(vp + (k + 2) * 64), v3 = ldv(vp + (k + 3) * 64);
                    const double x0 = xload(x, e, c0), x1 = xload(x, e, c1), x2 = xload(x, e, c2), x3 = xload(x, e, c3);
                    s0 = fma(v0, x0, s0);
                    s1 = fma(v1, x1, s1);
                    s0 = fma(v2, x2, s0);
                    s1 = fma(v3, x3, s1);
                }
                for (; k < w; ++k) s0 = fma(ldv(vp + k * 64), xload(x, e, ldc(cp + k * 64)), s0);
                if (prow[r] >= 0) res[prow[r] - win * W] = s0 + s1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = win * W + r * KK_TPB + tid;
            if (row < nrows) {
                double raw = res[r * KK_TPB + tid];
                if (e.acc == 1) y[row] = raw;
                else if (e.acc == 2) y[row] = yold[r] + raw;
                else {
                    if (e.acc == 3) raw += yold[r];
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
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) { off[r] = off2[r]; offn[r] = offn2[r]; prow[r] = prow2[r]; }
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

// SpMM on ELL: Y[:, j] = A X[:, j], j < nb <= NB (apply(f, ::Block), blocklanczos.jl:39): the matrix
// is streamed once for the whole block instead of once per vector.  RPL = rows per lane: 2 (16-byte matrix loads, 512 rows
// per block) or 1 (every gather instruction of a stencil row run covers 512 contiguous bytes and a block spans 256 rows,
// which halves the row window -- resident blocks x rows x nb columns -- the XCD's L2 has to hold for the +-nx neighbours).
template <int NB, int RPL>
__global__ __launch_bounds__(KK_TPB) void k_spmm_ell(const int32_t* __restrict__ ecol, const double* __restrict__ eval,
                                                     int64_t ell_ld, int width, int64_t nrows,
                                                     const double* __restrict__ X, int64_t ldx, double* __restrict__ Y,
                                                     int64_t ldy, int nb, int nb_logical, int64_t n_local,
                                                     const double* __restrict__ G, int64_t ldg, int64_t row_base, int64_t row_end) {
    // row-sharded operator: columns >= n_local read the ghost block G (column j of the block at G + j*ldg), filled by
    // ONE grouped exchange for all nb vectors before the launch (kk_halo_exchange_block)
    const int per = (nb_logical + 7) >> 3;
    const int nbx = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    for (int cblk = blockIdx.x >> 3; cblk < per; cblk += nbx) {
        const int lb = xcd * per + cblk;
        if (lb >= nb_logical) break;
        const int64_t row = row_base + ((int64_t)lb * KK_TPB + threadIdx.x) * RPL;   // rows [row_base, row_end), row_base even
        if (row >= row_end) continue;
        if (RPL == 2) {
            d2 acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = d2{0.0, 0.0};
            for (int k = 0; k < width; ++k) {
                const int2 cc = ldi2s(ecol + (int64_t)k * ell_ld + row);
                const d2 v = ld2s(eval + (int64_t)k * ell_ld + row);
                // branch-free gathers (address selected, load unconditional; columns j >= nb re-read column nb - 1 and their sums
                // are never stored): all 2 NB loads of a slot are in flight together -- see xload()
                const bool g0 = n_local >= 0 && cc.x >= n_local, g1 = n_local >= 0 && cc.y >= n_local;
                const double* b0 = g0 ? G + (cc.x - n_local) : X + cc.x;
                const double* b1 = g1 ? G + (cc.y - n_local) : X + cc.y;
                const int64_t st0 = g0 ? ldg : ldx, st1 = g1 ? ldg : ldx;
                double x0[NB], x1[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int jj = j < nb ? j : nb - 1;
                    x0[j] = b0[(int64_t)jj * st0];
                    x1[j] = b1[(int64_t)jj * st1];
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[j].x = fma(v.x, x0[j], acc[j].x);
                    acc[j].y = fma(v.y, x1[j], acc[j].y);
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
        } else {
            double acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = 0.0;
            for (int k = 0; k < width; ++k) {
                const int cc = __builtin_nontemporal_load(ecol + (int64_t)k * ell_ld + row);
                const double v = __builtin_nontemporal_load(eval + (int64_t)k * ell_ld + row);
                const bool g0 = n_local >= 0 && cc >= n_local;
                const double* b0 = g0 ? G + (cc - n_local) : X + cc;
                const int64_t st0 = g0 ? ldg : ldx;
                double x0[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) x0[j] = b0[(int64_t)(j < nb ? j : nb - 1) * st0];
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = fma(v, x0[j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (j < nb) Y[(int64_t)j * ldy + row] = acc[j];
        }
    }
}

// Multi-column apply of a grid stencil (kk_sparse_dev::dia_*): Y[:, j] = A X[:, j] by SWEEPING instead of gathering.
// A wave owns 62 consecutive positions of a grid line (lanes 1..62; lanes 0 and 63 carry the left / right neighbour so the
// +-1 entries are wave shifts) and walks down `lines` grid lines; per line it loads the next line of X once (coalesced,
// 8 bytes per lane) into a three-line register window (x[r-D], x[r], x[r+D] for all NB columns), so every element of X is
// read once per sweep (+ 2 halo lines per `lines`) -- the gather kernel re-fetches the +-D neighbours of 16 columns from
// HBM because the rows in flight per XCD exceed its L2 (5.9 GB fetched for 1.9 GB of algorithmic reads).
// Diagonals are dense arrays (no column indices): 5 or 9 coalesced 8-byte loads per row.
__device__ __forceinline__ double wave_from_left(double v) {    // lane l receives lane l-1 (lane 0: unchanged)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_from_right(double v) {   // lane l receives lane l+1 (lane 63: unchanged)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int NB, int PTS, bool CONST, bool NTY>
__global__ __launch_bounds__(KK_TPB) void k_spmm_dia(const double* __restrict__ dval, int64_t dld, int64_t D, int64_t nrows,
                                                     const double* __restrict__ X, int64_t ldx, double* __restrict__ Y,
                                                     int64_t ldy, int nb, int strips, int lines, int64_t Tlo, int64_t T,
                                                     int64_t row_lo, int64_t row_hi, dia_cst cst) {
    // grid lines [Tlo, T) are swept; results are stored for rows [row_lo, row_hi) only (the whole operator, or the
    // ghost-free interior of a row-sharded stencil -- its window loads stay inside the local vector)
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * (KK_TPB / 64) + (threadIdx.x >> 6);
    const int strip = (int)(wv % strips);
    const int64_t t0 = Tlo + (wv / strips) * lines;
    if (t0 >= T) return;
    const int64_t t1 = imin(t0 + lines, T);
    const int64_t i = (int64_t)strip * 62 + lane - 1;     // position inside the grid line (halo lanes: -1 / one past the strip)
    const bool own = lane >= 1 && lane <= 62 && i < D;
    int64_t r = t0 * D + i;                               // linear row of this lane on the current line
    const int64_t ixc = CONST ? (((i + cst.phase) % D) + D) % D : 0;   // CONST: true position inside the grid line (the block may start mid-line)
    double xm[NB], x0[NB], xp[NB];
    auto fetch = [&](double* dst, int64_t rr) {
        const bool ok = rr >= 0 && rr < nrows;
#pragma unroll
        for (int j = 0; j < NB; ++j) dst[j] = (ok && j < nb) ? X[(int64_t)j * ldx + rr] : 0.0;
    };
    fetch(xm, r - D);
    fetch(x0, r);
    for (int64_t t = t0; t < t1; ++t, r += D) {
        fetch(xp, r + D);
        const bool rok = r >= 0 && r < nrows;
        double d[PTS];
        if (CONST) {   // constant coefficients: c_q, except where a +-1 shift would leave the grid line (true position = ixc)
#pragma unroll
            for (int q = 0; q < PTS; ++q) {
                const int bq = PTS == 5 ? (q == 1 ? -1 : (q == 3 ? 1 : 0)) : q % 3 - 1;
                d[q] = (!rok || (bq < 0 && ixc == 0) || (bq > 0 && ixc == D - 1)) ? 0.0 : cst.c[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < PTS; ++q) d[q] = rok ? __builtin_nontemporal_load(dval + (int64_t)q * dld + r) : 0.0;
        }
        const bool st = own && rok && r >= row_lo && r < row_hi;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < nb) {
                double a;
                if (PTS == 5) {   // offsets -D, -1, 0, +1, +D
                    a = d[0] * xm[j];
                    a = fma(d[1], wave_from_left(x0[j]), a);
                    a = fma(d[2], x0[j], a);
                    a = fma(d[3], wave_from_right(x0[j]), a);
                    a = fma(d[4], xp[j], a);
                } else {          // three lines x {-1, 0, +1}
                    a = d[0] * wave_from_left(xm[j]);
                    a = fma(d[1], xm[j], a);
                    a = fma(d[2], wave_from_right(xm[j]), a);
                    a = fma(d[3], wave_from_left(x0[j]), a);
                    a = fma(d[4], x0[j], a);
                    a = fma(d[5], wave_from_right(x0[j]), a);
                    a = fma(d[6], wave_from_left(xp[j]), a);
                    a = fma(d[7], xp[j], a);
                    a = fma(d[8], wave_from_right(xp[j]), a);
                }
                // (non-temporal for long blocks: the Gram pass that follows reads A X non-temporally and runs 9 % faster -- block
                //  step -3 % in a same-box A/B; non-temporal LOADS of X make this kernel slower.  A template parameter: as a
                //  run-time branch hipcc merges the two stores into one plain store)
                if (st) { if (NTY) __builtin_nontemporal_store(a, Y + (int64_t)j * ldy + r); else Y[(int64_t)j * ldy + r] = a; }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) { xm[j] = x0[j]; x0[j] = xp[j]; }
    }
}

// The same sweep with ALIGNED 16-byte window loads and the next line in flight while the current one is multiplied (5-point stencil,
// constant coefficients, EVEN line length, lines starting at phase 0, vectors below 2 GB).  k_spmm_dia moves 8 bytes per lane and load
// (62 of 64 lanes, strips that start 8 bytes before a 16-byte boundary), waits for the line it has just requested, and carries 16
// columns x 3 lines in registers: 4.2 TB/s on the 10M-row block step where the read+write stream kernels reach 5.6-5.8.  Here
//   * a lane owns the positions 2l, 2l + 1 of its wave's 128-wide strip: ONE 16-byte load per column and line, one 16-byte store;
//   * the +-1 neighbours are the lanes next door (DPP wave shifts of the centre pair); only lane 0 / lane 63 fetch theirs -- one 8-byte
//     buffer load per column and line whose out-of-range offset switches the other 62 lanes off in the address unit (no branch);
//   * a wave carries NBW (2 / 4) of the block's columns (blockIdx.y = column group: the columns are independent and the value-free
//     stencil has no per-row data to share between them) through a FOUR-line window: line t + 2 is requested before line t is
//     multiplied, and the window rotates by renaming (the loop is unrolled four times) instead of by register moves.
// Same operands in the same order as k_spmm_dia: bit-identical results.
template <int NBW> struct dia_line { d2 v[NBW]; double e[NBW]; };
template <int CTRL>
__device__ __forceinline__ double dia_wave_shift_old(double old, double v) {   // 0x138: lane i takes lane i - 1, lane 0 takes `old`; 0x130: lane i takes lane i + 1, lane 63 `old`
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int NBW, bool NTY>
__global__ __launch_bounds__(KK_TPB, (NBW <= 2 ? 6 : 3)) void k_spmm_dia_al(int64_t D, int64_t nrows, const double* __restrict__ X, int64_t ldx, double* __restrict__ Y,
                                                        int64_t ldy, int nb, int strips, int lines, int64_t Tlo, int64_t T, int64_t row_lo,
                                                        int64_t row_hi, dia_cst cst) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * (KK_TPB / 64) + (threadIdx.x >> 6);
    const int strip = (int)(wv % strips);
    const int64_t t0 = Tlo + (wv / strips) * lines;
    if (t0 >= T) return;
    const int64_t t1 = imin(t0 + lines, T);
    const int j0 = (int)blockIdx.y * NBW;
    const int64_t p0 = (int64_t)strip * 128, p = p0 + 2 * lane;          // position of the lane's pair inside the grid line
    const bool own = p < D;                                              // (D even: a pair is inside the line or outside)
    const int64_t pe = lane == 0 ? p0 - 1 : p0 + 128;                    // lane 0: left neighbour of the strip, lane 63: right neighbour
    const bool eown = (lane == 0 && strip > 0) || (lane == 63 && pe < D);
    __amdgpu_buffer_rsrc_t rx[NBW], ry[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {   // (the launcher sends whole groups of NBW columns here)
        rx[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (int64_t)(j0 + j) * ldx), 0, (int)(nrows * 8), 0x00020000);
        ry[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + (int64_t)(j0 + j) * ldy), 0, (int)(nrows * 8), 0x00020000);
    }
    const unsigned off_none = 0xfffffff0u;
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    // line t of the window: the pairs two lines ahead of the line being multiplied, the strip's edge elements (needed when the line is
    // the CENTRE of the window) one line ahead.  Lines outside the operator read as zero (a negative 32-bit offset wraps beyond the records)
    auto fetch_pairs = [&](dia_line<NBW>& L, int64_t t) {
        const unsigned vo = (t <= t1 && own) ? (unsigned)((t * D + p) * 8) : off_none;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const v4u_ q = __builtin_amdgcn_raw_buffer_load_b128(rx[j], vo, 0, 0);
            L.v[j] = d2{__hiloint2double((int)q.y, (int)q.x), __hiloint2double((int)q.w, (int)q.z)};
        }
    };
    auto fetch_edges = [&](dia_line<NBW>& L, int64_t t) {
        const unsigned eo = (t < t1 && eown) ? (unsigned)((t * D + pe) * 8) : off_none;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const dia_v2u q2 = __builtin_amdgcn_raw_buffer_load_b64(rx[j], eo, 0, 0);
            L.e[j] = __hiloint2double((int)q2.y, (int)q2.x);
        }
    };
    // coefficients of the lane's two positions: the -1 entry does not exist at position 0, the +1 entry not at position D - 1
    const double cS = cst.c[0], cW0 = p == 0 ? 0.0 : cst.c[1], cW1 = cst.c[1], cC = cst.c[2], cE0 = cst.c[3], cE1 = (p + 2 == D) ? 0.0 : cst.c[3], cN = cst.c[4];
    auto line = [&](const dia_line<NBW>& Lm, const dia_line<NBW>& L0, const dia_line<NBW>& Lp, int64_t t) {
        const int64_t r = t * D + p;
        // (row_lo, row_hi even: both rows of the pair or none; a store that is not due is switched off by its offset, like the loads)
        const unsigned so = (own && r >= row_lo && r < row_hi) ? (unsigned)(r * 8) : off_none;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const double left = dia_wave_shift_old<0x138>(L0.e[j], L0.v[j].y), right = dia_wave_shift_old<0x130>(L0.e[j], L0.v[j].x);
            double ax = cS * Lm.v[j].x;
            ax = fma(cW0, left, ax);
            ax = fma(cC, L0.v[j].x, ax);
            ax = fma(cE0, L0.v[j].y, ax);
            ax = fma(cN, Lp.v[j].x, ax);
            double ay = cS * Lm.v[j].y;
            ay = fma(cW1, L0.v[j].x, ay);
            ay = fma(cC, L0.v[j].y, ay);
            ay = fma(cE1, right, ay);
            ay = fma(cN, Lp.v[j].y, ay);
            v4u_ q;
            q.x = (unsigned)__double2loint(ax); q.y = (unsigned)__double2hiint(ax); q.z = (unsigned)__double2loint(ay); q.w = (unsigned)__double2hiint(ay);
            // (non-temporal for long blocks: see k_spmm_dia)
            if (NTY) __builtin_amdgcn_raw_buffer_store_b128(q, ry[j], so, 0, 2); else __builtin_amdgcn_raw_buffer_store_b128(q, ry[j], so, 0, 0);
        }
    };
    dia_line<NBW> A, B, C, E;
    fetch_pairs(A, t0 - 1); fetch_pairs(B, t0); fetch_edges(B, t0); fetch_pairs(C, t0 + 1);
    for (int64_t t = t0;;) {
        fetch_pairs(E, t + 2); fetch_edges(C, t + 1); line(A, B, C, t); if (++t >= t1) break;
        fetch_pairs(A, t + 2); fetch_edges(E, t + 1); line(B, C, E, t); if (++t >= t1) break;
        fetch_pairs(B, t + 2); fetch_edges(A, t + 1); line(C, E, A, t); if (++t >= t1) break;
        fetch_pairs(C, t + 2); fetch_edges(B, t + 1); line(E, A, B, t); if (++t >= t1) break;
    }
}

// The SINGLE-VECTOR apply of a value-free 5-point stencil as a sweep (reference: src/apply.jl:1, the call sites
// factorizations/lanczos.jl:306-310 and arnoldi.jl:242).  k_spmv_dia<5, U, true, true> covers 512-row chunks and fetches the far neighbours
// of every chunk as loads of their own: three 16-byte x loads per row pair, two of them for lines some other block has requested a moment
// before -- cheap in HBM bytes, not in L2 requests (0.67 of peak at 10 M rows, 0.39 in the launch-bound 2 M-row applies).  Here, as in
// k_spmm_dia_al, a wave walks `lines` grid lines of its strip through a FOUR-line window that rotates by renaming:
//   * a lane owns positions 2l, 2l + 1 of a 128-wide strip: per line ONE 16-byte load of x (line t + 2 requested before line t is
//     multiplied), one of v_prev (requested with it: two lines ahead), one 16-byte store of y;
//   * the +-1 neighbours come from the lanes next door (DPP wave shifts); lanes 0 / 63 fetch the strip's edge elements through a
//     descriptor whose out-of-range offset switches the other 62 lanes off;
//   * a wave carries NS strips side by side (independent streams: more loads in flight per wave), a block 4 NS of them;
//   * the fused epilogues of k_spmv_dia -- scale by the device scalar, a0 x, -beta v_prev, the alpha dot in CGS or MGS order, |y|^2,
//     non-temporal store -- run on the line in registers.
// Blocks are dealt to the XCDs round-robin, so the (strip chunk, line group) grid is laid out in VIRTUAL COLUMNS: NV = nbl x bands of them
// (nbl chunks per line, the lines split into `bands` so that NV is a multiple of 8); block lb works on column lb % NV -- always the same
// XCD -- and steps DOWN the lines from launch row to launch row: the line group below, whose first halo line is this group's last line,
// belongs to the block NV places later on the same XCD (resident at the same time: the halo is an L2 hit), while the nbl blocks of one
// line group read a contiguous stretch of HBM.
// y: same operands in the same order as k_spmv_dia -- bit-identical; the inner products are summed in a different order (within rounding).
template <int NS> struct sw_line { d2 v[NS]; double e[NS]; d2 pv[NS]; };
template <int NS, bool NTY, bool VPREV>
__global__ __launch_bounds__(KK_TPB) void k_spmv_dia_sw(int64_t D, int64_t nrows, const double* __restrict__ x, double* __restrict__ y, spmv_epi e, dia_cst cst,
                                                        int nbl, int NV, int gpb, int lines, int64_t Tlo, int64_t T, int64_t row_lo, int64_t row_hi,
                                                        double* __restrict__ part_dot, double* __restrict__ part_nrm) {
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vc = (int)(blockIdx.x % (unsigned)NV);
    const int band = vc / nbl, sc = vc % nbl;
    const int crows = (int)(gridDim.x / (unsigned)NV);
    double dacc = 0, nacc = 0;
    const double xs = e.xs_dev ? *e.xs_dev : 1.0;
    const double bp = VPREV ? (e.bprev_dev ? *e.bprev_dev : e.bprev) : 0.0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(nrows * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(nrows * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(VPREV ? e.vprev : x), 0, (int)(nrows * 8), 0x00020000);
    const unsigned off_none = 0xfffffff0u;
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    int64_t p[NS], pe[NS];
    bool own[NS], eown[NS];
    double cW0[NS], cE1[NS];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int64_t strip = ((int64_t)sc * (KK_TPB / 64) + wave) * NS + j;
        const int64_t p0 = strip * 128;
        p[j] = p0 + 2 * lane;
        own[j] = p[j] < D;                                   // (D even: a pair is inside the line or outside)
        pe[j] = lane == 0 ? p0 - 1 : p0 + 128;               // lane 0: left neighbour of the strip, lane 63: right neighbour
        eown[j] = (lane == 0 && strip > 0 && p0 < D) || (lane == 63 && pe[j] < D);
        cW0[j] = p[j] == 0 ? 0.0 : cst.c[1];                 // the -1 entry does not exist at position 0 ...
        cE1[j] = p[j] + 2 == D ? 0.0 : cst.c[3];             // ... the +1 entry not at position D - 1
        any = any || p0 < D;
    }
    const double cS = cst.c[0], cW1 = cst.c[1], cC = cst.c[2], cE0 = cst.c[3], cN = cst.c[4];
    if (any) {   // (wave-uniform: a wave whose strips all lie beyond the line only joins the reductions below)
        for (int c = (int)(blockIdx.x / (unsigned)NV); c < gpb; c += crows) {
            const int64_t t0 = Tlo + ((int64_t)band * gpb + c) * lines;
            if (t0 >= T) break;
            const int64_t t1 = imin(t0 + lines, T);
            // line t of the window: the x pairs (and the v_prev pairs of the lines that are multiplied) two lines ahead, the strip's edge
            // elements one line ahead.  Lines outside the operator read as zero (a negative 32-bit offset wraps beyond the records)
            auto fetch_pairs = [&](sw_line<NS>& L, int64_t t) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const unsigned vo = (t <= t1 && own[j]) ? (unsigned)((t * D + p[j]) * 8) : off_none;
                    const v4u_ q = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
                    L.v[j] = d2{__hiloint2double((int)q.y, (int)q.x), __hiloint2double((int)q.w, (int)q.z)};
                    if (VPREV) {
                        const unsigned po = (t >= t0 && t < t1 && own[j]) ? (unsigned)((t * D + p[j]) * 8) : off_none;
                        const v4u_ q2 = __builtin_amdgcn_raw_buffer_load_b128(rp, po, 0, 0);
                        L.pv[j] = d2{__hiloint2double((int)q2.y, (int)q2.x), __hiloint2double((int)q2.w, (int)q2.z)};
                    }
                }
            };
            auto fetch_edges = [&](sw_line<NS>& L, int64_t t) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const unsigned eo = (t < t1 && eown[j]) ? (unsigned)((t * D + pe[j]) * 8) : off_none;
                    const dia_v2u q2 = __builtin_amdgcn_raw_buffer_load_b64(rx, eo, 0, 0);
                    L.e[j] = __hiloint2double((int)q2.y, (int)q2.x);
                }
            };
            auto line = [&](const sw_line<NS>& Lm, const sw_line<NS>& L0, const sw_line<NS>& Lp, int64_t t) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const int64_t r = t * D + p[j];
                    // (row_lo, row_hi even: both rows of the pair or none; a store that is not due is switched off by its offset, like the loads)
                    const bool due = own[j] && r >= row_lo && r < row_hi;
                    const unsigned so = due ? (unsigned)(r * 8) : off_none;
                    const double left = dia_wave_shift_old<0x138>(L0.e[j], L0.v[j].y), right = dia_wave_shift_old<0x130>(L0.e[j], L0.v[j].x);
                    double s0 = cS * Lm.v[j].x;
                    s0 = fma(cW0[j], left, s0);
                    s0 = fma(cC, L0.v[j].x, s0);
                    s0 = fma(cE0, L0.v[j].y, s0);
                    s0 = fma(cN, Lp.v[j].x, s0);
                    double s1 = cS * Lm.v[j].y;
                    s1 = fma(cW1, L0.v[j].x, s1);
                    s1 = fma(cC, L0.v[j].y, s1);
                    s1 = fma(cE1[j], right, s1);
                    s1 = fma(cN, Lp.v[j].y, s1);
                    // the epilogue of k_spmv_dia, operation by operation
                    const double u0 = s0 * xs, u1 = s1 * xs;
                    d2 out{e.a1 * u0, e.a1 * u1};
                    const d2 xc{L0.v[j].x * xs, L0.v[j].y * xs};
                    if (e.a0 != 0.0) { out.x = fma(e.a0, xc.x, out.x); out.y = fma(e.a0, xc.y, out.y); }
                    if (e.dot_mode == 1 && due) { dacc = fma(xc.x, out.x, dacc); dacc = fma(xc.y, out.y, dacc); }
                    if (VPREV) { out.x = fma(-bp, L0.pv[j].x, out.x); out.y = fma(-bp, L0.pv[j].y, out.y); }
                    if (r + 1 >= nrows) out.y = 0.0;   // odd nrows: keep the pad row zero
                    if (e.dot_mode == 2 && due) { dacc = fma(xc.x, out.x, dacc); dacc = fma(xc.y, out.y, dacc); }
                    if (e.want_nrm && due) { nacc = fma(out.x, out.x, nacc); nacc = fma(out.y, out.y, nacc); }
                    v4u_ q;
                    q.x = (unsigned)__double2loint(out.x); q.y = (unsigned)__double2hiint(out.x); q.z = (unsigned)__double2loint(out.y); q.w = (unsigned)__double2hiint(out.y);
                    if (NTY) __builtin_amdgcn_raw_buffer_store_b128(q, ry, so, 0, 2); else __builtin_amdgcn_raw_buffer_store_b128(q, ry, so, 0, 0);
                }
            };
            sw_line<NS> A, B, C, E;
            fetch_pairs(A, t0 - 1); fetch_pairs(B, t0); fetch_edges(B, t0); fetch_pairs(C, t0 + 1);
            for (int64_t t = t0;;) {
                fetch_pairs(E, t + 2); fetch_edges(C, t + 1); line(A, B, C, t); if (++t >= t1) break;
                fetch_pairs(A, t + 2); fetch_edges(E, t + 1); line(B, C, E, t); if (++t >= t1) break;
                fetch_pairs(B, t + 2); fetch_edges(A, t + 1); line(C, E, A, t); if (++t >= t1) break;
                fetch_pairs(C, t + 2); fetch_edges(B, t + 1); line(E, A, B, t); if (++t >= t1) break;
            }
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

// ranged launches: rows [r0, r1) of an ELL / diagonal operator; partial sums go to pd / pn + *nblk_io, *nblk_io advances
static void launch_spmv_ell_rows(kk_ctx ctx, const kk_sparse_dev& M, const double* x, double* y, const spmv_epi& e, int64_t r0,
                                 int64_t r1, double* pd, double* pn, int* nblk_io, int max_blocks) {
    if (r1 <= r0) return;
    const int nb_logical = (int)((r1 - r0 + 2 * KK_TPB - 1) / (2 * KK_TPB));
    const int per = (nb_logical + 7) / 8;
    const int nbx = std::max(1, std::min(per, max_blocks / 8));
    const int nblk = nbx * 8;
    hipLaunchKernelGGL(k_spmv_ell, dim3(nblk), dim3(KK_TPB), 0, ctx->stream, M.ell_col, M.ell_val, M.ell_ld, M.width, M.nrows, x, y,
                       e, nb_logical, pd + *nblk_io, pn + *nblk_io, r0, r1);
    *nblk_io += nblk;
}
static void launch_spmv_dia_rows(kk_ctx ctx, const kk_sparse_dev& M, const double* x, double* y, const spmv_epi& e, int64_t r0,
                                 int64_t r1, double* pd, double* pn, int* nblk_io, int max_blocks) {
    if (r1 <= r0) return;
    const bool cc = M.dia_const && ctx->spmv_dia_const;
    // row pairs per lane: 1 / 2 (/ 4 for the value-free form, whose lanes carry no diagonal values)
    // 0 = by size (tools/stencil_shape_sweep.py): one pair while the vectors fit the Infinity Cache and the launch is short (more
    // waves, shorter tail), the deepest form from 3e7 rows on, where the bytes in flight per CU are what sets the rate
    const int want = ctx->spmv_dia_pairs > 0 ? ctx->spmv_dia_pairs : (r1 - r0 >= 30000000 ? 4 : 1);
    const int U = want >= 4 ? (cc ? 4 : 2) : (want == 2 ? 2 : 1);
    const int nb_logical = (int)((r1 - r0 + 2 * U * KK_TPB - 1) / (2 * U * KK_TPB));
    const int per = (nb_logical + 7) / 8;
    const int nbx = std::max(1, std::min(per, max_blocks / 8));
    const int nblk = nbx * 8;
    dia_offs of;
    const int64_t D = M.dia_D;
    if (M.dia_pts == 5) { const int64_t o5[5] = {-D, -1, 0, 1, D}; for (int q = 0; q < 5; ++q) of.o[q] = o5[q]; for (int q = 5; q < 9; ++q) of.o[q] = 0; }
    else { const int64_t o9[9] = {-D - 1, -D, -D + 1, -1, 0, 1, D - 1, D, D + 1}; for (int q = 0; q < 9; ++q) of.o[q] = o9[q]; }
    dia_cst cst;
    for (int q = 0; q < 9; ++q) cst.c[q] = M.dia_c[q];
    cst.phase = M.dia_phase; cst.D = D;
#define SPMV_DIA_ARGS dim3(nblk), dim3(KK_TPB), 0, ctx->stream, M.dia_val, M.dia_ld, of, M.nrows, x, y, e, nb_logical, pd + *nblk_io, pn + *nblk_io, r0, r1, cst
    // 16-byte far loads + lane-shift neighbours (see k_spmv_dia): 5-point stencil, even grid-line length, byte offsets within 32 bits
    const bool al = ctx->spmv_dia_aligned && M.dia_pts == 5 && (D & 1) == 0 && M.nrows * 8 < ((int64_t)1 << 31) && (r0 & 1) == 0;
    // the sweeping form (k_spmv_dia_sw): value-free 5-point stencil whose lines start at phase 0, 16-byte aligned vectors, an even row range
    const int ns = ctx->spmv_dia_sw;
    const bool aligned16 = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)(e.vprev ? e.vprev : x)) & 15) == 0;
    // (a line so long that one launch row of blocks would not fit the partial-sum rows stays with k_spmv_dia: 4096 blocks x 512 positions)
    if ((ns == 1 || ns == 2) && al && cc && M.dia_phase == 0 && e.dot_mode != 3 && e.acc == 0 && aligned16 && (r1 & 1) == 0 && M.nrows * 8 < (int64_t)2000000000 &&
        (D + (KK_TPB / 64) * ns * 128 - 1) / ((KK_TPB / 64) * ns * 128) * 8 <= max_blocks) {
        const int64_t Tlo = r0 / D, T = (r1 + D - 1) / D;
        // lines per sweep: 0 = the measured best
        // (tools/spmv_dia_sw_sweep.py, profiles/r06_spmv_dia_sw_sweep.jsonl: 2 .. 4 lines are the fastest at 2 M and at 10 M rows -- short sweeps keep
        //  many waves in flight and the halo lines are L2 hits; 16 and more lose: fewer, longer waves)
        const int lines = ctx->spmv_dia_sw_lines > 0 ? ctx->spmv_dia_sw_lines : 4;
        const int wpb = (KK_TPB / 64) * ns * 128;                       // positions of a line one block covers
        const int nbl = (int)((D + wpb - 1) / wpb);                     // blocks per line group
        int g8 = nbl, b8 = 8;
        while (b8) { const int t_ = g8 % b8; g8 = b8; b8 = t_; }       // gcd(nbl, 8)
        int bands = 8 / g8;
        const int64_t ngroups = (T - Tlo + lines - 1) / lines;
        while (bands > 1 && ngroups < 4 * bands) bands >>= 1;           // (too few line groups to split: some XCDs idle rather than empty bands)
        const int NV = nbl * bands;
        const int gpb = (int)((ngroups + bands - 1) / bands);
        const int crows = std::max(1, std::min(gpb, max_blocks / NV));
        const int nblk_sw = crows * NV;
        ++ctx->spmv_dia_sw_launches;
#define SW_ARGS dim3(nblk_sw), dim3(KK_TPB), 0, ctx->stream, D, M.nrows, x, y, e, cst, nbl, NV, gpb, lines, Tlo, T, r0, r1, pd + *nblk_io, pn + *nblk_io
#define SW_CASE(NSV) do { if (e.nt_store) { if (e.vprev) hipLaunchKernelGGL((k_spmv_dia_sw<NSV, true, true>), SW_ARGS); else hipLaunchKernelGGL((k_spmv_dia_sw<NSV, true, false>), SW_ARGS); } \
                          else { if (e.vprev) hipLaunchKernelGGL((k_spmv_dia_sw<NSV, false, true>), SW_ARGS); else hipLaunchKernelGGL((k_spmv_dia_sw<NSV, false, false>), SW_ARGS); } } while (0)
        if (ns == 2) SW_CASE(2); else SW_CASE(1);
#undef SW_CASE
#undef SW_ARGS
        *nblk_io += nblk_sw;
        return;
    }
    if (al) {
        if (cc) {
            if (U == 4) hipLaunchKernelGGL((k_spmv_dia<5, 4, true, true>), SPMV_DIA_ARGS);
            else if (U == 2) hipLaunchKernelGGL((k_spmv_dia<5, 2, true, true>), SPMV_DIA_ARGS);
            else hipLaunchKernelGGL((k_spmv_dia<5, 1, true, true>), SPMV_DIA_ARGS);
        } else { if (U == 2) hipLaunchKernelGGL((k_spmv_dia<5, 2, false, true>), SPMV_DIA_ARGS); else hipLaunchKernelGGL((k_spmv_dia<5, 1, false, true>), SPMV_DIA_ARGS); }
    } else if (M.dia_pts == 5) {
        if (cc) {
            if (U == 4) hipLaunchKernelGGL((k_spmv_dia<5, 4, true>), SPMV_DIA_ARGS);
            else if (U == 2) hipLaunchKernelGGL((k_spmv_dia<5, 2, true>), SPMV_DIA_ARGS);
            else hipLaunchKernelGGL((k_spmv_dia<5, 1, true>), SPMV_DIA_ARGS);
        } else { if (U == 2) hipLaunchKernelGGL((k_spmv_dia<5, 2, false>), SPMV_DIA_ARGS); else hipLaunchKernelGGL((k_spmv_dia<5, 1, false>), SPMV_DIA_ARGS); }
    } else {
        if (cc) {
            if (U == 4) hipLaunchKernelGGL((k_spmv_dia<9, 4, true>), SPMV_DIA_ARGS);
            else if (U == 2) hipLaunchKernelGGL((k_spmv_dia<9, 2, true>), SPMV_DIA_ARGS);
            else hipLaunchKernelGGL((k_spmv_dia<9, 1, true>), SPMV_DIA_ARGS);
        } else { if (U == 2) hipLaunchKernelGGL((k_spmv_dia<9, 2, false>), SPMV_DIA_ARGS); else hipLaunchKernelGGL((k_spmv_dia<9, 1, false>), SPMV_DIA_ARGS); }
    }
#undef SPMV_DIA_ARGS
    *nblk_io += nblk;
}

// ---- launchers
int kk_launch_spmv(kk_ctx ctx, const kk_sparse_dev& M, const double* x, double* y, int64_t ld_y_rows,
                   const kk_spmv_fuse& f) {
    (void)ld_y_rows;
    if (M.plan) KK_TRY(kk_halo_exchange(ctx, M, x));   // row-sharded operator with a native plan: grouped ncclSend / ncclRecv
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
    e.nt_store = (M.nrows >= ctx->nt_store_rows) ? 1 : 0;
    double* pd = part_row(ctx, PART_SCAL_A);
    double* pn = part_row(ctx, PART_SCAL_B);
    int nblk = 0;
    // diagonal kernel: the whole operator when it has no ghost columns, the ghost-free interior rows [int_lo, int_hi) of a
    // row-sharded stencil (the boundary strips go through the gather kernel, which reads the ghost buffer)
    const bool dia_ok = M.format == 0 && M.dia_D > 0 && ctx->spmv_dia;
    const bool use_dia = dia_ok && M.n_ghost == 0;
    const bool use_split = dia_ok && M.n_ghost > 0 && M.int_hi > M.int_lo;
    std::unique_ptr<kk_prof_scope> ps(new kk_prof_scope(ctx, (use_dia || use_split) ? "k_spmv_dia" : (M.format == 0 ? "k_spmv_ell" : (M.format >= 2 ? "k_spmv_sell" : "k_spmv_csr"))));
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
            const int R = (int)(S.sell_sigma / KK_TPB);   // 256-row rounds per sorting window (build_tiled)
            nblk = (int)std::min<int64_t>((S.sell_nchunks + 4 * R - 1) / (4 * R), (int64_t)ctx->num_cus * 16);
            if (nblk < 1) nblk = 1;
#define SELLW_ARGS dim3(nblk), dim3(KK_TPB), 0, ctx->stream, S.sell_off, S.sell_perm, S.sell_col, S.sell_val, S.sell_nchunks, S.nrows, x, y, et, pd, pn
            if (R == 4) hipLaunchKernelGGL((k_spmv_sellw<4>), SELLW_ARGS);
            else if (R == 2) hipLaunchKernelGGL((k_spmv_sellw<2>), SELLW_ARGS);
            else if (R == 8) hipLaunchKernelGGL((k_spmv_sellw<8>), SELLW_ARGS);
            else hipLaunchKernelGGL((k_spmv_sellw<1>), SELLW_ARGS);
#undef SELLW_ARGS
        }
    } else if (use_dia) {
        launch_spmv_dia_rows(ctx, M, x, y, e, 0, M.nrows, pd, pn, &nblk, KK_MAX_BLOCKS);
    } else if (use_split) {
        launch_spmv_dia_rows(ctx, M, x, y, e, M.int_lo, M.int_hi, pd, pn, &nblk, KK_MAX_BLOCKS - 512);
        launch_spmv_ell_rows(ctx, M, x, y, e, 0, M.int_lo, pd, pn, &nblk, 256);
        launch_spmv_ell_rows(ctx, M, x, y, e, M.int_hi, M.nrows, pd, pn, &nblk, 256);
    } else if (M.format == 0) {
        launch_spmv_ell_rows(ctx, M, x, y, e, 0, M.nrows, pd, pn, &nblk, KK_MAX_BLOCKS);
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

// Y[:, j] = A X[:, j], j < nb (any nb: processed 16 / 8 / 4 columns at a time)
int kk_launch_spmm(kk_ctx ctx, const kk_sparse_dev& M, const double* X, int64_t ldx, double* Y, int64_t ldy, int nb) {
    // one launch for the block: ELL operators without ghosts, or with a native exchange plan (one grouped exchange for all
    // nb vectors); CSR / SELL formats and hook-ghosted operators go column by column
    // A native exchange plan makes the apply a COLLECTIVE step: every rank must issue the same sequence of grouped
    // exchanges, whatever storage format its own row block happened to get (ELL here, SELL on the neighbour) and whether or
    // not it has ghost columns of its own (a rank without any still serves its peers).  So: with a plan, always one grouped
    // exchange per <= 16 columns; the local format only decides which kernels read the received block.
    const bool planned = M.plan != nullptr;
    if (planned && nb > 16) {
        for (int j0 = 0; j0 < nb; j0 += 16)
            KK_TRY(kk_launch_spmm(ctx, M, X + (int64_t)j0 * ldx, ldx, Y + (int64_t)j0 * ldy, ldy, std::min(16, nb - j0)));
        return KK_OK;
    }
    const bool ghost_block = planned && M.n_ghost > 0;
    const bool one_launch = M.format == 0 && !M.halo && (M.n_ghost == 0 || M.plan);
    const double* G = nullptr;
    int64_t ldg = 0, nloc = -1;
    if (planned) {
        KK_TRY(kk_halo_exchange_block(ctx, M, X, ldx, nb, &G, &ldg));
        if (ghost_block) nloc = M.n_local;
    }
    if (!one_launch) {   // CSR / SELL formats and hook-ghosted operators: column by column
        for (int j = 0; j < nb; ++j) {
            kk_spmv_fuse f;
            if (planned) {   // the ghosts of column j are already here: same matrix, ghost buffer = column j of the block, no exchange
                kk_sparse_dev Mj = M;
                Mj.plan = nullptr;
                Mj.ghost = const_cast<double*>(G) + (int64_t)j * ldg;
                KK_TRY(kk_launch_spmv(ctx, Mj, X + (int64_t)j * ldx, Y + (int64_t)j * ldy, ldy, f));
            } else {
                KK_TRY(kk_launch_spmv(ctx, M, X + (int64_t)j * ldx, Y + (int64_t)j * ldy, ldy, f));
            }
        }
        return KK_OK;
    }
    // grid stencil: sweep the lines with a register window (every element of X read once) -- the whole operator, or the
    // ghost-free interior rows of a row-sharded one
    dia_cst cst;
    for (int q = 0; q < 9; ++q) cst.c[q] = M.dia_c[q];
    cst.phase = M.dia_phase; cst.D = M.dia_D;
    const bool cc = M.dia_const && ctx->spmv_dia_const;
    auto launch_dia = [&](int64_t row_lo, int64_t row_hi) {
        if (row_hi <= row_lo) return;
        const int strips = (int)((M.dia_D + 61) / 62);
        const int64_t Tlo = row_lo / M.dia_D, T = (row_hi + M.dia_D - 1) / M.dia_D;
        const int lines = ctx->spmm_dia_lines;
        const int64_t waves = (int64_t)strips * ((T - Tlo + lines - 1) / lines);
        dim3 g((unsigned)((waves + KK_TPB / 64 - 1) / (KK_TPB / 64))), b(KK_TPB);
        // aligned form (k_spmm_dia_al): one launch for all columns, NBW of them per wave
        const int nbw = ctx->spmm_dia_al;
        int j_first = 0;
        if ((nbw == 2 || nbw == 4) && cc && M.dia_pts == 5 && M.dia_D % 2 == 0 && M.dia_phase == 0 && M.nrows * 8 < (int64_t)2000000000 && ldx % 2 == 0 &&
            ldy % 2 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0 && row_lo % 2 == 0 && row_hi % 2 == 0) {
            const int strips2 = (int)((M.dia_D + 127) / 128);
            const int lines2 = ctx->spmm_dia_al_lines;
            const int64_t waves2 = (int64_t)strips2 * ((T - Tlo + lines2 - 1) / lines2);
            const int nfull = nb / nbw * nbw;   // whole groups of nbw columns; the others go through the 8-byte form below
            dim3 g2((unsigned)((waves2 + KK_TPB / 64 - 1) / (KK_TPB / 64)), (unsigned)(nfull / nbw)), b2(KK_TPB);
            kk_prof_scope ps(ctx, "k_spmm_dia");
#define DIA_AL(NBWT, NTY) hipLaunchKernelGGL((k_spmm_dia_al<NBWT, NTY>), g2, b2, 0, ctx->stream, M.dia_D, M.nrows, X, ldx, Y, ldy, nfull, strips2, lines2, Tlo, T, row_lo, row_hi, cst)
            const bool nty = M.nrows >= ctx->nt_store_rows;
            if (nfull > 0) {
                ++ctx->spmm_dia_al_launches;
                if (nbw == 4) { if (nty) DIA_AL(4, true); else DIA_AL(4, false); }
                else { if (nty) DIA_AL(2, true); else DIA_AL(2, false); }
            }
#undef DIA_AL
            j_first = nfull;
        }
        int j0 = j_first;
        while (j0 < nb) {
            const int rem = nb - j0;
            const int n = std::min(rem > 8 ? std::min(rem, 16) : rem, ctx->spmm_cols);
            const double* x = X + (int64_t)j0 * ldx;
            double* y = Y + (int64_t)j0 * ldy;
            kk_prof_scope ps(ctx, "k_spmm_dia");
#define DIA_ARGS M.dia_val, M.dia_ld, M.dia_D, M.nrows, x, ldx, y, ldy, n, strips, lines, Tlo, T, row_lo, row_hi, cst
#define DIA_CASE2(NBT, NTY) \
            if (cc) { if (M.dia_pts == 5) hipLaunchKernelGGL((k_spmm_dia<NBT, 5, true, NTY>), g, b, 0, ctx->stream, DIA_ARGS); \
                      else hipLaunchKernelGGL((k_spmm_dia<NBT, 9, true, NTY>), g, b, 0, ctx->stream, DIA_ARGS); } \
            else { if (M.dia_pts == 5) hipLaunchKernelGGL((k_spmm_dia<NBT, 5, false, NTY>), g, b, 0, ctx->stream, DIA_ARGS); \
                   else hipLaunchKernelGGL((k_spmm_dia<NBT, 9, false, NTY>), g, b, 0, ctx->stream, DIA_ARGS); }
#define DIA_CASE(NBT) if (M.nrows >= ctx->nt_store_rows) { DIA_CASE2(NBT, true) } else { DIA_CASE2(NBT, false) }
            if (n > 8) { DIA_CASE(16) }
            else if (n > 4) { DIA_CASE(8) }
            else { DIA_CASE(4) }
#undef DIA_CASE
#undef DIA_CASE2
#undef DIA_ARGS
            j0 += n;
        }
    };
    // gather kernel on the rows [r0, r1)
    auto launch_ell = [&](int64_t r0, int64_t r1) {
        if (r1 <= r0) return;
        const int rpl = ctx->spmm_rpl == 1 ? 1 : 2;
        const int nb_logical = (int)((r1 - r0 + rpl * KK_TPB - 1) / (rpl * KK_TPB));
        const int per = (nb_logical + 7) / 8;
        // Each XCD sweeps a contiguous band of rows; the rows its resident blocks work on at one time are the window whose
        // gathered entries must stay in that XCD's 4 MB L2 for the +-nx neighbours of a stencil to be L2 hits.  With nb
        // right-hand sides the window holds nb columns: spmm_bpc resident blocks per CU x 32 CUs x 256*rpl rows x nb x 8 bytes
        // (2 blocks per CU, 2 rows per lane, nb = 16: 4 MB), so the grid is capped instead of filling every slot as the
        // 1-column SpMV does.
        int nbx = std::min(per, KK_MAX_BLOCKS / 8);
        if (ctx->spmm_bpc > 0) nbx = std::min(nbx, std::max(1, ctx->num_cus / 8) * ctx->spmm_bpc);
        if (nbx < 1) nbx = 1;
        dim3 g(nbx * 8), b(KK_TPB);
        int j0 = 0;
        while (j0 < nb) {
            const int rem = nb - j0;
            const double* x = X + (int64_t)j0 * ldx;
            double* y = Y + (int64_t)j0 * ldy;
            const double* gj = G ? G + (int64_t)j0 * ldg : nullptr;
            kk_prof_scope ps(ctx, "k_spmm_ell");
            const int n = std::min(rem > 8 ? std::min(rem, 16) : rem, ctx->spmm_cols);   // spmm_cols < 16: narrower row window per XCD
#define SPMM_ARGS M.ell_col, M.ell_val, M.ell_ld, M.width, M.nrows, x, ldx, y, ldy, n, nb_logical, nloc, gj, ldg, r0, r1
#define SPMM_CASE(NBT) \
            if (rpl == 1) hipLaunchKernelGGL((k_spmm_ell<NBT, 1>), g, b, 0, ctx->stream, SPMM_ARGS); \
            else hipLaunchKernelGGL((k_spmm_ell<NBT, 2>), g, b, 0, ctx->stream, SPMM_ARGS);
            if (n > 8) { SPMM_CASE(16) }
            else if (n > 4) { SPMM_CASE(8) }
            else { SPMM_CASE(4) }
#undef SPMM_CASE
#undef SPMM_ARGS
            j0 += n;
        }
    };
    const bool dia_ok = M.dia_D > 0 && ctx->spmm_dia && nb >= 2;
    if (dia_ok && M.n_ghost == 0) {
        launch_dia(0, M.nrows);
    } else if (dia_ok && ghost_block && M.int_hi > M.int_lo) {
        launch_dia(M.int_lo, M.int_hi);
        launch_ell(0, M.int_lo);
        launch_ell(M.int_hi, M.nrows);
    } else {
        launch_ell(0, M.nrows);
    }
    KK_HIP(hipGetLastError());
    return KK_OK;
}

