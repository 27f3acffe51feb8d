// Persistent cooperative kernel for the STRICT modified Gram-Schmidt sweep (SURVEY.md 8(a) row a6, section 7 hard part (b)):
// the reference's sequential order  for q in V: s = <q, w>; w -= s q   (src/orthonormal.jl:414-439, used by
// lanczosrecurrence MGS2 at factorizations/lanczos.jl:325-338 and by arnoldirecurrence!! through orthogonalize!!)
// with the work vector w RESIDENT IN REGISTERS for the whole sweep.
//
//   * grid = one block per CU (co-resident: cooperative launch), PT threads each; thread t of block b owns the rows
//     ((i*G + b)*PT + t)*2 + {0,1}, i < NV, i.e. NV double2 = 2*NV doubles of w in VGPRs.  256 CUs x 1024 threads x 40
//     doubles = 10.48 M rows: the whole 10M-row work vector of BASELINE.json configs[1] lives on chip (80 MB of the
//     128 MB of vector registers).
//   * step j streams q_{j-1} and q_j once each:  w -= s_{j-1} q_{j-1}  (the pending axpy, as k_mgs_step fuses it) and the
//     partial <q_j, w>.  q_j was read one step earlier as the "next" vector with cache-allocating loads, so its second read
//     is served by the 256 MB Infinity Cache; HBM sees every basis vector ONCE per sweep (8 N bytes per vector instead of
//     the 32 N of the launch-per-vector kernel and the 16 N of the projection-based passes), w twice per sweep (16 N).
//   * the global inner product needs every block: one deterministic grid reduction per vector -- per-block partial ->
//     agent-scope release/acquire counter -> every block sums the G partials in the same fixed order (bitwise identical
//     s on all blocks, bitwise reproducible run to run).  Every spin is bounded by the wall clock: on a timeout the error
//     flag is raised, no block writes w back (HBM still holds the input) and the host reports the failure.
#include "kk_internal.h"
#include "kk_device.h"
#include "kk_xsync.h"
#include <algorithm>

// First read of a basis vector (as q_next): non-temporal for the grid-rows that will wait on chip (LDS / spare registers) and
// are therefore never read again, cache-allocating only for the rows that ARE read a second time one step later.  With every
// row allocated (round 3) a CU's 39 rows x 8 KB x 32 CUs = 10 MB per vector washed through each XCD's 4 MB L2 and the second
// read went out to the fabric; with only the 12 re-read rows allocated (3 MB per XCD) the second read is an L2 hit:
// 1084 -> 1244 it/s on the headline sweep (0: nothing non-temporal 1084, 2: everything 1087).
#ifndef KK_PERSIST_NT_FIRST
#define KK_PERSIST_NT_FIRST 1
#endif
// (every spin is bounded by the wall clock, 100 MHz ticks: the budget comes with the launch -- kk_persist_timeout_ticks)
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned long long gu64;

template <int NW>
__device__ __forceinline__ double block_sum_w(double v, double* sm /* >= NW doubles */) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[wave] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += sm[i];   // fixed order: identical bits in every thread
    __syncthreads();
    return t;
}

// Sum of `acc` over all threads of all blocks (every block is resident: one per CU).  Data-tagged hand-off
// (cdna_hip_programming.md Guideline 16, form R2 / "allgather"): a block publishes its partial as ONE 16-byte granule
// {epoch, high word, low word, epoch} on a 128-byte line of its own, written by a single write-through (sc1) dwordx4
// store -- the data is the flag, no counter, no fence; wave 0 of every block sweeps the G granules (one sc1 dwordx4 load
// each, all loads of a lane in flight before the first tag is looked at) until every tag pair equals the epoch and adds
// the partials in a fixed order, so all blocks obtain the same bits.  Each 8-byte half of a granule carries its own tag:
// even if the fabric split the 16 bytes into two naturally aligned halves, a half that is still old fails its tag.
// Measured against the round-3 form (two separate 8-byte granules per block, contiguous for all blocks):
// 3.06 -> 2.29 us per reduction on an idle chip (tools/grid_reduce_variants.hip, profiles/r03_grid_reduce_variants.jsonl).
// Two granule sets are used alternately (epoch parity): a fast block can publish step s+1 while a slow one still sweeps
// step s; it cannot reach step s+2 before the slow one has published s+1, i.e. has finished that sweep.  Epochs are
// unique over the life of the context (`ebase` advances from launch to launch), so the area is never cleared between
// launches.  Returns false on a timeout.
// first half: the block's partial, published (one write-through store by thread 0)
template <int PT>
__device__ __forceinline__ void grid_publish(double acc, int step, unsigned ebase, char* __restrict__ sync, double* sm, unsigned gstride) {
    const int G = gridDim.x;
    const double v = block_sum_w<PT / 64>(acc, sm);
    const unsigned epoch = ebase + (unsigned)step + 1u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, 2 * G * KK_SYNC_LINE, 0x00020000);
    const unsigned set_off = (unsigned)(step & 1) * (unsigned)G * KK_SYNC_LINE;   // (a set always spans G lines, whatever the stride)
    if (threadIdx.x == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        v4u t;
        t.x = epoch; t.y = (unsigned)(bits >> 32); t.z = (unsigned)bits; t.w = epoch;
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, set_off + blockIdx.x * gstride, 0, 16 /* sc1 */);
    }
}
// second half: wave 0 sweeps the partials of all blocks
template <int PT, bool XS /* compiled with the cross-rank level: the single-rank instantiation carries none of it (its live ranges cost the
                           register-starved <40, 19, 9, 512> kernel 15 more spill slots and 1.5 % of the headline sweep) */>
__device__ __forceinline__ bool grid_collect(int step, unsigned ebase, char* __restrict__ sync, int* __restrict__ err,
                                             double* sm, double* out, unsigned gstride /* bytes between the granules of two blocks: 128 = own line, 16 = packed */,
                                             const kk_xs_dev& xs /* row-sharded context: the sum over the ranks follows (kk_xsync.h) */, long long timeout_ticks,
                                             bool give_up_late = false /* test hook: this rank publishes its partial of THIS reduction and then declares the launch lost */) {
    const int G = gridDim.x;
    const unsigned epoch = ebase + (unsigned)step + 1u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, 2 * G * KK_SYNC_LINE, 0x00020000);
    const unsigned set_off = (unsigned)(step & 1) * (unsigned)G * KK_SYNC_LINE;
    if (threadIdx.x < 64) {   // wave 0 sweeps
        const int lane = threadIdx.x;
        const long long t0 = wall_clock64();
        double total = 0;
        int good = 1;
        bool origin = false;   // the failure (if any) is this wave's own timeout / the test hook, not an observed flag or a peer's abort
        for (;;) {
            asm volatile("" ::: "memory");   // compiler barrier: the granule loads below must be re-issued by every pass (the
                                             // buffer-load builtin is a plain read to LLVM and may otherwise be hoisted out of the spin loop)
            bool ok = true;
            double x = 0;
            // (the error flag travels with the first batch instead of costing a failed pass a round trip of its own)
            const int errv = __hip_atomic_load(err, RLX_AGENT);
            // 256 blocks' granules per batch, ALL loads of a lane issued before the first tag is looked at: written as
            // `for (b = lane; b < G; b += 64) { load; test; }` the sweep makes G / 64 = 4 dependent memory round trips
            // (~0.9 us each under streaming load) per pass
            for (int b0 = 0; b0 < G; b0 += 256) {
                v4u t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = b0 + i * 64 + lane;
                    const int bb = b < G ? b : 0;
                    t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, set_off + (unsigned)bb * gstride, 0, 16 /* sc1 */);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // summation order: b ascending per lane, then across the lanes (wave_sum)
                    if (b0 + i * 64 + lane < G) {
                        ok = ok && t[i].x == epoch && t[i].w == epoch;
                        x += __longlong_as_double((long long)(((unsigned long long)t[i].y << 32) | t[i].z));
                    }
                }
            }
            if (__all(ok)) { total = wave_sum(x); break; }
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > timeout_ticks || errv) { good = 0; origin = !errv; break; }
        }
        if (XS && xs.world > 0) {   // level 2: every block of this rank holds the same bits of the rank's partial -- now the sum over the ranks
            double t2 = 0;
            if (good && give_up_late) {   // what a rank looks like to its peers when ITS wait for one of them ran out a moment before that peer arrived
                if (blockIdx.x == 0) xs_publish(xs, (unsigned)step, 1, total);
                good = 0; origin = true;
            }
            if (good && !xs_allreduce(xs, (unsigned)step, 1, total, err, timeout_ticks, t2, &origin)) good = 0;
            total = t2;   // (lane 0: value 0)
            if (!good && origin && lane == 0) xs_abort(xs);   // this chip's own wait ran out: the peers must not wait for it (a block that only OBSERVED the
                                                              // local flag or a peer's abort writes nothing -- kk_xsync.h)
        }
        if (lane == 0) {
            if (!good) __hip_atomic_store(err, 1, RLX_AGENT);
            sm[0] = total;
            sm[1] = good ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    const double total = sm[0];
    const bool good = sm[1] != 0.0;
    __syncthreads();
    *out = total;
    return good;
}

// Row ownership: grid-row i (i < NV) is the contiguous span [i*stride, (i+1)*stride) of the vector, stride = G*PT*2; thread
// t of block b owns the double2 at element offset (b*PT + t)*2 inside every grid-row.  All streams go through buffer
// descriptors (base = the column, num_records = ld*8 bytes): the lane part of the address is ONE 32-bit byte offset shared
// by all loads, the grid-row part a scalar offset, and rows beyond ld read as zero / are not written (hardware bounds
// check) -- no per-load 64-bit address registers and no tail masks, which is what lets 40 doubles of w per thread live in
// the 128 registers of a 1024-thread block.
__device__ __forceinline__ d2 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, bool nt) {
    const v4u t = nt ? __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2) : __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    d2 o;
    o.x = __longlong_as_double((long long)(((unsigned long long)t.y << 32) | t.x));
    o.y = __longlong_as_double((long long)(((unsigned long long)t.w << 32) | t.z));
    return o;
}
#ifndef KK_PERSIST_ST_AUX
#define KK_PERSIST_ST_AUX 2     // non-temporal commit store of the work vector (headline +0.75 % in a same-box A/B: the apply that follows runs faster)
#endif
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, d2 v) {
    const unsigned long long a = (unsigned long long)__double_as_longlong(v.x), b = (unsigned long long)__double_as_longlong(v.y);
    v4u t;
    t.x = (unsigned)a; t.y = (unsigned)(a >> 32); t.z = (unsigned)b; t.w = (unsigned)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, KK_PERSIST_ST_AUX);
}
// descriptor of one column.  The inputs are wave-uniform by construction; passing them through readfirstlane makes that
// PROVABLE to the compiler (a loop-carried column pointer may otherwise be treated as divergent and every buffer access
// turned into a waterfall loop: cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t col_rsrc(const double* p, int64_t ld) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane((int)(ld * 8));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

// x -= s*p with the result tied to x's own registers: left to the register allocator the updated work vector migrates
// into the registers of the loaded operand, old and new copies overlap and the 40-doubles-per-thread budget is lost to
// copies (measured: 164 spilled registers at NV = 20 without this)
__device__ __forceinline__ void fnma_inplace(double& x, double s, double p) {
    asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(x) : "v"(s), "v"(p));
}

// The first NL grid-rows of every basis vector are PARKED IN LDS between the step that reads the vector as q_next (for the
// inner product) and the following step that needs it again as q_prev (for the pending axpy): thread t keeps its own
// double2 of grid-row i at lq[i * PT] (consecutive lanes, consecutive 16-byte slots: conflict-free ds_write_b128 /
// ds_read_b128; only the owning thread ever touches a slot, so program order is all the synchronisation there is).  With
// 512 threads per CU the 160 KB of LDS hold 19 of the (at N = 10^7) 39 grid-rows: the second read of a basis vector
// shrinks from 8 N to ~4.1 N bytes of fabric traffic -- 12.1 N per vector instead of the 16 N of ANY projection-based
// pass (project + unproject read the basis twice), which is what lets the reference's sequential order overtake them.
template <int NV, int NL, int NR, int PT, int B, bool NTPREV, bool NORM /* false: axpy + dot with q_next, true: axpy + squared norm */,
          bool PREV_LDS /* grid-rows < NL of q_prev come from LDS, the NR after them from registers */>
__device__ __forceinline__ void persist_step(d2 (&wr)[NV], d2 (&qk)[NR > 0 ? NR : 1], d2 (&qpre)[B], __amdgpu_buffer_rsrc_t rp, __amdgpu_buffer_rsrc_t rn,
                                             double sp, unsigned sbytes, unsigned voff, d2* __restrict__ lq, double& a0, double& a1) {
#pragma unroll
    for (int i0 = 0; i0 < NV; i0 += B) {
        d2 p[B], q[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            if (i0 + u < NV) {
                if (PREV_LDS && i0 + u < NL) p[u] = lq[(i0 + u) * PT];
                else if (PREV_LDS && i0 + u < NL + NR) p[u] = qk[i0 + u - NL];
                else p[u] = bload(rp, voff, (unsigned)(i0 + u) * sbytes, NTPREV);
                // first read of the next vector: rows that will wait on chip are never read again -> non-temporal for them
                // (KK_PERSIST_NT_FIRST: 0 none, 1 the parked rows, 2 every row)
                if (!NORM) q[u] = (i0 == 0) ? qpre[u] : bload(rn, voff, (unsigned)(i0 + u) * sbytes, KK_PERSIST_NT_FIRST == 2 || (KK_PERSIST_NT_FIRST == 1 && i0 + u < NL + NR));   // batch 0 was requested before the grid reduction
            }
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            if (i0 + u < NV) {
                d2& x = wr[i0 + u];
                fnma_inplace(x.x, sp, p[u].x);
                fnma_inplace(x.y, sp, p[u].y);
                const d2 y = NORM ? x : q[u];
                if (u & 1) { a1 = fma(y.x, x.x, a1); a1 = fma(y.y, x.y, a1); }
                else { a0 = fma(y.x, x.x, a0); a0 = fma(y.y, x.y, a0); }
                if (!NORM && i0 + u < NL) lq[(i0 + u) * PT] = q[u];   // park q_next for the next step's axpy
                else if (!NORM && i0 + u < NL + NR) qk[i0 + u - NL] = q[u];
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the loads of one batch together: w stays the only long-lived register set
    }
}

// nsweeps strict MGS sweeps of w against V[:, 0:m) (+ an optional pending axpy w -= *carry_s * carry_q in front, + the
// squared norm of the result).  out_s[sweep * out_stride + j] = coefficient of sweep `sweep`, vector j.
// Every step has the same shape -- pending axpy with (q_prev, s_prev), then the inner product with q_next -- so that the
// work vector stays in ONE register set through the loop; a step with nothing pending (the first one without a carry)
// runs the axpy with s_prev = 0 against a column of V.
// APPLY (round 6): the work vector is not loaded but FORMED -- the Lanczos step's w = (A v) * xs - beta v_prev for a value-free 5-point grid stencil
// (even line length, lines starting at phase 0: BASELINE config 2's operator) with alpha0 = <v, w> summed by one more grid reduction (index -1, the
// other granule set than step 0's) and handed to the sweep as its pending coefficient: the apply launch, its 8 N bytes of store, this kernel's 8 N
// bytes of load and the launch gap between the two go (lanczos.jl:306-310 + orthonormal.jl:414-439 in one launch).  w has the bits of
// k_spmv_dia / k_spmv_dia_sw; alpha0 is summed in another order (to rounding).
struct persist_apply_args { const double* x; const double* vprev; const double* xs_dev; const double* bprev_dev; double bprev; double* alpha_out; int64_t nrows; dia_cst cst; };
template <int NV, int NL, int NR, int PT, bool NTPREV, bool XS = false, bool APPLY = false>
__global__ __launch_bounds__(PT) void k_mgs_persist(const double* __restrict__ V, int64_t ld, int m, int nsweeps,
                                                    double* __restrict__ w, const double* __restrict__ carry_q,
                                                    const double* __restrict__ carry_s, double* __restrict__ out_s,
                                                    int out_stride, double* __restrict__ nrm_out3,
                                                    char* __restrict__ sync, int* __restrict__ err, int fault, unsigned gstride,
                                                    unsigned ebase, int normalize, double* __restrict__ ok_out, double token, kk_xs_dev xs, long long timeout_ticks,
                                                    persist_apply_args ap) {
    __shared__ double sm[PT / 64];
    extern __shared__ d2 park[];   // NL * PT double2 (dynamic): the parked grid-rows of the current basis vector
    if (fault == 1 && blockIdx.x == 0) {   // test hook (option "persist_fault"): block 0 behaves like a block whose spin ran out
        if (threadIdx.x == 0) { __hip_atomic_store(err, 1, RLX_AGENT); if (XS) xs_abort(xs); }
        return;
    }
#ifndef KK_PERSIST_B512
#define KK_PERSIST_B512 4
#endif
    constexpr int B = (PT == 1024 && NV > 16) ? 2 : (PT == 512 ? KK_PERSIST_B512 : 4);   // loads in flight per stream and lane; 128-register budget at 1024 threads
    const unsigned sbytes = gridDim.x * PT * 16u;                      // one grid-row in bytes
    const unsigned voff = (blockIdx.x * PT + threadIdx.x) * 16u;       // this lane's byte offset inside a grid-row
    d2* lq = park + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rw = col_rsrc(w, ld);
    d2 wr[NV];
    const double* qp = carry_q ? carry_q : V;   // nothing pending: s_prev = 0 against a (finite) basis column
    double sp = 0.0;
    if (APPLY) {
        // grid-row i of this thread: rows i * srows + r0, r0 = (block * PT + thread) * 2 (the ownership of the loads this replaces); four grid-rows at a
        // time: centre pair, the two far pairs (aligned: rows and D are even), the two single neighbours, the v_prev pair -- all loads first
        // (branch-free: an index outside the operator reads x[0] and is replaced by 0), then the products in the slot order -D, -1, 0, +1, +D
        const double* __restrict__ x = ap.x;
        const double* __restrict__ vp = ap.vprev;
        const int64_t nr = ap.nrows, D = ap.cst.D;
        const int64_t srows = (int64_t)gridDim.x * PT * 2, r0 = ((int64_t)blockIdx.x * PT + threadIdx.x) * 2;
        const double xsv = ap.xs_dev ? *ap.xs_dev : 1.0;
        const double bp = ap.bprev_dev ? *ap.bprev_dev : ap.bprev;
        const double c0 = ap.cst.c[0], c1 = ap.cst.c[1], c2 = ap.cst.c[2], c3 = ap.cst.c[3], c4 = ap.cst.c[4];
        double dacc = 0.0;
#pragma unroll
        for (int i0 = 0; i0 < NV; i0 += 4) {
            d2 xc[4], xm[4], xq[4], pv[4];
            double xl[4], xr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u < NV) {
                    const int64_t row = (int64_t)(i0 + u) * srows + r0;
                    const bool in = row < nr;
                    const int64_t rc = in ? row : 0;
                    const bool okm = in && rc >= D, okp = in && rc + D < nr, okl = in && rc >= 1, okr = in && rc + 2 < nr;
                    xc[u] = ld2(x + rc);
                    pv[u] = ld2s(vp + rc);
                    const d2 a = ld2(x + (okm ? rc - D : 0)), b = ld2(x + (okp ? rc + D : 0));
                    const double l = x[okl ? rc - 1 : 0], r = x[okr ? rc + 2 : 0];
                    xm[u] = d2{okm ? a.x : 0.0, okm ? a.y : 0.0};
                    xq[u] = d2{okp ? b.x : 0.0, (okp && rc + D + 1 < nr) ? b.y : 0.0};
                    xl[u] = okl ? l : 0.0; xr[u] = okr ? r : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u < NV) {
                    const int64_t row = (int64_t)(i0 + u) * srows + r0;
                    const bool in = row < nr;
                    const int64_t i0l = (int64_t)((unsigned)(in ? row : 0) % (unsigned)D);   // position of the pair's first row inside its grid line
                    const double cw0 = i0l == 0 ? 0.0 : c1, ce1 = i0l + 2 == D ? 0.0 : c3;   // no -1 entry at position 0, no +1 entry at position D - 1
                    double s0 = 0.0, s1 = 0.0;
                    s0 = fma(c0, xm[u].x, s0); s1 = fma(c0, xm[u].y, s1);
                    s0 = fma(cw0, xl[u], s0);  s1 = fma(c1, xc[u].x, s1);
                    s0 = fma(c2, xc[u].x, s0); s1 = fma(c2, xc[u].y, s1);
                    s0 = fma(c3, xc[u].y, s0); s1 = fma(ce1, xr[u], s1);
                    s0 = fma(c4, xq[u].x, s0); s1 = fma(c4, xq[u].y, s1);
                    const double t0 = s0 * xsv, t1 = s1 * xsv;
                    d2 out{1.0 * t0, 1.0 * t1};
                    const d2 xs2{xc[u].x * xsv, xc[u].y * xsv};
                    out.x = fma(-bp, pv[u].x, out.x); out.y = fma(-bp, pv[u].y, out.y);
                    if (row + 1 >= nr) out.y = 0.0;
                    if (!in) out = d2{0.0, 0.0};
                    dacc = fma(xs2.x, out.x, dacc); dacc = fma(xs2.y, out.y, dacc);   // alpha0 = <v, w> after the three-term part (lanczos.jl:308)
                    wr[i0 + u] = out;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        double a0tot;
        grid_publish<PT>(dacc, -1, ebase, sync, sm, gstride);
        if (!grid_collect<PT, false>(-1, ebase, sync, err, sm, &a0tot, gstride, xs, timeout_ticks)) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) ap.alpha_out[0] = a0tot;
        sp = a0tot;
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) wr[i] = bload(rw, voff, (unsigned)i * sbytes, false);
        sp = carry_q ? *carry_s : 0.0;
    }
    const int nsteps = m * nsweeps;
    d2 qk[NR > 0 ? NR : 1];   // NR more grid-rows of the current basis vector parked in spare registers
    if (NL + NR > 0) {   // park the first q_prev (the carried vector, or the dummy that goes with s_prev = 0): every step then has ONE shape
        const __amdgpu_buffer_rsrc_t r0 = col_rsrc(qp, ld);
        // in batches of 4 loads -> 4 LDS writes: left alone, hipcc runs all NL loads through ONE register quad, i.e. NL
        // dependent memory round trips (~1 us each) at the head of every launch
#pragma unroll
        for (int i0 = 0; i0 < NL; i0 += 4) {
            d2 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u < NL) t[u] = bload(r0, voff, (unsigned)(i0 + u) * sbytes, false);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u < NL) lq[(i0 + u) * PT] = t[u];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) qk[i] = bload(r0, voff, (unsigned)(NL + i) * sbytes, false);
    }
    // The first batch of the NEXT basis vector is requested before the grid reduction of the current one and lands while the
    // blocks wait for each other: the registers it lands in (one batch of B loads) are idle during the reduction anyway, and
    // the step after the reduction starts on data instead of on a memory round trip.
    d2 qpre[B];
    {
        const __amdgpu_buffer_rsrc_t r0 = col_rsrc(V, ld);
#pragma unroll
        for (int u = 0; u < B; ++u) qpre[u] = bload(r0, voff, (unsigned)(u < NV ? u : 0) * sbytes, false);
    }
    for (int s = 0; s < nsteps; ++s) {
        const double* qn = V + (int64_t)(s % m) * ld;
        double a0 = 0, a1 = 0, total;
        persist_step<NV, NL, NR, PT, B, NTPREV, false, true>(wr, qk, qpre, col_rsrc(qp, ld), col_rsrc(qn, ld), sp, sbytes, voff, lq, a0, a1);
        // The first batch of the next vector goes out around the reduction: AFTER the block's partial has been
        // published -- a CU's memory instructions leave through one in-order queue, and a publication behind 32 KB of loads
        // reaches the fabric that much later (kk_kernels_panel.hip); the sweep then queues behind the batch, which is fine.
        const __amdgpu_buffer_rsrc_t r2 = col_rsrc(V + (int64_t)((s + 1) % m) * ld, ld);   // (wraps to column 0 after the last one: a valid address, the values are not used)
#ifndef KK_PERSIST_PUBFIRST
#define KK_PERSIST_PUBFIRST 1
#endif
#if !KK_PERSIST_PUBFIRST
#pragma unroll
        for (int u = 0; u < B; ++u) qpre[u] = bload(r2, voff, (unsigned)(u < NV ? u : 0) * sbytes, KK_PERSIST_NT_FIRST == 2 || (KK_PERSIST_NT_FIRST == 1 && u < NL + NR));
        __builtin_amdgcn_sched_barrier(0);
#endif
        grid_publish<PT>(a0 + a1, s, ebase, sync, sm, gstride);
#if KK_PERSIST_PUBFIRST
#pragma unroll
        for (int u = 0; u < B; ++u) qpre[u] = bload(r2, voff, (unsigned)(u < NV ? u : 0) * sbytes, KK_PERSIST_NT_FIRST == 2 || (KK_PERSIST_NT_FIRST == 1 && u < NL + NR));
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (!grid_collect<PT, XS>(s, ebase, sync, err, sm, &total, gstride, xs, timeout_ticks)) return;   // timeout: w in HBM is untouched
        if (blockIdx.x == 0 && threadIdx.x == 0) out_s[(s / m) * out_stride + (s % m)] = total;
        sp = total;
        qp = qn;
    }
    double inv = 1.0;
    bool scale = false;
    {   // last pending axpy, fused with the squared norm of the result
        double a0 = 0, a1 = 0, total;
        persist_step<NV, NL, NR, PT, B, NTPREV, true, true>(wr, qk, qpre, col_rsrc(qp, ld), col_rsrc(qp, ld), sp, sbytes, voff, lq, a0, a1);
        if (nrm_out3) {
            grid_publish<PT>(a0 + a1, nsteps, ebase, sync, sm, gstride);
            if (!grid_collect<PT, XS>(nsteps, ebase, sync, err, sm, &total, gstride, xs, timeout_ticks, XS && fault == 2 /* option "persist_fault_late" */)) return;
            // every block holds the same bits of |w|^2: the normalised commit below needs no second exchange
            const double rt = sqrt(total);
            inv = 1.0 / rt;
            scale = normalize && rt > 0.0 && inv <= 1.79769313486231570815e308;   // a zero (or overflowing) norm leaves w as it is; the host applies the same test
            if (blockIdx.x == 0 && threadIdx.x == 0) { nrm_out3[0] = total; nrm_out3[1] = rt; nrm_out3[2] = inv; }
        }
    }
    // commit: a block that timed out in any grid reduction raised the flag and left; the blocks that got through re-read it
    // here, so that either every block writes its rows of w back or (up to the microsecond around a 3 s timeout) none does
    // and HBM still holds the input -- the host then repeats the sweep on the launch-per-vector route (persist_check).
    // The completion token travels with the scalars of the sweep (one read-back instead of a second one for the flag).
    // Across ranks the abort word plays the part of the flag: a rank that arrives late passes the last reduction on the partials its peers
    // left before they gave up on it (kk_xsync.h).
    if (__hip_atomic_load(err, RLX_AGENT)) return;
    if (XS && xs.world > 0 && xs_aborted(xs)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ok_out[0] = token; ok_out[1] = scale ? 1.0 : inv; }   // SC_PERSIST_OK, SC_XS
    // `normalize`: scale!!(w, 1/|w|) of the NEXT expand! (factorizations/lanczos.jl:257, arnoldi.jl:209; orthonormalize!!
    // orthonormal.jl:522-527) folded into the write-back -- the same product w[i] * (1/|w|) k_scal forms, so the
    // stored vector has the bits of the separate pass, which is no longer needed (16 N bytes and one launch per expand!)
    // Scale in place FIRST, store afterwards, nothing in between.  (A VALU write to the data registers of a 16-byte buffer
    // store in the instruction right behind it can still reach the store: hipcc inserts the wait state only for stores
    // without an SGPR offset; with one, gfx950 picked up the NEXT grid-row's product in lanes 12-15 of every row of 16,
    // about one launch in a hundred -- found with the same commit in kk_kernels_panel.hip.)
    const double f = scale ? inv : 1.0;   // (x * 1.0 is x, bit for bit)
#pragma unroll
    for (int i = 0; i < NV; ++i) { wr[i].x *= f; wr[i].y *= f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) bstore(rw, voff, (unsigned)i * sbytes, wr[i]);
}

// ---- launcher ------------------------------------------------------------------------------
// One block per CU, all of them resident at once.  hipLaunchCooperativeKernel guarantees that -- and costs ~35 us of idle GPU
// per launch on this stack (the launch travels through a device-wide cooperative queue, fenced against the stream on both
// sides: 3.5 % of a 10M-row Lanczos iteration, 12 % of a 2M-row Arnoldi step).  An ORDINARY launch of num_cus blocks that
// each need a whole CU (512 threads x 256 registers, or 152 KB of LDS) is resident just the same whenever the kernels of this
// stream have the device to themselves, which in-order execution makes the normal case; when they have not (another process
// or stream holds CUs) a block that cannot start is exactly the situation the bounded spins already handle: the flag is
// raised after 3 s, no block commits, the sweep is repeated on the launch-per-vector route and the persistent route backs
// off (persist_check_at).  Option "persist_coop" = 1 restores the cooperative API.
// Spin budget of one persistent launch.  The kernels are ordinary launches (below): a block that is not resident -- the GPU is
// shared with another queue or process -- is waited for this long, then the launch gives up without committing and the sweep
// is repeated on the launch-per-vector route.  3 s (rounds 3-4) was four orders of magnitude above a sweep; now 50 x the time
// the sweep's bytes take at 2 TB/s + 1 ms, at least 20 ms, at most 3 s (option "persist_timeout_ms" > 0 fixes it).  A
// row-sharded launch also waits for its PEERS' launches, i.e. for their hosts: never less than 1 s there.
long long kk_persist_timeout_ticks(kk_ctx ctx, int64_t ld, int nvec, bool cross_rank) {
    double ms;
    if (ctx->persist_timeout_ms > 0) ms = ctx->persist_timeout_ms;
    else {
        const double est_ms = 1.0 + 1e3 * ((double)(nvec + 2) * (double)ld * 8.0) / 2e12;
        ms = std::min(3000.0, std::max(20.0, 50.0 * est_ms));
        if (cross_rank) ms = std::max(ms, 1000.0);
    }
    return (long long)(ms * 1e5);   // 100 MHz wall clock
}
int kk_launch_resident(kk_ctx ctx, const void* fn, int threads, void** args, size_t dyn_lds, const char* what) {
    const dim3 g(ctx->num_cus), b(threads);
    hipError_t e = ctx->persist_coop ? hipLaunchCooperativeKernel(fn, g, b, args, dyn_lds, ctx->stream)
                                     : hipLaunchKernel(fn, g, b, args, dyn_lds, ctx->stream);
    if (e != hipSuccess) {
        kk_set_error("launch of %s (%d blocks of %d threads, %zu bytes of dynamic LDS) failed: %s", what, ctx->num_cus, threads, dyn_lds, hipGetErrorString(e));
        return KK_ERR_HIP;
    }
    return KK_OK;
}

// Eligible when the vector fits the register file of the chip (NV <= 20 double2 per thread at 1024 threads per CU, 40 at
// 512) and the blocks fit the synchronisation area.  On a row-sharded context the sum over the ranks must be available
// INSIDE the launch (kk_xs_on: every rank has every peer's sync area mapped, kk_comm_init) -- an RCCL all-reduce per vector
// cannot be issued from a kernel; without it the sharded sweep takes the low-synchronisation route.
int64_t kk_mgs_persist_capacity(kk_ctx ctx) {   // rows of a work vector the register file of the chip can hold
    const int pt = ctx->persist_threads;
    return (int64_t)ctx->num_cus * pt * 2 * (pt == 1024 ? 20 : 40);
}
bool kk_mgs_persist_eligible(kk_ctx ctx, int64_t ld_local, int m, int nsweeps) {
    if (!ctx->mgs_persist || (kk_sharded(ctx) && !kk_xs_on(ctx)) || !ctx->d_sync) return false;
    const int64_t ld = kk_dec_ld(ctx, ld_local);   // cross-rank context: the longest shard of the slab must fit (same answer on every rank)
    if (ctx->num_cus > KK_SYNC_MAX_BLOCKS || ld * 8 >= ((int64_t)1 << 31)) return false;
    return ld <= kk_mgs_persist_capacity(ctx);
}

// grid-rows parked in LDS: what fits next to the reduction scratch in the 160 KB of a CU, at most all of them; 0 = off
template <int NV, int PT>
struct persist_park { static constexpr int full = (160 * 1024 - 256) / (PT * 16); static constexpr int n = NV < full ? NV : full; };

template <int NV, int NL, int NR, int PT, bool NT, bool XS, bool APPLY = false>
static int launch_persist_inst2(kk_ctx ctx, void** args);
template <int NV, int NL, int NR, int PT, bool NT>
static int launch_persist_inst(kk_ctx ctx, void** args, bool xs_on, bool apply = false) {
    if (apply) {   // (the in-kernel apply exists for the default shape only: 512 threads, parked rows in LDS + registers, non-temporal second read, one rank)
        if constexpr (PT == 512 && NT && NR > 0) return launch_persist_inst2<NV, NL, NR, PT, NT, false, true>(ctx, args);
        kk_set_error("k_mgs_persist: in-kernel apply requested for an instantiation that has none (internal error)");
        return KK_ERR_UNSUPPORTED;
    }
    return xs_on ? launch_persist_inst2<NV, NL, NR, PT, NT, true>(ctx, args) : launch_persist_inst2<NV, NL, NR, PT, NT, false>(ctx, args);
}
template <int NV, int NL, int NR, int PT, bool NT, bool XS, bool APPLY>
static int launch_persist_inst2(kk_ctx ctx, void** args) {
    const void* fn = (const void*)k_mgs_persist<NV, NL, NR, PT, NT, XS, APPLY>;
    const size_t dyn = (size_t)NL * PT * sizeof(double) * 2;
    // the opt-in to more than 64 KB of dynamic LDS is a per-DEVICE attribute of the function: one call per instantiation and
    // device (a process may drive contexts on several GPUs), made with the context's device current
    static bool configured[KK_MAX_DEVICES] = {};
    const int dev = ctx->device;
    if (dyn > 0 && (dev < 0 || dev >= KK_MAX_DEVICES || !configured[dev])) {
        hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (ea != hipSuccess) return kk_hip_fail(ea, "hipFuncSetAttribute(k_mgs_persist, MaxDynamicSharedMemorySize)", __FILE__, __LINE__);
        if (dev >= 0 && dev < KK_MAX_DEVICES) configured[dev] = true;
    }
    return kk_launch_resident(ctx, fn, PT, args, dyn, "k_mgs_persist");
}
// grid-rows of the current basis vector kept in spare registers on top of the LDS-parked ones: only where the work vector
// leaves room (512 threads: 256 registers per lane, w takes 4 NV of them) -- KK_PERSIST_NR picks the count at build time
#ifndef KK_PERSIST_NR
#define KK_PERSIST_NR 9   // (A/B on the headline sweep, B = 4: NR 0 / 4 / 6 / 8 / 9 -> 976 / 1094 / 1186 / 1240 / 1259 it/s: the rows that are NOT parked
                         // must fit the XCD's L2 next to everything else -- 12 of them still do, 14 no longer)
#endif

template <int NV, int PT>
static int launch_persist(kk_ctx ctx, void** args, bool ntprev, bool xs_on, bool apply = false) {
    constexpr int NL = persist_park<NV, PT>::n;
    // (256-thread blocks -- one wave per SIMD with the whole 512-register file -- were tried: w then takes 320 registers per
    // lane and 39 + 12 of 77 grid-rows can be parked, no more than the 27 of 39 here: the on-chip capacity is what it is)
    constexpr int NR = (PT == 512 && NV - NL > 0) ? (NV - NL < KK_PERSIST_NR ? NV - NL : KK_PERSIST_NR) : 0;
    if (ctx->persist_lds == 2) return ntprev ? launch_persist_inst<NV, NL, NR, PT, true>(ctx, args, xs_on, apply) : launch_persist_inst<NV, NL, NR, PT, false>(ctx, args, xs_on);
    if (ctx->persist_lds) return ntprev ? launch_persist_inst<NV, NL, 0, PT, true>(ctx, args, xs_on) : launch_persist_inst<NV, NL, 0, PT, false>(ctx, args, xs_on);
    return ntprev ? launch_persist_inst<NV, 0, 0, PT, true>(ctx, args, xs_on) : launch_persist_inst<NV, 0, 0, PT, false>(ctx, args, xs_on);
}

// `normalize`: store w / |w| instead of w (needs nrm_out3; a zero norm leaves w unscaled -- kk_persist_norm_applies is the
// host's copy of the kernel's test).  The completion token lands in the scalar workspace (SC_PERSIST_OK) and is checked by
// persist_check after the read-back of the sweep's scalars.
// k_mgs_persist can form the Lanczos step's w itself: value-free 5-point stencil with an even line length whose lines start at phase 0, no ghost
// columns, one rank, the default kernel shape (512 threads, rows parked in LDS and registers, non-temporal second read), the vector on the strict kernel
bool kk_sweep_apply_ok_persist(kk_ctx ctx, const kk_sparse_dev& M, int64_t ld, int m) {
    return ctx->persist_apply && ctx->spmv_dia && ctx->spmv_dia_const && !kk_sharded(ctx) && !ctx->allreduce && !(ctx->comm && ctx->comm->active) && M.format == 0 &&
           M.dia_D > 0 && M.dia_const && M.dia_pts == 5 && (M.dia_D & 1) == 0 && M.dia_phase == 0 && M.n_ghost == 0 && !M.halo && !M.plan &&
           M.nrows * 8 < ((int64_t)1 << 31) && M.dia_D < ((int64_t)1 << 31) && ctx->persist_threads == 512 && ctx->persist_lds == 2 && ctx->persist_nt &&
           !kk_mgs_panel_eligible(ctx, ld) && kk_mgs_persist_eligible(ctx, ld, m, 1) &&
           (ld + (int64_t)ctx->num_cus * 512 * 2 - 1) / ((int64_t)ctx->num_cus * 512 * 2) > 24;   // (NV 32 / 40: the instantiations with rows parked in registers -- what 6 M rows and more take)
}

int kk_launch_mgs_persist(kk_ctx ctx, const double* V, int64_t ld, int m, int nsweeps, double* w, const double* carry_q,
                          const double* carry_s, double* out_s, int out_stride, double* nrm_out3, bool normalize_w, const kk_sweep_apply* apply) {
    const int pt = ctx->persist_threads;
    const int nv = (int)((ld + (int64_t)ctx->num_cus * pt * 2 - 1) / ((int64_t)ctx->num_cus * pt * 2));
    KK_HIP(hipSetDevice(ctx->device));   // the attribute call and the cooperative launch act on the CURRENT device
    char* sync = (char*)ctx->d_sync;
    int* err = (int*)((char*)ctx->d_sync + KK_SYNC_ERR_OFFSET);
    // epochs: nsteps + 1 reductions, tags ebase + 1 .. ebase + nsteps + 1; unique over the life of the context, so that the
    // granule area never needs clearing (it is zeroed at creation and when the 32-bit epoch counter is about to wrap)
    const unsigned need = (unsigned)(m * nsweeps) + 2u;
    if (ctx->persist_epoch > 0xffffffffu - need - 2u) {
        KK_HIP(hipMemsetAsync(sync, 0, (size_t)KK_SYNC_ERR_OFFSET, ctx->stream));
        ctx->persist_epoch = 0;
    }
    const bool do_apply = apply && apply->on;
    // (in-kernel apply: one more reduction, index -1 = tag ebase + 0, which must not be 0 -- the tag of the zeroed area: the launch's tags start one later)
    unsigned ebase = ctx->persist_epoch + (do_apply ? 1u : 0u);
    ctx->persist_epoch += need + (do_apply ? 1u : 0u);
    int fault = 0;
    if (ctx->persist_fault > 0) { --ctx->persist_fault; fault = 1; }
    int normalize = (normalize_w && nrm_out3) ? 1 : 0;
    unsigned gstride = ctx->persist_sync ? 16u : (unsigned)KK_SYNC_LINE;
    persist_apply_args ap = persist_apply_args();
    if (do_apply) {
        ap.x = apply->x; ap.vprev = apply->f.vprev; ap.xs_dev = apply->f.xscale_dev; ap.bprev_dev = apply->f.bprev_dev; ap.bprev = apply->f.bprev;
        ap.alpha_out = apply->f.dot_out; ap.nrows = apply->M->nrows;
        for (int q = 0; q < 9; ++q) ap.cst.c[q] = apply->M->dia_c[q];
        ap.cst.phase = apply->M->dia_phase; ap.cst.D = apply->M->dia_D;
        ++ctx->persist_apply_launches;
    }
    ctx->persist_token += 1.0;
    double token = ctx->persist_token;
    double* ok_out = ctx->ws + WS_SCAL + SC_PERSIST_OK;
    kk_xs_dev xs = kk_xs_launch_args(ctx, (unsigned)(m * nsweeps) + (nrm_out3 ? 1u : 0u));   // one cross-rank reduction per grid reduction (row-sharded context)
    if (!fault && ctx->persist_fault_late > 0 && xs.world > 0 && nrm_out3) { --ctx->persist_fault_late; fault = 2; }
    long long timeout_ticks = kk_persist_timeout_ticks(ctx, ld, m * nsweeps, xs.world > 0);
    void* args[] = {(void*)&V, (void*)&ld, (void*)&m, (void*)&nsweeps, (void*)&w, (void*)&carry_q, (void*)&carry_s,
                    (void*)&out_s, (void*)&out_stride, (void*)&nrm_out3, (void*)&sync, (void*)&err, (void*)&fault, (void*)&gstride,
                    (void*)&ebase, (void*)&normalize, (void*)&ok_out, (void*)&token, (void*)&xs, (void*)&timeout_ticks, (void*)&ap};
    const bool nt = ctx->persist_nt != 0;
    kk_prof_scope ps(ctx, "k_mgs_persist");
    if (pt == 1024) {
        if (nv <= 4) return launch_persist<4, 1024>(ctx, args, nt, xs.world > 0);
        if (nv <= 8) return launch_persist<8, 1024>(ctx, args, nt, xs.world > 0);
        if (nv <= 12) return launch_persist<12, 1024>(ctx, args, nt, xs.world > 0);
        if (nv <= 16) return launch_persist<16, 1024>(ctx, args, nt, xs.world > 0);
        if (nv <= 20) return launch_persist<20, 1024>(ctx, args, nt, xs.world > 0);
    } else {
        if (nv <= 8) return launch_persist<8, 512>(ctx, args, nt, xs.world > 0);
        if (nv <= 16) return launch_persist<16, 512>(ctx, args, nt, xs.world > 0);
        if (nv <= 24) return launch_persist<24, 512>(ctx, args, nt, xs.world > 0);
        if (nv <= 32) return launch_persist<32, 512>(ctx, args, nt, xs.world > 0, do_apply);
        if (nv <= 40) return launch_persist<40, 512>(ctx, args, nt, xs.world > 0, do_apply);
    }
    kk_set_error("kk_launch_mgs_persist: vector of %lld rows does not fit the register file", (long long)ld);
    return KK_ERR_UNSUPPORTED;
}

// ---- hand-shake of the cross-rank sync areas (kk_comm_init): every rank pushes a known value into every peer's area and waits
// for all of them, four times (both granule sets, twice) -- the mapping, the visibility of system-scope stores across the
// fabric and the tag protocol are exercised once with a short timeout BEFORE the persistent kernels rely on them.
// out[0] = 1 when every round produced sum_r (r + 1) * round, else 0.
__global__ __launch_bounds__(64) void k_xs_selftest(kk_xs_dev xs, int* __restrict__ err, long long timeout_ticks, int* __restrict__ out) {
    int ok = 1;
    for (unsigned round = 0; round < 4 && ok; ++round) {
        double total = 0;
        const double mine = (double)((xs.rank + 1) * (int)(round + 1));
        if (!xs_allreduce(xs, round, 1, mine, err, timeout_ticks, total)) ok = 0;
        else if (readlane_d(total, 0) != (double)(xs.world * (xs.world + 1) / 2 * (int)(round + 1))) ok = 0;
    }
    if (threadIdx.x == 0) out[0] = ok;
}
// ---- the price of one cross-rank reduction, measured by the hand-shake itself: `nred` reductions back to
// back from ONE wave per rank -- the first aligns the ranks, the 100 MHz wall clock brackets the rest.  out[0] = 1 on success,
// out[1] = ticks of reductions 1 .. nred - 1.  What a persistent launch pays per basis vector ON TOP of its single-chip
// reduction is this round trip: store into every peer's area over the fabric, poll of the own area.
__global__ __launch_bounds__(64) void k_xs_timing(kk_xs_dev xs, int* __restrict__ err, long long timeout_ticks, int nred, long long* __restrict__ out) {
    int ok = 1;
    long long t0 = 0;
    double v = (double)(xs.rank + 1);
    for (int r = 0; r < nred && ok; ++r) {
        double total = 0;
        if (!xs_allreduce(xs, (unsigned)r, 1, v, err, timeout_ticks, total)) ok = 0;
        v = readlane_d(total, 0) * 0.5;   // (the next partial depends on this total: no reduction can be issued ahead)
        if (r == 0) t0 = wall_clock64();
        // a fabric on which a reduction takes milliseconds is not one to spin on: give the whole measurement 0.25 s (3.8 ms per reduction), then
        // report failure -- kk_comm_init leaves the RCCL routes in charge (first contact with a real node must cost seconds at most, never minutes)
        else if (wall_clock64() - t0 > 25000000ll) ok = 0;
    }
    const long long t1 = wall_clock64();
    if (!ok && threadIdx.x == 0) xs_abort(xs);   // (a rank that stops measuring tells the others: they leave at once instead of waiting for its next partial)
    if (threadIdx.x == 0) { out[0] = ok; out[1] = t1 - t0; }
}
int kk_launch_xs_timing(kk_ctx ctx, const kk_xs_dev& xs, int nred, long long* out_dev) {
    int* err = (int*)((char*)ctx->d_sync + KK_SYNC_ERR_OFFSET);
    hipLaunchKernelGGL(k_xs_timing, dim3(1), dim3(64), 0, ctx->stream, xs, err, 200000000ll /* 2 s */, nred, out_dev);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_xs_selftest(kk_ctx ctx, const kk_xs_dev& xs, int* out_dev) {
    int* err = (int*)((char*)ctx->d_sync + KK_SYNC_ERR_OFFSET);
    hipLaunchKernelGGL(k_xs_selftest, dim3(1), dim3(64), 0, ctx->stream, xs, err, 200000000ll /* 2 s */, out_dev);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
