// Persistent PANEL kernel for modified Gram-Schmidt sweeps on vectors whose streams are SHORT -- the lengths where one grid
// reduction per basis vector, not bytes, bounds the register-resident kernel of kk_kernels_persist.hip (config 3 of
// BASELINE.json: 2M rows; strong-scaling shards).  SURVEY.md 8(a) rows a6 / a8; reference order being restated:
// src/orthonormal.jl:414-452 (MGS, MGS2), src/factorizations/arnoldi.jl:239-245, lanczos.jl:325-338.
//
//   * one block of 512 threads per CU (cooperative launch), the block's rows of the work vector w in registers for the
//     whole launch (NV double2 per thread, as in k_mgs_persist);
//   * the basis is taken P vectors at a time.  A panel is loaded ONCE into registers (P * NV double2 per thread) and serves
//     both its inner products and its update; while it is being used the NEXT panel is already on its way into a second
//     register set (ordinary buffer loads, consumed one loop iteration later), so the basis stream never stops for the
//     reduction: per panel the kernel costs max(stream of P vectors, one grid reduction), q is read from HBM exactly once
//     per sweep (8 N bytes per vector) and nothing is parked or re-read;
//   * ONE grid reduction per panel carries P (P + 1) / 2 values: d_i = <q_i, w> and the in-panel Gram entries g_ik =
//     <q_i, q_k>, k < i, formed from the registers that hold the panel anyway.  The MGS coefficients of the panel follow by
//     the exact forward substitution  s_i = d_i - sum_{k<i} g_ik s_k  ( = <q_i, w - sum_{k<i} s_k q_k> ), the algebra of
//     k_lowsync_solve restricted to the panel, with Gram entries of the vectors as they ARE (no bookkeeping, no
//     orthonormality assumption).  Between panels the order is strictly sequential.  P = 1 is the reference's strict order,
//     bit for bit the operations of k_mgs_persist (option mgs_mode = 0 forces it);
//   * the reduction itself: wave 0 publishes the block's P (P + 1) / 2 partials as 16-byte tagged granules (write-through
//     stores; value-major and packed, 4 KB per value) and sweeps those of all blocks with sc1 loads, summed in a fixed order
//     -- all blocks obtain the same bits.  What the first versions of this kernel taught (tools/panel_trace.hip):
//       - a CU's memory instructions leave through ONE in-order queue.  A publication (or a sweep) issued behind the ~128 KB
//         of loads of a panel reaches the fabric when most of them have been served, whichever wave issues it (a reduction
//         wave without rows and without loads of its own changes nothing: KK_PANEL_DW = 7): the reduction then ADDS to the
//         landing time of the panel.  Hence the order of a step: inner products -> barrier -> PUBLISH -> barrier -> request
//         the next panel -> sweep (queued behind it, served when the panel has landed, by which time every block has long
//         published: one pass) -> barrier -> update.  3.6 M rows: 6.8 -> 5.2 us per vector, equal to the kernel with the
//         reductions compiled out;
//       - a quarter of the next panel is requested BEFORE the inner products (KK_PANEL_EARLY): it keeps the memory pipe from
//         running empty during the hand-off and costs the publication ~0.2 us;
//       - non-temporal panel loads (every basis vector is read once): -8 %;
//       - rows owned contiguously per block instead of strided over the grid; granules packed instead of one line per block
//         (the sweeps of 256 blocks are fabric traffic, and their latency hides here);
//       - workgroup barriers are raw s_barrier (+ lgkmcnt(0)): a __syncthreads() makes hipcc drain the panel loads in flight
//         (vmcnt(0)) before every barrier.
//       - wider panels do not pay at 8 rows per thread: P = 3 (w + two panels = 224 registers of 256) 3.98 us per vector at 2 M
//         rows against 3.70 for P = 2.
//     What remains per panel is one pipeline fill (~1.5 us) plus the hand-off: 3.6 us per vector at 2 M rows (P = 2) against
//     5.1 for the projection pair and for k_mgs_persist, 2.4 for the bare stream.
#include "kk_internal.h"
#include "kk_device.h"
#include "kk_xsync.h"

// (spin budget of a launch: kk_persist_timeout_ticks, handed to the kernel in xs_timeout / `timeout_ticks`)
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define KK_PANEL_PT 512

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#ifndef KK_PANEL_AUX
#define KK_PANEL_AUX 2   // non-temporal: every basis vector is read once per sweep (7.2 vs 6.3 TB/s read ceiling, tools/hbm_peak.hip)
#endif
template <int AUX = 0>
__device__ __forceinline__ d2 pload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    d2 o;
    o.x = __longlong_as_double((long long)(((unsigned long long)t.y << 32) | t.x));
    o.y = __longlong_as_double((long long)(((unsigned long long)t.w << 32) | t.z));
    return o;
}
__device__ __forceinline__ void pstore(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, d2 v) {
    const unsigned long long a = (unsigned long long)__double_as_longlong(v.x), b = (unsigned long long)__double_as_longlong(v.y);
    v4u t;
    t.x = (unsigned)a; t.y = (unsigned)(a >> 32); t.z = (unsigned)b; t.w = (unsigned)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, 0);
}
// descriptor of one column (wave-uniform by construction, made provably so: cdna_hip_programming.md T20); bytes = 0 turns
// every load through it into zeros -- how the vectors beyond the end of the last panel are switched off without a branch
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pcol_rsrc(const double* p, int bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// workgroup barrier that leaves global loads in flight: LDS traffic of this wave done, then s_barrier (no vmcnt wait)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void fnma2(d2& x, double s, const d2& q) {
    asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(x.x) : "v"(s), "v"(q.x));
    asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(x.y) : "v"(s), "v"(q.y));
}

// ---- the block's two roles meet at two workgroup barriers per reduction: (1) partials of the data waves are in smA,
// (2) totals (and the timeout flag) of the reduction wave are in smB.
#ifdef KK_PANEL_TRACE   // tools/panel_trace.hip: wall-clock stamps of block 0 (one data wave, the reduction wave) per panel
__device__ long long* g_panel_trace = nullptr;
#define PTRACE(slot, p) do { if (g_panel_trace && blockIdx.x == 0 && (threadIdx.x == 64 || threadIdx.x == 0)) g_panel_trace[(p) * 16 + (slot)] = wall_clock64(); } while (0)
#define PTRACE_SET(slot, p, v) do { if (g_panel_trace && blockIdx.x == 0 && threadIdx.x == 0) g_panel_trace[(p) * 16 + (slot)] = (v); } while (0)
#else
#define PTRACE(slot, p) do { } while (0)
#define PTRACE_SET(slot, p, v) do { } while (0)
#endif
#ifndef KK_PANEL_DW
#define KK_PANEL_DW 8                       // data waves per block: 8 = every wave holds rows and wave 0 reduces on the side; 7 = wave 0 reduces only
#endif
#define KK_PANEL_W0 (8 - KK_PANEL_DW)       // first data wave
#define KK_PANEL_DT (KK_PANEL_DW * 64)      // data threads per block
#define KK_PANEL_NVMID (KK_PANEL_DW == 8 ? 8 : 9)   // grid-rows of the middle instantiation: what a 2M-row vector (config 3) needs

// wave 0, between barriers (1) and (1b): publish the block's partials.  Granule (v, block) of a set sits at ((v * G + block) * 16) bytes.
__device__ __forceinline__ void panel_publish(int nval, unsigned epoch, int set, char* __restrict__ sync, const double* smA) {
    const int G = gridDim.x;
    const int lane = threadIdx.x;
    const unsigned set_bytes = (unsigned)G * 16u * 8u;   // room for 8 values per set
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, 2 * (int)set_bytes, 0x00020000);
    if (lane < nval) {   // lane v publishes the block's partial of value v
        double b = 0;
#pragma unroll
        for (int k = KK_PANEL_W0; k < 8; ++k) b += smA[k * 8 + lane];   // fixed order
        const unsigned long long bits = (unsigned long long)__double_as_longlong(b);
        v4u t;
        t.x = epoch; t.y = (unsigned)(bits >> 32); t.z = (unsigned)bits; t.w = epoch;
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, (unsigned)set * set_bytes + ((unsigned)lane * (unsigned)G + blockIdx.x) * 16u, 0, 16 /* sc1 */);
    }
}
// wave 0, between barriers (1b) and (2): sweep the partials of all blocks, totals to smB[0 .. nval), timeout flag to smB[8]
__device__ __forceinline__ bool panel_sweep(int nval, unsigned epoch, int set, char* __restrict__ sync, int* __restrict__ err, double* smB, const kk_xs_dev& xs,
                                            unsigned xred /* index of this reduction within the launch */, long long timeout_ticks, int pidx = 0) {
    const int G = gridDim.x;
    const int lane = threadIdx.x;
    const unsigned set_bytes = (unsigned)G * 16u * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, 2 * (int)set_bytes, 0x00020000);
    const unsigned set_off = (unsigned)set * set_bytes;
    const long long t0 = wall_clock64();
    int good = 1;
    bool origin = false;   // the failure (if any) is this wave's own timeout, not an observed flag / abort
    long long npass = 0;
    double mine = 0;   // lane v: this rank's partial of value v (row-sharded context)
    for (int v = 0; v < nval && good; ++v) {   // value after value: by the time value 0 is complete the others usually are too
        const unsigned voff_v = set_off + (unsigned)v * (unsigned)G * 16u;
        double total = 0;
        for (;;) {
            // compiler barrier: the buffer-load builtin is a plain read to LLVM -- without it the granule loads are hoisted
            // out of the spin loop as loop invariants and the wave polls registers (found the hard way: every launch timed out)
            asm volatile("" ::: "memory");
            ++npass;
            const int errv = __hip_atomic_load(err, RLX_AGENT);
            bool ok = true;
            double x = 0;
            for (int b0 = 0; b0 < G; b0 += 256) {
                v4u t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = b0 + i * 64 + lane;
                    const int bb = b < G ? b : 0;
                    t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_v + (unsigned)bb * 16u, 0, 16 /* sc1 */);
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
        if (xs.world > 0) { if (lane == v) mine = total; }
        else if (lane == 0) smB[v] = total;
    }
    if (xs.world > 0) {   // level 2: the sum over the ranks (kk_xsync.h); lanes v * 8 .. v * 8 + 7 receive the total of value v
        double t2 = 0;
        if (good && !xs_allreduce(xs, xred, nval, mine, err, timeout_ticks, t2, &origin)) good = 0;
        if (good && (lane & 7) == 0 && (lane >> 3) < nval) smB[lane >> 3] = t2;
        if (!good && origin && lane == 0) xs_abort(xs);   // (only the wave whose OWN wait ran out tells the peers: kk_xsync.h)
    }
    if (lane == 0 && !good) { __hip_atomic_store(err, 1, RLX_AGENT); smB[8] = 1.0; }
    PTRACE(9, pidx);   // totals complete
    PTRACE_SET(10, pidx, npass);
    (void)npass;
    return good != 0;
}
// The same sweep with the VALUES SPREAD OVER THE WAVES (round 5): wave v sweeps value v, all at once.  panel_sweep takes the values one
// after the other -- each a round trip of its own to the L2 (~1 us while the chip streams), and only the first hides behind the landing
// of the next panel: with three values (two vectors per reduction) two round trips per panel were exposed, 6.0 us per panel against
// 5.4 for the same traffic with a one-value reduction (tools/sstore_publish.hip).  Same granules, same summation order per value (blocks
// ascending per lane, wave_sum): the totals have the bits of panel_sweep.  Called by ALL waves between barriers (1b) and (2); smB[0 .. 8)
// totals, [8] failure flag, [9 .. 16) the rank's partials on a row-sharded context, where wave 0 runs the cross-rank level behind one more
// barrier.
template <int NVAL>
__device__ __forceinline__ void panel_sweep_par(unsigned epoch, int set, char* __restrict__ sync, int* __restrict__ err, double* smB, const kk_xs_dev& xs,
                                                unsigned xred, long long timeout_ticks) {
    static_assert(NVAL <= 7, "one wave per value, seven slots for the rank's partials");
    const int G = gridDim.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < NVAL) {
        const unsigned set_bytes = (unsigned)G * 16u * 8u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, 2 * (int)set_bytes, 0x00020000);
        const unsigned voff_v = (unsigned)set * set_bytes + (unsigned)wave * (unsigned)G * 16u;
        const long long t0 = wall_clock64();
        int good = 1;
        double total = 0;
        for (;;) {
            asm volatile("" ::: "memory");   // (the granule loads must be re-issued by every pass: see panel_sweep)
            const int errv = __hip_atomic_load(err, RLX_AGENT);
            bool ok = true;
            double x = 0;
            for (int b0 = 0; b0 < G; b0 += 256) {
                v4u t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = b0 + i * 64 + lane;
                    const int bb = b < G ? b : 0;
                    t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_v + (unsigned)bb * 16u, 0, 16 /* sc1 */);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (b0 + i * 64 + lane < G) {
                        ok = ok && t[i].x == epoch && t[i].w == epoch;
                        x += __longlong_as_double((long long)(((unsigned long long)t[i].y << 32) | t[i].z));
                    }
                }
            }
            if (__all(ok)) { total = wave_sum(x); break; }
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > timeout_ticks || errv) { good = errv ? -1 : 0; break; }
        }
        if (lane == 0) {
            // failure flag: 1.0 observed (local flag already raised), 2.0 this block's own wait ran out (originator: tells the peers below);
            // several waves may fail at once -- the larger value stays (positive doubles order like their bit patterns)
            if (good <= 0) { __hip_atomic_store(err, 1, RLX_AGENT); atomicMax((unsigned long long*)&smB[8], (unsigned long long)__double_as_longlong(good == 0 ? 2.0 : 1.0)); }
            smB[xs.world > 0 ? 9 + wave : wave] = total;
        }
    }
    if (xs.world > 0) {   // (uniform) level 2: the sum over the ranks, by wave 0, from the rank's partials the waves left in smB[9 ..]
        lds_barrier();
        if (wave == 0) {
            bool good = smB[8] == 0.0;
            bool origin = smB[8] == 2.0;
            const double mine = lane < NVAL ? smB[9 + lane] : 0.0;
            double t2 = 0;
            if (good && !xs_allreduce(xs, xred, NVAL, mine, err, timeout_ticks, t2, &origin)) good = false;
            if (good && (lane & 7) == 0 && (lane >> 3) < NVAL) smB[lane >> 3] = t2;
            if (!good && lane == 0) { if (origin) xs_abort(xs); __hip_atomic_store(err, 1, RLX_AGENT); smB[8] = 1.0; }
        }
    }
}
// one whole reduction as seen by a wave 0 that holds no rows (KK_PANEL_DW = 7)
__device__ __forceinline__ bool panel_reduce_sync(int nval, unsigned epoch, int set, char* __restrict__ sync, int* __restrict__ err, const double* smA,
                                                  double* smB, const kk_xs_dev& xs, unsigned xred, long long timeout_ticks, int pidx = 0) {
    lds_barrier();   // (1)
    panel_publish(nval, epoch, set, sync, smA);
    PTRACE(8, pidx);
    lds_barrier();   // (1b)
    const bool good = panel_sweep(nval, epoch, set, sync, err, smB, xs, xred, timeout_ticks, pidx);
    lds_barrier();   // (2)
    return good;
}

// data waves, first half: hand the NVAL per-thread partials to the reduction wave and wait until it has PUBLISHED the block's
// partials -- only then may the next panel be requested: a CU's memory instructions leave through one in-order queue, and a
// publication queued behind 126 KB of panel loads reaches the fabric when most of them have been served (measured: the whole
// reduction then ADDS to the landing time of the panel instead of hiding behind it).
template <int NVAL>
__device__ __forceinline__ void panel_handoff(const double (&acc)[NVAL], double* smA, unsigned epoch, int set, char* __restrict__ sync, int pidx = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PTRACE(1, pidx);   // partial sums formed (the panel has landed)
#ifdef KK_PANEL_NOSYNC   // tools/panel_trace.hip: the stream structure of the kernel without any reduction
    return;
#endif
#pragma unroll
    for (int v = 0; v < NVAL; ++v) {
        const double t = wave_sum(acc[v]);
        if (lane == 0) smA[wave * 8 + v] = t;
    }
    lds_barrier();   // (1)  partials in smA
    PTRACE(2, pidx);
    if (KK_PANEL_W0 == 0 && threadIdx.x < 64) panel_publish(NVAL, epoch, set, sync, smA);   // wave 0 also holds rows: it publishes before it requests
    lds_barrier();   // (1b) published
}
// second half: the totals (same bits in every thread of every block).  Returns false after a timeout anywhere on the chip.
template <int NVAL>
__device__ __forceinline__ bool panel_totals(const double (&acc)[NVAL], double (&tot)[NVAL], double* smB, unsigned epoch, int set, char* __restrict__ sync,
                                             int* __restrict__ err, const kk_xs_dev& xs, unsigned xred, long long timeout_ticks, int pidx = 0) {
#ifdef KK_PANEL_NOSYNC
#pragma unroll
    for (int v = 0; v < NVAL; ++v) tot[v] = acc[v] * 1e-30;
    return true;
#endif
#ifndef KK_PANEL_SWEEP_SERIAL
    if (KK_PANEL_W0 == 0) panel_sweep_par<NVAL>(epoch, set, sync, err, smB, xs, xred, timeout_ticks);   // wave v sweeps value v (its loads queue behind its own panel loads: fine)
#else
    if (KK_PANEL_W0 == 0 && threadIdx.x < 64) panel_sweep(NVAL, epoch, set, sync, err, smB, xs, xred, timeout_ticks, pidx);   // round-4 form: wave 0 takes the values one after the other
#endif
    lds_barrier();   // (2)
    PTRACE(3, pidx);   // totals available
#pragma unroll
    for (int v = 0; v < NVAL; ++v) tot[v] = smB[v];
    return smB[8] == 0.0;
}

// panel p = vectors (sweep-major sequence) s0 .. s0 + P - 1 of the nsteps = m * nsweeps vectors of the launch
// rows [J0, J1) of every vector of the panel
template <int NV, int P, int J0 = 0, int J1 = NV>
__device__ __forceinline__ void panel_issue(d2 (&q)[P][NV], const double* __restrict__ V, int64_t ld, int m, int s0, int nsteps, unsigned voff,
                                            unsigned sbytes, int64_t brow, int bbytes) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int s = s0 + i;
        const bool valid = s < nsteps;
        const __amdgpu_buffer_rsrc_t r = pcol_rsrc(V + (int64_t)((valid ? s : 0) % m) * ld + brow, valid ? bbytes : 0);
#pragma unroll
        for (int j = J0; j < J1; ++j) q[i][j] = pload<KK_PANEL_AUX>(r, voff, (unsigned)j * sbytes);
    }
}

// step of panel `cur`, first half: partial inner products <q_i, w> and in-panel Gram entries, handed to the reduction wave
template <int NV, int P>
__device__ __forceinline__ void panel_dots(const d2 (&wr)[NV], const d2 (&cur)[P][NV], double (&acc)[P * (P + 1) / 2], double* smA, unsigned ebase, char* sync,
                                           int pidx) {
    constexpr int NVAL = P * (P + 1) / 2;
#pragma unroll
    for (int v = 0; v < NVAL; ++v) acc[v] = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            acc[i] = fma(cur[i][j].x, wr[j].x, acc[i]);
            acc[i] = fma(cur[i][j].y, wr[j].y, acc[i]);
#pragma unroll
            for (int k = 0; k < i; ++k) {
                const int g = P + i * (i - 1) / 2 + k;
                acc[g] = fma(cur[i][j].x, cur[k][j].x, acc[g]);
                acc[g] = fma(cur[i][j].y, cur[k][j].y, acc[g]);
            }
        }
    }
    panel_handoff<NVAL>(acc, smA, ebase + (unsigned)pidx + 1u, pidx & 1, sync, pidx);
}
// second half: totals -> coefficients -> update of w
template <int NV, int P>
__device__ __forceinline__ bool panel_update(d2 (&wr)[NV], d2 (&cur)[P][NV], const double (&acc)[P * (P + 1) / 2], int s0, int nsteps, int m,
                                             double* smB, double* __restrict__ out_s, int out_stride, unsigned ebase, char* sync, int* err, const kk_xs_dev& xs, long long timeout_ticks, int pidx) {
    constexpr int NVAL = P * (P + 1) / 2;
    double tot[NVAL];
    if (!panel_totals<NVAL>(acc, tot, smB, ebase + (unsigned)pidx + 1u, pidx & 1, sync, err, xs, (unsigned)pidx, timeout_ticks, pidx)) return false;
    // (I + L) s = d, L = strictly lower in-panel Gram block: exact forward substitution, the same bits in every thread
    double s[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double t = tot[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t = fma(-tot[P + i * (i - 1) / 2 + k], s[k], t);
        s[i] = t;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int j = 0; j < NV; ++j) fnma2(wr[j], s[i], cur[i][j]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 64 * KK_PANEL_W0) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int sv = s0 + i;
            if (sv < nsteps) out_s[(sv / m) * out_stride + (sv % m)] = s[i];
        }
    }
    PTRACE(4, pidx);   // update done
    return true;
}

// nsweeps MGS sweeps of w against V[:, 0:m) (+ an optional pending axpy w -= *carry_s * carry_q in front, + the squared
// norm of the result, + the normalised commit): the interface of k_mgs_persist.
// APPLY (round 6): the work vector is not loaded but FORMED -- w = (A x) * xs for a value-free 5-point grid stencil (even line length D, lines
// starting at phase 0), the operator of BASELINE config 3: the Arnoldi step's separate apply launch, its 16 N bytes of store and this kernel's 16 N
// bytes of load go (arnoldi.jl:242 + orthonormal.jl:414-439 in one launch).  Same products in the same order as k_spmv_dia<5, U, true>: same bits.
struct panel_apply_args { const double* x; const double* xs_dev; int64_t nrows; dia_cst cst; };
template <int NV, int P, bool APPLY = false>
__global__ __launch_bounds__(KK_PANEL_PT) void k_mgs_panel(const double* __restrict__ V, int64_t ld, int m, int nsweeps, double* __restrict__ w,
                                                           const double* __restrict__ carry_q, const double* __restrict__ carry_s,
                                                           double* __restrict__ out_s, int out_stride, double* __restrict__ nrm_out3,
                                                           char* __restrict__ sync, int* __restrict__ err, int fault, unsigned ebase, int normalize,
                                                           double* __restrict__ ok_out, double token, kk_xs_dev xs, long long timeout_ticks, panel_apply_args ap) {
    __shared__ double smA[64];
    __shared__ double smB[16];
    if (fault == 1 && blockIdx.x == 0) {   // test hook (option "persist_fault"): block 0 behaves like a block whose spin ran out
        if (threadIdx.x == 0) { __hip_atomic_store(err, 1, RLX_AGENT); xs_abort(xs); }
        return;
    }
    const int nsteps = m * nsweeps;
    const int npanels = (nsteps + P - 1) / P;
    constexpr int NVAL = P * (P + 1) / 2;
    if (KK_PANEL_W0 == 0 && threadIdx.x == 0) smB[8] = 0.0;
    if (KK_PANEL_W0 == 1 && threadIdx.x < 64) {
        // ---------------- a reduction wave without rows
        if (threadIdx.x == 0) smB[8] = 0.0;
        lds_barrier();   // (0)
#ifdef KK_PANEL_NOSYNC
        return;
#endif
        for (int p = 0; p < npanels; ++p)
            if (!panel_reduce_sync(NVAL, ebase + (unsigned)p + 1u, p & 1, sync, err, smA, smB, xs, (unsigned)p, timeout_ticks, p)) return;
        if (nrm_out3) panel_reduce_sync(1, ebase + (unsigned)npanels + 1u, npanels & 1, sync, err, smA, smB, xs, (unsigned)npanels, timeout_ticks);
        return;
    }
    // ---------------- the data waves
    // Row ownership: block b holds the CONTIGUOUS rows [b * rpb, (b + 1) * rpb), rpb = nvr * 896 with nvr = ceil(ld / (G * 896))
    // <= NV; thread t its double2 number r * 448 + t of them, r < NV.  Every stream goes through a descriptor of the BLOCK's
    // slice of the column (base + b * rpb, as many bytes as the slice has inside the vector): rows r >= nvr and the tail of
    // the last blocks read as zeros / are not written, with no mask anywhere.  (Contiguous per block, not strided over the
    // grid as in k_mgs_persist: with 16 .. 32 loads per lane in flight a grid-strided thread touches 16 .. 32 windows 1.8 MB
    // apart at once, and the same request stream then ran at 3.6 TB/s -- tools/panel_trace.hip, reductions compiled out.)
    const unsigned dt = threadIdx.x - 64 * KK_PANEL_W0;
    const int nvr = (int)((ld + (int64_t)gridDim.x * KK_PANEL_DT * 2 - 1) / ((int64_t)gridDim.x * KK_PANEL_DT * 2));
    const int64_t rpb = (int64_t)nvr * KK_PANEL_DT * 2;
    const int64_t brow = (int64_t)blockIdx.x * rpb;
    const int64_t left = ld - brow;
    const int bbytes = (int)((left < 0 ? 0 : (left < rpb ? left : rpb)) * 8);
    const unsigned sbytes = KK_PANEL_DT * 16u;                           // one row of 448 double2 in bytes
    const unsigned voff = dt * 16u;                                      // this lane's byte offset inside such a row
    const __amdgpu_buffer_rsrc_t rw = pcol_rsrc(w + brow, bbytes);
    d2 wr[NV];
    d2 qa[P][NV], qb[P][NV];
    panel_issue<NV, P>(qa, V, ld, m, 0, nsteps, voff, sbytes, brow, bbytes);   // first panel on its way before anything else
    if (APPLY) {
        // w = (A x) * xs on this thread's row pairs: centre pair, the two far pairs (aligned: row and D are even) and the two single neighbours,
        // all loads of all pairs first (branch-free: an index outside the operator reads x[0] and is replaced by 0), then the products in the
        // slot order -D, -1, 0, +1, +D of k_spmv_dia
        const double* __restrict__ x = ap.x;
        const int64_t nr = ap.nrows, D = ap.cst.D;
        const double xsv = ap.xs_dev ? *ap.xs_dev : 1.0;
        const double c0 = ap.cst.c[0], c1 = ap.cst.c[1], c2 = ap.cst.c[2], c3 = ap.cst.c[3], c4 = ap.cst.c[4];
        d2 xc[NV], xm[NV], xp[NV];
        double xl[NV], xr[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t row = brow + ((int64_t)j * KK_PANEL_DT + dt) * 2;
            const bool in = j < nvr && row < nr;
            const int64_t rc = in ? row : 0;
            const bool okm = in && rc >= D, okp = in && rc + D < nr, okl = in && rc >= 1, okr = in && rc + 2 < nr;
            xc[j] = ld2(x + rc);
            const d2 a = ld2(x + (okm ? rc - D : 0)), b = ld2(x + (okp ? rc + D : 0));
            const double l = x[okl ? rc - 1 : 0], r = x[okr ? rc + 2 : 0];
            xm[j] = d2{okm ? a.x : 0.0, okm ? a.y : 0.0};
            xp[j] = d2{okp ? b.x : 0.0, (okp && rc + D + 1 < nr) ? b.y : 0.0};
            xl[j] = okl ? l : 0.0; xr[j] = okr ? r : 0.0;
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t row = brow + ((int64_t)j * KK_PANEL_DT + dt) * 2;
            const bool in = j < nvr && row < nr;
            const int64_t i0 = (int64_t)((unsigned)(in ? row : 0) % (unsigned)D);   // position of the pair's first row inside its grid line (rows, D < 2^31)
            const double cw0 = i0 == 0 ? 0.0 : c1, ce1 = i0 + 2 == D ? 0.0 : c3;     // no -1 entry at position 0, no +1 entry at position D - 1
            double s0 = 0.0, s1 = 0.0;
            s0 = fma(c0, xm[j].x, s0); s1 = fma(c0, xm[j].y, s1);
            s0 = fma(cw0, xl[j], s0);  s1 = fma(c1, xc[j].x, s1);
            s0 = fma(c2, xc[j].x, s0); s1 = fma(c2, xc[j].y, s1);
            s0 = fma(c3, xc[j].y, s0); s1 = fma(ce1, xr[j], s1);
            s0 = fma(c4, xp[j].x, s0); s1 = fma(c4, xp[j].y, s1);
            const double t0 = s0 * xsv, t1 = s1 * xsv;
            wr[j] = in ? d2{1.0 * t0, (row + 1 >= nr) ? 0.0 : 1.0 * t1} : d2{0.0, 0.0};
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) wr[j] = pload(rw, voff, (unsigned)j * sbytes);
    }
    if (carry_q) {   // pending axpy of the caller (Lanczos: w -= alpha0 v): one extra read of that vector, through the idle second panel set
        const __amdgpu_buffer_rsrc_t rc = pcol_rsrc(carry_q + brow, bbytes);
        const double cs = *carry_s;
#pragma unroll
        for (int j = 0; j < NV; ++j) qb[0][j] = pload(rc, voff, (unsigned)j * sbytes);   // all loads first: one round trip, not NV
#pragma unroll
        for (int j = 0; j < NV; ++j) fnma2(wr[j], cs, qb[0][j]);
    }
    lds_barrier();   // (0) (the timeout flag slot is initialised)
#ifndef KK_PANEL_EARLY
#define KK_PANEL_EARLY 4
#endif
    constexpr int EARLY = (NV * KK_PANEL_EARLY) / 16;   // rows of each vector of the next panel requested BEFORE this panel's partials are published
    double acc[NVAL];
    for (int p = 0; p < npanels; p += 2) {
        // per panel: inner products -> the block's partials are published -> request the NEXT panel -> totals -> update
        panel_issue<NV, P, 0, EARLY>(qb, V, ld, m, (p + 1) * P, nsteps, voff, sbytes, brow, bbytes);   // head of the next panel: keeps the memory pipe primed
        panel_dots<NV, P>(wr, qa, acc, smA, ebase, sync, p);
        PTRACE(0, p);
        panel_issue<NV, P, EARLY, NV>(qb, V, ld, m, (p + 1) * P, nsteps, voff, sbytes, brow, bbytes);   // the rest, in flight across this panel's reduction
        if (!panel_update<NV, P>(wr, qa, acc, p * P, nsteps, m, smB, out_s, out_stride, ebase, sync, err, xs, timeout_ticks, p)) return;   // timeout: w in HBM is untouched
        if (p + 1 >= npanels) break;
        panel_issue<NV, P, 0, EARLY>(qa, V, ld, m, (p + 2) * P, nsteps, voff, sbytes, brow, bbytes);
        panel_dots<NV, P>(wr, qb, acc, smA, ebase, sync, p + 1);
        PTRACE(0, p + 1);
        panel_issue<NV, P, EARLY, NV>(qa, V, ld, m, (p + 2) * P, nsteps, voff, sbytes, brow, bbytes);
        if (!panel_update<NV, P>(wr, qb, acc, (p + 1) * P, nsteps, m, smB, out_s, out_stride, ebase, sync, err, xs, timeout_ticks, p + 1)) return;
    }
    double inv = 1.0;
    bool scale = false;
    if (nrm_out3) {
        double an[1] = {0.0}, tot[1];
#pragma unroll
        for (int j = 0; j < NV; ++j) { an[0] = fma(wr[j].x, wr[j].x, an[0]); an[0] = fma(wr[j].y, wr[j].y, an[0]); }
        panel_handoff<1>(an, smA, ebase + (unsigned)npanels + 1u, npanels & 1, sync);
        if (!panel_totals<1>(an, tot, smB, ebase + (unsigned)npanels + 1u, npanels & 1, sync, err, xs, (unsigned)npanels, timeout_ticks)) return;
        const double rt = sqrt(tot[0]);
        inv = 1.0 / rt;
        scale = normalize && rt > 0.0 && inv <= 1.79769313486231570815e308;
        if (blockIdx.x == 0 && threadIdx.x == 64 * KK_PANEL_W0) { nrm_out3[0] = tot[0]; nrm_out3[1] = rt; nrm_out3[2] = inv; }
    }
    // commit (see k_mgs_persist): every block writes its rows back or -- flag raised by a block that timed out, abort word of a peer -- none does
    if (__hip_atomic_load(err, RLX_AGENT)) return;
    if (xs.world > 0 && xs_aborted(xs)) return;
    if (blockIdx.x == 0 && threadIdx.x == 64 * KK_PANEL_W0) { ok_out[0] = token; ok_out[1] = scale ? 1.0 : inv; }   // SC_PERSIST_OK, SC_XS
    // scale in place FIRST, store afterwards, nothing in between: a VALU write to the data registers of a 16-byte buffer store
    // in the instruction after it can reach the store (hipcc inserts the wait state only for stores WITHOUT an SGPR offset;
    // with one, gfx950 still picked up the NEXT row's product in lanes 12-15 of every row of 16 -- one launch in ~100)
    const double f = scale ? inv : 1.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) { wr[j].x *= f; wr[j].y *= f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NV; ++j) pstore(rw, voff, (unsigned)j * sbytes, wr[j]);
}

// ==========================================================================================================================
// Cross-panel LAG-1 form.  In k_mgs_panel the inner products of panel p + 1 wait for the update of panel p,
// which waits for the grid reduction of panel p: one reduction (+ one refill of the memory pipe, because the registers of panel
// p are not free for panel p + 2 before that) is exposed per panel -- 3.6 us per vector at 2 M rows against 2.4 us for the
// bare stream.  Here the reduction of panel p is taken OFF the critical path:
//   * the values of reduction R_p are formed from the work vector as it is BEFORE the update of panel p - 1:
//         d~_p = Q_p' w_(p-1),   C_p = Q_p' Q_(p-1)   (P x P, both panels are in registers anyway),   G_p = strictly lower Q_p' Q_p
//     and the MGS coefficients follow exactly:   rhs = d~_p - C_p s_(p-1)  ( = Q_p' (w_(p-1) - Q_(p-1) s_(p-1)) ),
//     s_p = (I + L_p)^-1 rhs -- the algebra of the in-panel correction applied between two panels; no orthonormality assumed;
//   * iteration p of the data waves:  partial sums of R_p -> hand-off;  totals of R_(p-1) (published one iteration ago) ->
//     s_(p-1) -> w -= Q_(p-1) s_(p-1);  the registers of Q_(p-1) take the loads of Q_(p+2).  THREE register-resident panels:
//     one being used up, one landed, one in flight -- the basis stream never waits for a reduction or a register;
//   * the reductions belong to wave 0, which holds NO rows (7 data waves of 64 lanes): its sweep of R_p spins while the data
//     waves stream, and -- having no panel loads of its own -- its granule loads are not queued behind any (in-order vmcnt).
//     Per panel the kernel costs max(stream of P vectors, one grid reduction).
// Same interface, epochs, granule sets, cross-rank level and commit as k_mgs_panel.  Not the reference's association of the
// operations (like panels of P > 1): auto mode only; mgs_mode 0 keeps the strict kernel.
//
// MEASURED (round 5, profiles/r05_panel_lag_ab.jsonl, Arnoldi MGS2 cycle of 60 on the convection-diffusion operator): slower
// than k_mgs_panel at every length -- 0.25 M rows 4738 vs 5932 it/s, 1 M 4289 vs 5552, 1.8 M 3520 vs 4787; with ONE vector per
// reduction (the only width whose three panels fit the registers at 2 M rows) 3926 vs 4594.  What bounds it is the chain of wave
// 0 -- publish R_p, sweep R_p (2-3 passes of ~1 us under a streaming chip), barrier, publish R_(p+1) -- about 6-7 us per
// reduction, as long as the reduction k_mgs_panel exposes; and it holds two vectors per reduction where k_mgs_panel holds three
// at <= 4 grid-rows.  The lag buys nothing while the reduction itself, not its position, is the cost.  Option "panel_lag" = 1
// selects it (default 0); tests/test_gpu_panel.py runs both.
// ==========================================================================================================================
// (kept compiled and tested: exact algebra, same interface)
// ==========================================================================================================================
#define KK_LAG_DT 448   // data threads per block (waves 1..7)
__device__ __forceinline__ void lag_publish(int nval, unsigned epoch, int set, char* __restrict__ sync, const double* smA) {
    const int G = gridDim.x;
    const int lane = threadIdx.x;
    const unsigned set_bytes = (unsigned)G * 16u * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, 2 * (int)set_bytes, 0x00020000);
    if (lane < nval) {
        double b = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) b += smA[k * 8 + lane];   // data waves 1..7, fixed order
        const unsigned long long bits = (unsigned long long)__double_as_longlong(b);
        v4u t;
        t.x = epoch; t.y = (unsigned)(bits >> 32); t.z = (unsigned)bits; t.w = epoch;
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, (unsigned)set * set_bytes + ((unsigned)lane * (unsigned)G + blockIdx.x) * 16u, 0, 16 /* sc1 */);
    }
}
// data waves, iteration p: partial sums of R_p from the landed panel `cur` (= Q_p), the panel before it `prev` (= Q_(p-1), zeros for
// p = 0) and the work vector as it stands -> smA
template <int NV, int P>
__device__ __forceinline__ void lag_partials(const d2 (&wr)[NV], const d2 (&cur)[P][NV], const d2 (&prev)[P][NV], double* smA) {
    constexpr int NVAL = P + P * P + P * (P - 1) / 2;
    double acc[NVAL];
#pragma unroll
    for (int v = 0; v < NVAL; ++v) acc[v] = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            acc[i] = fma(cur[i][j].x, wr[j].x, acc[i]);
            acc[i] = fma(cur[i][j].y, wr[j].y, acc[i]);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                acc[P + i * P + k] = fma(cur[i][j].x, prev[k][j].x, acc[P + i * P + k]);
                acc[P + i * P + k] = fma(cur[i][j].y, prev[k][j].y, acc[P + i * P + k]);
            }
#pragma unroll
            for (int k = 0; k < i; ++k) {
                const int g = P + P * P + i * (i - 1) / 2 + k;
                acc[g] = fma(cur[i][j].x, cur[k][j].x, acc[g]);
                acc[g] = fma(cur[i][j].y, cur[k][j].y, acc[g]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NVAL; ++v) {
        const double t = wave_sum(acc[v]);
        if (lane == 0) smA[wave * 8 + v] = t;
    }
}
// data waves: totals of R_q (q = the panel held in `pan`) -> coefficients (sp: in = s_(q-1), out = s_q) -> w -= pan s_q
template <int NV, int P>
__device__ __forceinline__ void lag_update(d2 (&wr)[NV], const d2 (&pan)[P][NV], const double* tot, double (&sp)[P], int s0, int nsteps, int m,
                                           double* __restrict__ out_s, int out_stride) {
    double s[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        double t = tot[i];
#pragma unroll
        for (int k = 0; k < P; ++k) t = fma(-tot[P + i * P + k], sp[k], t);              // - C s_(q-1)
#pragma unroll
        for (int k = 0; k < i; ++k) t = fma(-tot[P + P * P + i * (i - 1) / 2 + k], s[k], t);   // forward substitution inside the panel
        s[i] = t;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int j = 0; j < NV; ++j) fnma2(wr[j], s[i], pan[i][j]);
        sp[i] = s[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 64) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int sv = s0 + i;
            if (sv < nsteps) out_s[(sv / m) * out_stride + (sv % m)] = s[i];
        }
    }
}

template <int NV, int P>
__global__ __launch_bounds__(KK_PANEL_PT) void k_mgs_panel_lag(const double* __restrict__ V, int64_t ld, int m, int nsweeps, double* __restrict__ w,
                                                               const double* __restrict__ carry_q, const double* __restrict__ carry_s,
                                                               double* __restrict__ out_s, int out_stride, double* __restrict__ nrm_out3,
                                                               char* __restrict__ sync, int* __restrict__ err, int fault, unsigned ebase, int normalize,
                                                               double* __restrict__ ok_out, double token, kk_xs_dev xs, long long timeout_ticks) {
    constexpr int NVAL = P + P * P + P * (P - 1) / 2;
    static_assert(NVAL <= 8, "one reduction carries at most 8 values");
    __shared__ double smA[64];   // [wave][8] partial sums of the data waves
    __shared__ double smB[32];   // [set][16]: totals of a reduction, [8] = failure flag
    if (fault && blockIdx.x == 0) {
        if (threadIdx.x == 0) { __hip_atomic_store(err, 1, RLX_AGENT); xs_abort(xs); }
        return;
    }
    const int nsteps = m * nsweeps;
    const int npanels = (nsteps + P - 1) / P;
    if (threadIdx.x < 64) {
        // ---------------- the reduction wave
        if (threadIdx.x == 0) { smB[8] = 0.0; smB[24] = 0.0; }
        lds_barrier();   // (0)
        for (int p = 0; p < npanels; ++p) {
            lds_barrier();   // (1_p) partial sums of R_p are in smA
            lag_publish(NVAL, ebase + (unsigned)p + 1u, p & 1, sync, smA);
            lds_barrier();   // (2_p) smA may be overwritten; the totals of R_(p-1) (swept before (1_p)) are in smB[(p-1) & 1]
            if (!panel_sweep(NVAL, ebase + (unsigned)p + 1u, p & 1, sync, err, smB + (p & 1) * 16, xs, (unsigned)p, timeout_ticks, p)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the flag is in LDS: the data waves see it behind their next barrier pair
                return;                                               // (a wave that has ended no longer counts in s_barrier)
            }
        }
        lds_barrier();   // (2_npanels) totals of the last panel
        if (nrm_out3) {
            lds_barrier();   // (1_n) partial of |w|^2
            lag_publish(1, ebase + (unsigned)npanels + 1u, npanels & 1, sync, smA);
            panel_sweep(1, ebase + (unsigned)npanels + 1u, npanels & 1, sync, err, smB + (npanels & 1) * 16, xs, (unsigned)npanels, timeout_ticks);
            lds_barrier();   // (2_n)
        }
        return;
    }
    // ---------------- the data waves: rows owned contiguously per block, as in k_mgs_panel, 448 double2 per grid-row
    const unsigned dt = threadIdx.x - 64;
    const int nvr = (int)((ld + (int64_t)gridDim.x * KK_LAG_DT * 2 - 1) / ((int64_t)gridDim.x * KK_LAG_DT * 2));
    const int64_t rpb = (int64_t)nvr * KK_LAG_DT * 2;
    const int64_t brow = (int64_t)blockIdx.x * rpb;
    const int64_t left = ld - brow;
    const int bbytes = (int)((left < 0 ? 0 : (left < rpb ? left : rpb)) * 8);
    const unsigned sbytes = KK_LAG_DT * 16u;
    const unsigned voff = dt * 16u;
    const __amdgpu_buffer_rsrc_t rw = pcol_rsrc(w + brow, bbytes);
    d2 wr[NV];
    d2 q0[P][NV], q1[P][NV], q2[P][NV];   // panels p = 0, 1, 2 (mod 3)
    panel_issue<NV, P>(q0, V, ld, m, 0, nsteps, voff, sbytes, brow, bbytes);
    panel_issue<NV, P>(q1, V, ld, m, P, nsteps, voff, sbytes, brow, bbytes);
#pragma unroll
    for (int j = 0; j < NV; ++j) wr[j] = pload(rw, voff, (unsigned)j * sbytes);
    if (carry_q) {   // pending axpy of the caller (Lanczos: w -= alpha0 v), through the set that is idle until panel 2 is requested
        const __amdgpu_buffer_rsrc_t rc = pcol_rsrc(carry_q + brow, bbytes);
        const double cs = *carry_s;
#pragma unroll
        for (int j = 0; j < NV; ++j) q2[0][j] = pload(rc, voff, (unsigned)j * sbytes);
#pragma unroll
        for (int j = 0; j < NV; ++j) fnma2(wr[j], cs, q2[0][j]);
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) q2[i][j] = d2{0.0, 0.0};   // "panel -1": C_0 = 0
    lds_barrier();   // (0)
    double sp[P];
#pragma unroll
    for (int i = 0; i < P; ++i) sp[i] = 0.0;
    // one iteration: cur = Q_p (landed), prev = Q_(p-1) (updates w now, then takes the loads of Q_(p+2)), the third set is in flight
#define LAG_STEP(cur, prev, p)                                                                                                           \
    do {                                                                                                                                 \
        lag_partials<NV, P>(wr, cur, prev, smA);                                                                                         \
        lds_barrier(); /* (1_p) */                                                                                                       \
        lds_barrier(); /* (2_p) */                                                                                                       \
        if ((p) > 0) {                                                                                                                   \
            const double* tot = smB + (((p) - 1) & 1) * 16;                                                                              \
            if (tot[8] != 0.0) return; /* timeout somewhere on the chip (or on a peer): w in HBM is untouched */                        \
            lag_update<NV, P>(wr, prev, tot, sp, ((p) - 1) * P, nsteps, m, out_s, out_stride);                                           \
        }                                                                                                                                \
        panel_issue<NV, P>(prev, V, ld, m, ((p) + 2) * P, nsteps, voff, sbytes, brow, bbytes);                                           \
    } while (0)
    int p = 0;
    for (;;) {
        LAG_STEP(q0, q2, p); if (++p >= npanels) break;
        LAG_STEP(q1, q0, p); if (++p >= npanels) break;
        LAG_STEP(q2, q1, p); if (++p >= npanels) break;
    }
#undef LAG_STEP
    lds_barrier();   // (2_npanels)
    {
        const double* tot = smB + ((npanels - 1) & 1) * 16;
        if (tot[8] != 0.0) return;
        const int last = (npanels - 1) % 3;   // uniform: the set that holds the last panel
        if (last == 0) lag_update<NV, P>(wr, q0, tot, sp, (npanels - 1) * P, nsteps, m, out_s, out_stride);
        else if (last == 1) lag_update<NV, P>(wr, q1, tot, sp, (npanels - 1) * P, nsteps, m, out_s, out_stride);
        else lag_update<NV, P>(wr, q2, tot, sp, (npanels - 1) * P, nsteps, m, out_s, out_stride);
    }
    double inv = 1.0;
    bool scale = false;
    if (nrm_out3) {
        double an = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) { an = fma(wr[j].x, wr[j].x, an); an = fma(wr[j].y, wr[j].y, an); }
        const double t = wave_sum(an);
        if ((threadIdx.x & 63) == 0) smA[(threadIdx.x >> 6) * 8] = t;
        lds_barrier();   // (1_n)
        lds_barrier();   // (2_n)
        const double* tot = smB + (npanels & 1) * 16;
        if (tot[8] != 0.0) return;
        const double rt = sqrt(tot[0]);
        inv = 1.0 / rt;
        scale = normalize && rt > 0.0 && inv <= 1.79769313486231570815e308;
        if (blockIdx.x == 0 && threadIdx.x == 64) { nrm_out3[0] = tot[0]; nrm_out3[1] = rt; nrm_out3[2] = inv; }
    }
    // commit (see k_mgs_persist / k_mgs_panel)
    if (__hip_atomic_load(err, RLX_AGENT)) return;
    if (xs.world > 0 && xs_aborted(xs)) return;
    if (blockIdx.x == 0 && threadIdx.x == 64) { ok_out[0] = token; ok_out[1] = scale ? 1.0 : inv; }
    const double f = scale ? inv : 1.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) { wr[j].x *= f; wr[j].y *= f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NV; ++j) pstore(rw, voff, (unsigned)j * sbytes, wr[j]);
}

// ---- launcher ------------------------------------------------------------------------------
// vectors of at most 16 rows of 512 double2 per block (4.19 M rows on 256 CUs): w plus two panels fit the 256 registers of a 512-thread block
int64_t kk_mgs_panel_capacity(kk_ctx ctx) { return (int64_t)ctx->num_cus * KK_PANEL_DT * 2 * 16; }
bool kk_mgs_panel_eligible(kk_ctx ctx, int64_t ld_local) {
    if (!ctx->mgs_panel || !ctx->mgs_persist || (kk_sharded(ctx) && !kk_xs_on(ctx)) || !ctx->d_sync) return false;
    const int64_t ld = kk_dec_ld(ctx, ld_local);   // cross-rank context: the longest shard of the slab must fit (same answer on every rank)
    if (ctx->num_cus > KK_SYNC_MAX_BLOCKS || ld * 8 >= ((int64_t)1 << 31)) return false;
    return ld <= kk_mgs_panel_capacity(ctx);
}

template <int NV, int P>
static int launch_panel_inst(kk_ctx ctx, void** args, bool apply = false) {
    if (apply) return kk_launch_resident(ctx, (const void*)k_mgs_panel<NV, P, true>, KK_PANEL_PT, args, 0, "k_mgs_panel");
    return kk_launch_resident(ctx, (const void*)k_mgs_panel<NV, P, false>, KK_PANEL_PT, args, 0, "k_mgs_panel");
}
// the panel kernel can apply the operator itself: value-free 5-point stencil with an even line length whose lines start at phase 0, no ghost columns
bool kk_sweep_apply_ok(kk_ctx ctx, const kk_sparse_dev& M, int64_t ld) {
    return ctx->panel_apply && ctx->spmv_dia && ctx->spmv_dia_const && !kk_sharded(ctx) && !ctx->allreduce && M.format == 0 && M.dia_D > 0 && M.dia_const && M.dia_pts == 5 &&
           (M.dia_D & 1) == 0 && M.dia_phase == 0 && M.n_ghost == 0 && !M.halo && !M.plan && M.nrows * 8 < ((int64_t)1 << 31) && M.dia_D < ((int64_t)1 << 31) &&
           kk_mgs_panel_eligible(ctx, ld) &&
           // (vectors of at most KK_PANEL_NVMID grid-rows per block: the 16-row instantiation has no registers left for the apply's operands)
           (kk_dec_ld(ctx, ld) + (int64_t)ctx->num_cus * KK_PANEL_DT * 2 - 1) / ((int64_t)ctx->num_cus * KK_PANEL_DT * 2) <= KK_PANEL_NVMID;
}

template <int NV, int P>
static int launch_panel_lag_inst(kk_ctx ctx, void** args) {
    return kk_launch_resident(ctx, (const void*)k_mgs_panel_lag<NV, P>, KK_PANEL_PT, args, 0, "k_mgs_panel_lag");
}

// panel width by vector length: what two register-resident panels + w leave room for (4 NV (1 + 2 P) <= ~200 registers)
int kk_mgs_panel_width(kk_ctx ctx, int64_t ld_local, bool strict) {
    if (strict) return 1;
    const int64_t ld = kk_dec_ld(ctx, ld_local);   // (every rank of a cross-rank context must sweep panels of the same width: ADVICE r5)
    const int nv = (int)((ld + (int64_t)ctx->num_cus * KK_PANEL_DT * 2 - 1) / ((int64_t)ctx->num_cus * KK_PANEL_DT * 2));
    const int by_size = nv <= 4 ? 3 : (nv <= KK_PANEL_NVMID ? 2 : 1);
    return ctx->panel_width > 0 ? std::min(ctx->panel_width, by_size) : by_size;
}

int kk_launch_mgs_panel(kk_ctx ctx, const double* V, int64_t ld, int m, int nsweeps, double* w, const double* carry_q,
                        const double* carry_s, double* out_s, int out_stride, double* nrm_out3, bool normalize_w, bool strict,
                        const kk_sweep_apply* apply) {
    // (the register tile is chosen for the longest shard of the slab: the kernel itself works out from its own ld how many of the
    // NV rows per lane exist locally -- the rest read as zeros)
    const int64_t ld_dec = kk_dec_ld(ctx, ld);
    const int nv = (int)((ld_dec + (int64_t)ctx->num_cus * KK_PANEL_DT * 2 - 1) / ((int64_t)ctx->num_cus * KK_PANEL_DT * 2));
    // the lag-1 kernel (three register-resident panels of two vectors, 448 double2 per grid-row): vectors of <= 8 such rows per
    // block, i.e. 1.83 M rows on 256 CUs; beyond that (and in the strict order) the kernel above
    const int nvl = (int)((ld_dec + (int64_t)ctx->num_cus * KK_LAG_DT * 2 - 1) / ((int64_t)ctx->num_cus * KK_LAG_DT * 2));
    // (two vectors per reduction or not at all: with ONE vector per reduction the lag-1 form is bound by the latency of the
    // reduction chain of wave 0 -- publish, sweep, publish -- at ~3.6 us per vector, slower than k_mgs_panel's 3.0 at 2 M rows,
    // profiles/r05_panel_lag_ab.jsonl; three panels of two vectors fit the registers up to 8 grid-rows = 1.83 M rows on 256 CUs)
    const bool lag = !strict && ctx->panel_lag && nvl <= 8 && (ctx->panel_width == 0 || ctx->panel_width >= 2);
    const int P = lag ? 2 : kk_mgs_panel_width(ctx, ld, strict);
    KK_HIP(hipSetDevice(ctx->device));
    char* sync = (char*)ctx->d_sync;
    int* err = (int*)((char*)ctx->d_sync + KK_SYNC_ERR_OFFSET);
    const unsigned need = (unsigned)((m * nsweeps + P - 1) / P) + 3u;   // one epoch per panel + the norm
    if (ctx->persist_epoch > 0xffffffffu - need - 1u) {
        KK_HIP(hipMemsetAsync(sync, 0, (size_t)KK_SYNC_ERR_OFFSET, ctx->stream));
        ctx->persist_epoch = 0;
    }
    unsigned ebase = ctx->persist_epoch;
    ctx->persist_epoch += need;
    int fault = 0;
    if (ctx->persist_fault > 0) { --ctx->persist_fault; fault = 1; }
    int normalize = (normalize_w && nrm_out3) ? 1 : 0;
    ctx->persist_token += 1.0;
    double token = ctx->persist_token;
    double* ok_out = ctx->ws + WS_SCAL + SC_PERSIST_OK;
    kk_xs_dev xs = kk_xs_launch_args(ctx, (unsigned)((m * nsweeps + P - 1) / P) + (nrm_out3 ? 1u : 0u));   // cross-rank reductions of this launch (row-sharded context)
    long long timeout_ticks = kk_persist_timeout_ticks(ctx, ld, m * nsweeps, xs.world > 0);
    panel_apply_args ap = panel_apply_args();
    const bool do_apply = apply && apply->on && !lag && !carry_q && nv <= KK_PANEL_NVMID;
    KK_CHECK(!(apply && apply->on) || do_apply, KK_ERR_UNSUPPORTED, "kk_launch_mgs_panel: the requested in-kernel apply is not available for this launch (internal error)");
    if (do_apply) {
        ap.x = apply->x; ap.xs_dev = apply->f.xscale_dev; ap.nrows = apply->M->nrows;
        for (int q = 0; q < 9; ++q) ap.cst.c[q] = apply->M->dia_c[q];
        ap.cst.phase = apply->M->dia_phase; ap.cst.D = apply->M->dia_D;
        ++ctx->panel_apply_launches;
    }
    void* args[] = {(void*)&V, (void*)&ld, (void*)&m, (void*)&nsweeps, (void*)&w, (void*)&carry_q, (void*)&carry_s,
                    (void*)&out_s, (void*)&out_stride, (void*)&nrm_out3, (void*)&sync, (void*)&err, (void*)&fault,
                    (void*)&ebase, (void*)&normalize, (void*)&ok_out, (void*)&token, (void*)&xs, (void*)&timeout_ticks, (void*)&ap};
    kk_prof_scope ps(ctx, "k_mgs_panel");
    if (lag) {
        if (nvl <= 4) return launch_panel_lag_inst<4, 2>(ctx, args);
        if (nvl <= 6) return launch_panel_lag_inst<6, 2>(ctx, args);
        return launch_panel_lag_inst<8, 2>(ctx, args);
    }
    if (nv <= 4) return P >= 3 ? launch_panel_inst<4, 3>(ctx, args, do_apply) : (P == 2 ? launch_panel_inst<4, 2>(ctx, args, do_apply) : launch_panel_inst<4, 1>(ctx, args, do_apply));
    if (nv <= KK_PANEL_NVMID) return P >= 2 ? launch_panel_inst<KK_PANEL_NVMID, 2>(ctx, args, do_apply) : launch_panel_inst<KK_PANEL_NVMID, 1>(ctx, args, do_apply);
    if (nv <= 16) return launch_panel_inst<16, 1>(ctx, args, false);
    kk_set_error("kk_launch_mgs_panel: vector of %lld rows does not fit two register-resident panels", (long long)ld);
    return KK_ERR_UNSUPPORTED;
}
