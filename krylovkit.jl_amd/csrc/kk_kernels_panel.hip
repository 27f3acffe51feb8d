// Persistent PANEL kernel for modified Gram-Schmidt sweeps on vectors whose streams are SHORT -- the lengths where one grid
// reduction per basis vector, not bytes, bounds the register-resident kernel of kk_kernels_persist.hip (config 3 of
// BASELINE.json: 2M rows; strong-scaling shards).  SURVEY.md 8(a) rows a6 / a8; reference order being restated:
// src/orthonormal.jl:414-452 (MGS, MGS2), src/factorizations/arnoldi.jl:239-245, lanczos.jl:325-338.
//
//   * one block of 512 threads per CU (cooperative launch): seven DATA waves hold the block's rows of the work vector w in
//     registers for the whole launch (NV double2 per thread, as in k_mgs_persist), wave 0 is the block's REDUCTION wave and
//     owns no rows (see the last item);
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
//   * the reduction itself: the block's P (P + 1) / 2 partials are published as 16-byte tagged granules (write-through
//     stores; value-major and packed, 4 KB per value) and swept with sc1 loads, summed in a fixed order -- all blocks obtain
//     the same bits.  Publishing and sweeping is the job of WAVE 0 alone, which issues no panel loads: memory returns are in
//     order per wave, so a sweep issued by a wave with a panel in flight completes only after that panel has landed, the
//     update that frees the register set for the panel after next waits for it, and every panel is then requested into an
//     EMPTY memory pipe -- latency + stream per panel instead of the stream alone (the first version of this kernel, all
//     eight waves loading and sweeping: 7.5 us per vector at 4M rows against 4.8 of stream).  With the sweep in a wave of
//     its own the data waves get the totals ~2.5 us after they published and have the next panel requested while the current
//     one is still arriving.  Workgroup barriers are raw s_barrier (+ lgkmcnt(0)): a __syncthreads() would make hipcc drain
//     the panel loads in flight (vmcnt(0)) before every barrier.
#include "kk_internal.h"
#include "kk_device.h"

#define KK_PANEL_TIMEOUT_TICKS 300000000ll   // 3 s of the 100 MHz wall clock
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define KK_PANEL_PT 512

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d2 pload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
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
#define KK_PANEL_DW 7                       // data waves per block (waves 1..7); wave 0 reduces
#define KK_PANEL_DT (KK_PANEL_DW * 64)      // data threads per block

// data waves: hand the NVAL per-thread partials to the reduction wave, come back with the totals (same bits in every thread
// of every block).  Returns false after a timeout anywhere on the chip.
template <int NVAL>
__device__ __forceinline__ bool panel_reduce_data(const double (&acc)[NVAL], double (&tot)[NVAL], double* smA, const double* smB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NVAL; ++v) {
        const double t = wave_sum(acc[v]);
        if (lane == 0) smA[wave * 8 + v] = t;
    }
    lds_barrier();   // (1)
    lds_barrier();   // (2)
#pragma unroll
    for (int v = 0; v < NVAL; ++v) tot[v] = smB[v];
    return smB[8] == 0.0;
}

// reduction wave: one grid reduction of nval values.  Granule (v, block) of a set sits at ((v * G + block) * 16) bytes.
__device__ __forceinline__ bool panel_reduce_sync(int nval, unsigned epoch, int set, char* __restrict__ sync, int* __restrict__ err, const double* smA,
                                                  double* smB) {
    const int G = gridDim.x;
    const int lane = threadIdx.x;
    lds_barrier();   // (1)
    const unsigned set_bytes = (unsigned)G * 16u * 8u;   // room for 8 values per set
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, 2 * (int)set_bytes, 0x00020000);
    const unsigned set_off = (unsigned)set * set_bytes;
    if (lane < nval) {   // lane v publishes the block's partial of value v
        double b = 0;
#pragma unroll
        for (int k = 1; k <= KK_PANEL_DW; ++k) b += smA[k * 8 + lane];   // fixed order
        const unsigned long long bits = (unsigned long long)__double_as_longlong(b);
        v4u t;
        t.x = epoch; t.y = (unsigned)(bits >> 32); t.z = (unsigned)bits; t.w = epoch;
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, set_off + ((unsigned)lane * (unsigned)G + blockIdx.x) * 16u, 0, 16 /* sc1 */);
    }
    const long long t0 = wall_clock64();
    int good = 1;
    for (int v = 0; v < nval && good; ++v) {   // value after value: by the time value 0 is complete the others usually are too
        const unsigned voff_v = set_off + (unsigned)v * (unsigned)G * 16u;
        double total = 0;
        for (;;) {
            // compiler barrier: the buffer-load builtin is a plain read to LLVM -- without it the granule loads are hoisted
            // out of the spin loop as loop invariants and the wave polls registers (found the hard way: every launch timed out)
            asm volatile("" ::: "memory");
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
            if (wall_clock64() - t0 > KK_PANEL_TIMEOUT_TICKS || errv) { good = 0; break; }
        }
        if (lane == 0) smB[v] = total;
    }
    if (lane == 0 && !good) { __hip_atomic_store(err, 1, RLX_AGENT); smB[8] = 1.0; }
    lds_barrier();   // (2)
    return good != 0;
}

// panel p = vectors (sweep-major sequence) s0 .. s0 + P - 1 of the nsteps = m * nsweeps vectors of the launch
template <int NV, int P>
__device__ __forceinline__ void panel_issue(d2 (&q)[P][NV], const double* __restrict__ V, int64_t ld, int m, int s0, int nsteps, unsigned voff,
                                            unsigned sbytes) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int s = s0 + i;
        const bool valid = s < nsteps;
        const __amdgpu_buffer_rsrc_t r = pcol_rsrc(V + (int64_t)((valid ? s : 0) % m) * ld, valid ? (int)(ld * 8) : 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) q[i][j] = pload(r, voff, (unsigned)j * sbytes);
    }
}

template <int NV, int P>
__device__ __forceinline__ bool panel_step(d2 (&wr)[NV], d2 (&cur)[P][NV], int s0, int nsteps, int m, double* smA, const double* smB,
                                           double* __restrict__ out_s, int out_stride) {
    constexpr int NVAL = P * (P + 1) / 2;
    double acc[NVAL], tot[NVAL];
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
    if (!panel_reduce_data<NVAL>(acc, tot, smA, smB)) return false;
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
    if (blockIdx.x == 0 && threadIdx.x == 64) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int sv = s0 + i;
            if (sv < nsteps) out_s[(sv / m) * out_stride + (sv % m)] = s[i];
        }
    }
    return true;
}

// nsweeps MGS sweeps of w against V[:, 0:m) (+ an optional pending axpy w -= *carry_s * carry_q in front, + the squared
// norm of the result, + the normalised commit): the interface of k_mgs_persist.
template <int NV, int P>
__global__ __launch_bounds__(KK_PANEL_PT) void k_mgs_panel(const double* __restrict__ V, int64_t ld, int m, int nsweeps, double* __restrict__ w,
                                                           const double* __restrict__ carry_q, const double* __restrict__ carry_s,
                                                           double* __restrict__ out_s, int out_stride, double* __restrict__ nrm_out3,
                                                           char* __restrict__ sync, int* __restrict__ err, int fault, unsigned ebase, int normalize,
                                                           double* __restrict__ ok_out, double token) {
    __shared__ double smA[64];
    __shared__ double smB[16];
    if (fault && blockIdx.x == 0) {   // test hook (option "persist_fault"): block 0 behaves like a block whose spin ran out
        if (threadIdx.x == 0) __hip_atomic_store(err, 1, RLX_AGENT);
        return;
    }
    const int nsteps = m * nsweeps;
    const int npanels = (nsteps + P - 1) / P;
    constexpr int NVAL = P * (P + 1) / 2;
    if (threadIdx.x < 64) {
        // ---------------- the reduction wave: no rows, no panel loads -- nothing queues in front of its sweeps
        if (threadIdx.x == 0) smB[8] = 0.0;
        lds_barrier();   // (0)
        for (int p = 0; p < npanels; ++p)
            if (!panel_reduce_sync(NVAL, ebase + (unsigned)p + 1u, p & 1, sync, err, smA, smB)) return;
        if (nrm_out3) panel_reduce_sync(1, ebase + (unsigned)npanels + 1u, npanels & 1, sync, err, smA, smB);
        return;
    }
    // ---------------- the data waves
    const unsigned dt = threadIdx.x - 64;
    const unsigned sbytes = gridDim.x * KK_PANEL_DT * 16u;               // one grid-row in bytes
    const unsigned voff = (blockIdx.x * KK_PANEL_DT + dt) * 16u;         // this lane's byte offset inside a grid-row
    const __amdgpu_buffer_rsrc_t rw = pcol_rsrc(w, (int)(ld * 8));
    d2 wr[NV];
    d2 qa[P][NV], qb[P][NV];
    panel_issue<NV, P>(qa, V, ld, m, 0, nsteps, voff, sbytes);   // first panel on its way before anything else
#pragma unroll
    for (int j = 0; j < NV; ++j) wr[j] = pload(rw, voff, (unsigned)j * sbytes);
    if (carry_q) {   // pending axpy of the caller (Lanczos: w -= alpha0 v): one extra read of that vector, through the idle second panel set
        const __amdgpu_buffer_rsrc_t rc = pcol_rsrc(carry_q, (int)(ld * 8));
        const double cs = *carry_s;
#pragma unroll
        for (int j = 0; j < NV; ++j) qb[0][j] = pload(rc, voff, (unsigned)j * sbytes);   // all loads first: one round trip, not NV
#pragma unroll
        for (int j = 0; j < NV; ++j) fnma2(wr[j], cs, qb[0][j]);
    }
    lds_barrier();   // (0) (the timeout flag slot is initialised)
    for (int p = 0; p < npanels; p += 2) {
        panel_issue<NV, P>(qb, V, ld, m, (p + 1) * P, nsteps, voff, sbytes);   // next panel in flight across this panel's reduction
        if (!panel_step<NV, P>(wr, qa, p * P, nsteps, m, smA, smB, out_s, out_stride)) return;   // timeout: w in HBM is untouched
        if (p + 1 >= npanels) break;
        panel_issue<NV, P>(qa, V, ld, m, (p + 2) * P, nsteps, voff, sbytes);
        if (!panel_step<NV, P>(wr, qb, (p + 1) * P, nsteps, m, smA, smB, out_s, out_stride)) return;
    }
    double inv = 1.0;
    bool scale = false;
    if (nrm_out3) {
        double acc[1] = {0.0}, tot[1];
#pragma unroll
        for (int j = 0; j < NV; ++j) { acc[0] = fma(wr[j].x, wr[j].x, acc[0]); acc[0] = fma(wr[j].y, wr[j].y, acc[0]); }
        if (!panel_reduce_data<1>(acc, tot, smA, smB)) return;
        const double rt = sqrt(tot[0]);
        inv = 1.0 / rt;
        scale = normalize && rt > 0.0 && inv <= 1.79769313486231570815e308;
        if (blockIdx.x == 0 && threadIdx.x == 64) { nrm_out3[0] = tot[0]; nrm_out3[1] = rt; nrm_out3[2] = inv; }
    }
    // commit (see k_mgs_persist): every block writes its rows back or -- flag raised by a block that timed out -- none does
    if (__hip_atomic_load(err, RLX_AGENT)) return;
    if (blockIdx.x == 0 && threadIdx.x == 64) { ok_out[0] = token; ok_out[1] = scale ? 1.0 : inv; }   // SC_PERSIST_OK, SC_XS
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

// ---- launcher ------------------------------------------------------------------------------
// vectors of at most 16 grid-rows (3.67 M rows on 256 CUs): w plus two panels fit the 256 registers of a 512-thread block
int64_t kk_mgs_panel_capacity(kk_ctx ctx) { return (int64_t)ctx->num_cus * KK_PANEL_DT * 2 * 16; }
bool kk_mgs_panel_eligible(kk_ctx ctx, int64_t ld) {
    if (!ctx->mgs_panel || !ctx->mgs_persist || kk_sharded(ctx) || !ctx->d_sync) return false;
    if (ctx->num_cus > KK_SYNC_MAX_BLOCKS || ld * 8 >= ((int64_t)1 << 31)) return false;
    return ld <= kk_mgs_panel_capacity(ctx);
}

template <int NV, int P>
static int launch_panel_inst(kk_ctx ctx, void** args) {
    hipError_t e = hipLaunchCooperativeKernel((const void*)k_mgs_panel<NV, P>, dim3(ctx->num_cus), dim3(KK_PANEL_PT), args, 0, ctx->stream);
    if (e != hipSuccess) return kk_hip_fail(e, "hipLaunchCooperativeKernel(k_mgs_panel)", __FILE__, __LINE__);
    return KK_OK;
}

// panel width by vector length: what two register-resident panels + w leave room for (4 NV (1 + 2 P) <= ~200 registers)
int kk_mgs_panel_width(kk_ctx ctx, int64_t ld, bool strict) {
    if (strict) return 1;
    const int nv = (int)((ld + (int64_t)ctx->num_cus * KK_PANEL_DT * 2 - 1) / ((int64_t)ctx->num_cus * KK_PANEL_DT * 2));
    const int by_size = nv <= 4 ? 3 : (nv <= 9 ? 2 : 1);
    return ctx->panel_width > 0 ? std::min(ctx->panel_width, by_size) : by_size;
}

int kk_launch_mgs_panel(kk_ctx ctx, const double* V, int64_t ld, int m, int nsweeps, double* w, const double* carry_q,
                        const double* carry_s, double* out_s, int out_stride, double* nrm_out3, bool normalize_w, bool strict) {
    const int nv = (int)((ld + (int64_t)ctx->num_cus * KK_PANEL_DT * 2 - 1) / ((int64_t)ctx->num_cus * KK_PANEL_DT * 2));
    const int P = kk_mgs_panel_width(ctx, ld, strict);
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
    void* args[] = {(void*)&V, (void*)&ld, (void*)&m, (void*)&nsweeps, (void*)&w, (void*)&carry_q, (void*)&carry_s,
                    (void*)&out_s, (void*)&out_stride, (void*)&nrm_out3, (void*)&sync, (void*)&err, (void*)&fault,
                    (void*)&ebase, (void*)&normalize, (void*)&ok_out, (void*)&token};
    kk_prof_scope ps(ctx, "k_mgs_panel");
    if (nv <= 4) return P >= 3 ? launch_panel_inst<4, 3>(ctx, args) : (P == 2 ? launch_panel_inst<4, 2>(ctx, args) : launch_panel_inst<4, 1>(ctx, args));
    if (nv <= 9) return P >= 2 ? launch_panel_inst<9, 2>(ctx, args) : launch_panel_inst<9, 1>(ctx, args);
    if (nv <= 16) return launch_panel_inst<16, 1>(ctx, args);
    kk_set_error("kk_launch_mgs_panel: vector of %lld rows does not fit two register-resident panels", (long long)ld);
    return KK_ERR_UNSUPPORTED;
}
