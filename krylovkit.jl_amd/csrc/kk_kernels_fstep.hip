// libkrylov_hip.so: the WHOLE Lanczos expand! of a short vector in ONE launch.
//
// Below ~2.5e5 rows an expand! is bound by the HOST: the projection route enqueues ten stream operations per step (apply, its
// finalize, scale, project, finalize, triangular solve, unproject, finalize, read-back copy, event) at ~3 us each -- 31-34 us per
// step against ~5 us of kernels (profiles/r05_small_n.jsonl).  A captured graph does not help on this runtime: replaying the same
// nine nodes costs 7.9 us of host time but 28 us on the device, the nodes are serialised with a barrier packet each
// (tools/graph_launch_cost.hip, profiles/r06_graph_launch_cost.json); ONE kernel + copy + event is 9.5 us.  So the step becomes one
// kernel, and its scalars come home through a store into pinned host memory that the host polls -- no copy, no event:
//   lanczosrecurrence (src/factorizations/lanczos.jl:297-336) for ClassicalGramSchmidt2 and for ModifiedGramSchmidt2 in the
//   library's low-synchronisation form (the algebra of the row-sharded step, kk_krylov.hip `sh_fused`):
//     w = A v - beta v_prev                         (ELL gather, same products in the same order as k_spmv_ell; v = column m - 1, normalised)
//     [alpha0 | p = V'w | g = V'v]                  ONE grid reduction of 2 m + 1 values
//     rhs = p - alpha0 g ; (I + L) s = rhs          (L = strictly-lower Gram rows of V, row m - 1 = g: exact low-sync solve; CGS2: s = rhs)
//     w -= V (s + alpha0 e_m) ; beta = |w|          second grid reduction (one value)
//     column m <- w / beta                          (normalised commit, as the persistent kernels store it; w itself when beta is 0 / overflows)
// Every block owns a contiguous run of rows, keeps its part of w and v in registers between the phases, and reads V twice (L2- /
// Infinity-Cache-resident at these sizes).  Grid reductions: every block publishes its partial of value t as a tagged 16-byte granule
// {epoch, hi, lo, epoch} (system-coherent store: the XCDs' L2s are not coherent with each other), thread t of EVERY block polls the G
// granules of value t and adds them in block order -- the data is the flag, all blocks obtain the same bits, the small solve runs
// redundantly in every block.  Spins are bounded by the wall clock; a launch that gives up raises the error flag and commits nothing
// the fall-back route (the ordinary projection step) does not overwrite.
// Not bitwise equal to the call-by-call route (the inner products are summed in another order): equal to rounding, 1e-10 against the
// oracle like every route (tests/test_gpu_fstep.py).
#include "kk_internal.h"
#include "kk_device.h"

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned fs_v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fs_publish(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned epoch, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    fs_v4u t;
    t.x = epoch; t.y = (unsigned)(bits >> 32); t.z = (unsigned)bits; t.w = epoch;
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, 0, 16 /* sc1 */);
}
// sum over the granules of blocks [b_lo, b_hi) of one value (at `off`, 16 bytes apart; sixteen loads in flight at a time), in block
// order; false after a timeout / a raised flag
__device__ __forceinline__ bool fs_collect(__amdgpu_buffer_rsrc_t rs, unsigned off, int b_lo, int b_hi, unsigned epoch, const int* __restrict__ err, long long t0,
                                           long long timeout_ticks, double& total) {
    for (;;) {
        asm volatile("" ::: "memory");   // the granule loads must be re-issued by every pass
        const int errv = __hip_atomic_load(err, RLX_AGENT);
        bool ok = true;
        double x = 0;
        for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
            fs_v4u t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)((b0 + i < b_hi ? b0 + i : b0) * 16), 0, 16 /* sc1 */);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (b0 + i < b_hi) {
                    ok = ok && t[i].x == epoch && t[i].w == epoch;
                    x += __longlong_as_double((long long)(((unsigned long long)t[i].y << 32) | t[i].z));
                }
            }
        }
        if (ok) { total = x; return true; }
        __builtin_amdgcn_s_sleep(1);
        if (errv || wall_clock64() - t0 > timeout_ticks) return false;
    }
}
__device__ __forceinline__ void fs_host_store(double* p, double x) {   // system-scope store into the pinned host slot
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// TPB threads per block (256; 1024 exists behind option "fstep_threads" and is slower), NP row pairs per thread, JB = 4 columns of V per batch of
// loads; LOWSYNC: MGS2 in its low-synchronisation form, else CGS2
template <int TPB, int NP, bool LOWSYNC>
__global__ __launch_bounds__(TPB) void k_lanczos_fstep(const int32_t* __restrict__ ecol, const double* __restrict__ eval, int64_t ell_ld, int width,
                                                            int64_t nrows, double* __restrict__ V, int64_t ld, int m, const double* __restrict__ bprev_dev,
                                                            double bprev, int cgs_order, double* __restrict__ L, int cap, char* __restrict__ sync,
                                                            int* __restrict__ err, unsigned epoch, long long timeout_ticks, double* __restrict__ ws_scal,
                                                            double* __restrict__ host_out, double token, int normalize, int fault, int arnoldi, int npass) {
    __shared__ double wsum[TPB / 64][2 * KK_FS_MAX_M + 2];   // per-wave partials of the 2 m + 1 values
    __shared__ double tot[2 * KK_FS_MAX_M + 2];                    // totals; later rhs / coefficients in tot[1 .. m]
    __shared__ double part4[(2 * KK_FS_MAX_M + 2) * 4];            // quarter sums of the grid reduction
    __shared__ double gsave[KK_FS_MAX_M];                          // the new Gram row g = V'v (first pass)
    __shared__ double hsum[KK_FS_MAX_M];                           // sum of the passes' coefficients (Arnoldi: the column of H)
    __shared__ double red[TPB / 64];
    __shared__ int bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const double* v = V + (int64_t)(m - 1) * ld;
    const double* vprev = V + (int64_t)(m >= 2 ? m - 2 : 0) * ld;
    double* wout = V + (int64_t)m * ld;
    const double bp = arnoldi ? 0.0 : (bprev_dev ? *bprev_dev : bprev);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sync, 0, KK_FS_SYNC_BYTES, 0x00020000);
    if (tid == 0) bad = 0;
    if (fault && blockIdx.x == 0) {   // test hook: block 0 behaves like a block whose wait ran out
        if (tid == 0) __hip_atomic_store(err, 1, RLX_AGENT);
        return;
    }
    // ---- phase 1: w = A v - beta v_prev on this block's rows (registers), alpha0 partial
    d2 wr[NP], vr[NP];
    double a0p = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int64_t row = (((int64_t)blockIdx.x * NP + i) * TPB + tid) * 2;
        wr[i] = d2{0.0, 0.0}; vr[i] = d2{0.0, 0.0};
        if (row < nrows) {   // ell_ld is even and >= nrows; pad entries have val 0, col 0
            double s0 = 0, s1 = 0;
            const int32_t* cp = ecol + row;
            const double* vp = eval + row;
            int k = 0;
            for (; k + 4 <= width; k += 4) {   // the slot order of k_spmv_ell: same bits of A v
                int2 cc[4]; d2 a[4]; double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { cc[u] = *reinterpret_cast<const int2*>(cp + (int64_t)(k + u) * ell_ld); a[u] = ld2(vp + (int64_t)(k + u) * ell_ld); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { xa[u] = v[cc[u].x]; xb[u] = v[cc[u].y]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s0 = fma(a[u].x, xa[u], s0); s1 = fma(a[u].y, xb[u], s1); }
            }
            for (; k < width; ++k) {
                const int2 cc = *reinterpret_cast<const int2*>(cp + (int64_t)k * ell_ld);
                const d2 a = ld2(vp + (int64_t)k * ell_ld);
                s0 = fma(a.x, v[cc.x], s0);
                s1 = fma(a.y, v[cc.y], s1);
            }
            const d2 xv = ld2(v + row);
            d2 out{s0, s1};
            if (!arnoldi) {
                const d2 pv = ld2(vprev + row);
                if (cgs_order) { a0p = fma(xv.x, out.x, a0p); a0p = fma(xv.y, out.y, a0p); }      // <v, A v>        lanczos.jl:298
                out.x = fma(-bp, pv.x, out.x); out.y = fma(-bp, pv.y, out.y);
            }
            if (row + 1 >= nrows) out.y = 0.0;   // odd nrows: keep the pad row zero
            if (!arnoldi && !cgs_order) { a0p = fma(xv.x, out.x, a0p); a0p = fma(xv.y, out.y, a0p); }     // <v, w>          lanczos.jl:308
            wr[i] = out; vr[i] = xv;
        }
    }
    {
        const double t = wave_sum(a0p);
        if (lane == 0) wsum[wave][0] = t;
    }
    // ---- the orthogonalisation passes (one: Lanczos, Arnoldi with CGS / MGS; two: Arnoldi with CGS2 / MGS2).  Per pass: p_j = <V_j, w> (first pass also
    // g_j = <V_j, v>, the new Gram row) on this block's rows, four columns at a time with the NEXT batch requested before the current one is reduced;
    // ONE grid reduction; the small solve, redundantly in every block; w -= V s.
    constexpr int JB = 4;   // (16 / NP columns per batch were tried: the short factorizations this kernel serves pay for the clamped loads -- 15.2 -> 20.0 us per step at 1 k rows)
    constexpr bool PIPE = NP <= 4;
    auto vload = [&](d2 (&q)[JB][NP], int j0) {
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            const int j = j0 + u < m ? j0 + u : m - 1;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int64_t row = (((int64_t)blockIdx.x * NP + i) * TPB + tid) * 2;
#if defined(KK_FS_EXP) && (KK_FS_EXP & 2)       // (timing experiment 2: without the loads of V)
                q[u][i] = d2{(double)j, (double)row};
#else
                q[u][i] = row < ld ? ld2(V + (int64_t)j * ld + row) : d2{0.0, 0.0};   // (rows nrows .. ld - 1 of every column are zero)
#endif
            }
        }
    };
    double* rhs = tot + 1;            // rhs[i] overwrites p[i]
    for (int i = tid; i < m; i += TPB) hsum[i] = 0.0;
    double a0 = 0.0, s_last = 0.0;
    for (int pass = 0; pass < npass; ++pass) {
        const bool first = pass == 0;
        const int nval = first ? 2 * m + 1 : m + 1;       // [alpha0 (Lanczos) | p | g (first pass)]
        auto vdots = [&](const d2 (&q)[JB][NP], int j0) {
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                if (j0 + u >= m) break;   // (uniform)
                double pp = 0, gg = 0;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    pp = fma(q[u][i].x, wr[i].x, pp); pp = fma(q[u][i].y, wr[i].y, pp);
                    gg = fma(q[u][i].x, vr[i].x, gg); gg = fma(q[u][i].y, vr[i].y, gg);
                }
#if !(defined(KK_FS_EXP) && (KK_FS_EXP & 1))   // (timing experiment 1: without the wave reductions -- tools/fstep_where.sh; never defined in the product build)
                pp = wave_sum(pp);
                if (first) gg = wave_sum(gg);
#endif
                if (lane == 0) { wsum[wave][1 + j0 + u] = pp; if (first) wsum[wave][1 + m + j0 + u] = gg; }
            }
        };
        {
            d2 qa[JB][NP], qb[PIPE ? JB : 1][PIPE ? NP : 1];
            vload(qa, 0);
            for (int j0 = 0; j0 < m; j0 += (PIPE ? 2 : 1) * JB) {
                if constexpr (PIPE) {
                    if (j0 + JB < m) vload(qb, j0 + JB);
                    vdots(qa, j0);
                    if (j0 + 2 * JB < m) vload(qa, j0 + 2 * JB);
                    if (j0 + JB < m) vdots(qb, j0 + JB);
                } else {
                    vdots(qa, j0);
                    if (j0 + JB < m) vload(qa, j0 + JB);
                }
            }
        }
        __syncthreads();
        // ---- grid reduction of the pass: granule of (value t, block b) at (t * G + b) * 16 of set pass & 1 (a fast block may publish the second pass
        // while a slow one still polls the first: different sets), tag epoch + pass
        const unsigned set_off = (unsigned)(pass & 1) * (unsigned)KK_FS_SET_BYTES;
        for (int t = tid; t < nval; t += TPB) {
            double sv = 0;
#pragma unroll
            for (int wv = 0; wv < TPB / 64; ++wv) sv += wsum[wv][t];   // fixed order
            fs_publish(rs, set_off + (unsigned)((t * G + (int)blockIdx.x) * 16), epoch + (unsigned)pass, sv);
        }
        // four threads per value, each the granules of a quarter of the blocks (<= 32: one or two batches of loads); the quarters are added in a fixed order
        const long long t0 = wall_clock64();
        const int gq = (G + 3) >> 2;
        for (int idx = tid; idx < nval * 4; idx += TPB) {
            const int t = idx >> 2, qd = idx & 3;
            const int b_lo = qd * gq, b_hi = b_lo + gq < G ? b_lo + gq : G;
            double x = 0;
#if defined(KK_FS_EXP) && (KK_FS_EXP & 4)       // (timing experiment 4: without waiting for the other blocks)
            x = 1e-3;
#else
            if (b_lo < b_hi && !fs_collect(rs, set_off + (unsigned)(t * G * 16), b_lo, b_hi, epoch + (unsigned)pass, err, t0, timeout_ticks, x)) bad = 1;
#endif
            part4[idx] = x;
        }
        __syncthreads();
        if (bad) { if (tid == 0) __hip_atomic_store(err, 1, RLX_AGENT); return; }
        for (int t = tid; t < nval; t += TPB) tot[t] = ((part4[4 * t] + part4[4 * t + 1]) + part4[4 * t + 2]) + part4[4 * t + 3];
#if defined(KK_FS_EXP) && (KK_FS_EXP & 8)       // (timing experiment 8: keep the numbers finite when 1 / 2 are on)
        for (int t = tid; t < nval; t += TPB) tot[t] = 1e-3;
#endif
        __syncthreads();
        // ---- coefficients (the algebra of k_lanczos_coef): rhs = p - alpha0 g ; low-sync: (I + L) s = rhs with row m - 1 of L = g
        if (first) {
            a0 = arnoldi ? 0.0 : tot[0];
            for (int i = tid; i < m; i += TPB) gsave[i] = tot[1 + m + i];     // the Gram row: kept for the second pass' solve and for the host
        }
        __syncthreads();
        const double* g = gsave;
        if (first && !arnoldi) for (int i = tid; i < m; i += TPB) rhs[i] = fma(-a0, g[i], rhs[i]);
        if (LOWSYNC && first && blockIdx.x == 0)
            for (int i = tid; i < m - 1; i += TPB) L[(int64_t)(m - 1) * cap + i] = g[i];   // the new Gram row (device mirror; the host's copy travels below)
        __syncthreads();
        if (LOWSYNC) {
            // column-oriented forward substitution; thread i owns row i (m <= KK_FS_MAX_M <= TPB)
            const int i = tid;
            const bool act = i < m;
            const double* lrow = (i == m - 1) ? g : L + (int64_t)(act ? i : 0) * cap;
            constexpr int AH = 8;     // entries of the row requested ahead of their use (L2 latency off the dependent chain)
            double la[AH];
#pragma unroll
            for (int u = 0; u < AH; ++u) la[u] = (act && u < i) ? lrow[u] : 0.0;
            for (int j0 = 0; j0 < m - 1; j0 += AH) {
#pragma unroll
                for (int u = 0; u < AH; ++u) {
                    const int j = j0 + u;
                    if (j < m - 1) {   // uniform
                        const double lij = la[u];
                        const int jn = j + AH;
                        la[u] = (act && jn < i) ? lrow[jn] : 0.0;
                        const double sj = rhs[j];
                        if (i > j && act) rhs[i] = fma(-lij, sj, rhs[i]);
                        __syncthreads();
                    }
                }
            }
        }
        for (int i = tid; i < m; i += TPB) hsum[i] += rhs[i];     // Arnoldi: h = s (+ s2)
        if (first) s_last = rhs[m - 1];
        __syncthreads();
        if (first && !arnoldi && tid == 0) rhs[m - 1] = s_last + a0;   // Lanczos: alpha0 folded into the last coefficient: w -= V (s + alpha0 e_m)
        __syncthreads();
        // ---- w -= V coef on this block's rows (column order 0 .. m - 1, pipelined like the dots)
        auto vaxpy = [&](const d2 (&q)[JB][NP], int j0) {
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const double cf = j0 + u < m ? rhs[j0 + u] : 0.0;
#pragma unroll
                for (int i = 0; i < NP; ++i) { wr[i].x = fma(-cf, q[u][i].x, wr[i].x); wr[i].y = fma(-cf, q[u][i].y, wr[i].y); }
            }
        };
        {
            d2 qa[JB][NP], qb[PIPE ? JB : 1][PIPE ? NP : 1];
            vload(qa, 0);
            for (int j0 = 0; j0 < m; j0 += (PIPE ? 2 : 1) * JB) {
                if constexpr (PIPE) {
                    if (j0 + JB < m) vload(qb, j0 + JB);
                    vaxpy(qa, j0);
                    if (j0 + 2 * JB < m) vload(qa, j0 + 2 * JB);
                    if (j0 + JB < m) vaxpy(qb, j0 + JB);
                } else {
                    vaxpy(qa, j0);
                    if (j0 + JB < m) vload(qa, j0 + JB);
                }
            }
        }
        __syncthreads();   // (rhs / tot are rewritten by the next pass' reduction)
    }
    double an = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) { an = fma(wr[i].x, wr[i].x, an); an = fma(wr[i].y, wr[i].y, an); }
    an = wave_sum(an);
    if (lane == 0) red[wave] = an;
    __syncthreads();
    // ---- grid reduction 2: |w|^2 (value slot nval_max of the area, its own epoch)
    const unsigned noff = (unsigned)((2 * KK_FS_MAX_M + 1) * KK_FS_MAX_BLOCKS * 16);   // (last value slot of set 0)
    const unsigned nepoch = epoch + (unsigned)npass;
    if (tid == 0) {
        double s = 0;
#pragma unroll
        for (int wv = 0; wv < TPB / 64; ++wv) s += red[wv];
        fs_publish(rs, noff + blockIdx.x * 16u, nepoch, s);
    }
    __syncthreads();   // (red[] read above by thread 0 before it is reused)
    if (wave == 0) {   // lane l: the granules of blocks l and l + 64 (G <= 128); the lanes' values are added by wave_sum: same order in every block
        const long long t1 = wall_clock64();
        double x = 0;
        bool fine = true;
        for (;;) {
            asm volatile("" ::: "memory");
            const int errv = __hip_atomic_load(err, RLX_AGENT);
            const fs_v4u ta = __builtin_amdgcn_raw_buffer_load_b128(rs, noff + (unsigned)((lane < G ? lane : 0) * 16), 0, 16 /* sc1 */);
            const fs_v4u tb = __builtin_amdgcn_raw_buffer_load_b128(rs, noff + (unsigned)((lane + 64 < G ? lane + 64 : 0) * 16), 0, 16 /* sc1 */);
            const bool ok = (lane >= G || (ta.x == nepoch && ta.w == nepoch)) && (lane + 64 >= G || (tb.x == nepoch && tb.w == nepoch));
            if (__all(ok)) {
                x = lane < G ? __longlong_as_double((long long)(((unsigned long long)ta.y << 32) | ta.z)) : 0.0;
                if (lane + 64 < G) x += __longlong_as_double((long long)(((unsigned long long)tb.y << 32) | tb.z));
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            if (errv || wall_clock64() - t1 > timeout_ticks) { fine = false; break; }
        }
        x = wave_sum(x);
        if (lane == 0) { red[0] = x; if (!fine) bad = 1; }
    }
    __syncthreads();
    if (bad) { if (tid == 0) __hip_atomic_store(err, 1, RLX_AGENT); return; }
    const double n2 = red[0];
    const double rt = sqrt(n2);
    const double inv = 1.0 / rt;
    const bool scale = normalize && rt > 0.0 && inv <= 1.79769313486231570815e308;
    const double f = scale ? inv : 1.0;
    // ---- commit: column m <- w (normalised), scalars to the device workspace and -- block 0 -- to the pinned host slot
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int64_t row = (((int64_t)blockIdx.x * NP + i) * TPB + tid) * 2;
        if (row < nrows) st2(wout + row, d2{wr[i].x * f, wr[i].y * f});
    }
    if (blockIdx.x == 0) {
        if (tid == 0) {
            ws_scal[SC_ALPHA0] = a0; ws_scal[SC_NRM2] = n2; ws_scal[SC_NRM] = rt; ws_scal[SC_INVNRM] = inv; ws_scal[SC_XS] = scale ? 1.0 : inv;
        }
        // host slot: [0] token (written LAST), [1] alpha0, [2] last coefficient before the fold, [3 .. 5] |w|^2, |w|, 1 / |w|, [6] stored normalised?,
        // [8 .. 8 + m) the Gram row g = V'v.  System-scope stores; the token follows a system-scope fence.
        for (int i = tid; i < m; i += TPB) { fs_host_store(host_out + 8 + i, gsave[i]); fs_host_store(host_out + 8 + KK_FS_MAX_M + i, hsum[i]); }
        if (tid == 0) {
            fs_host_store(host_out + 1, a0);
            fs_host_store(host_out + 2, s_last);
            fs_host_store(host_out + 3, n2);
            fs_host_store(host_out + 4, rt);
            fs_host_store(host_out + 5, inv);
            fs_host_store(host_out + 6, scale ? 1.0 : 0.0);
        }
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: every store above is visible to the host before the token
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out), (unsigned long long)__double_as_longlong(token), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- launcher --------------------------------------------------------------------------------------------------------
int64_t kk_fstep_capacity_rows(kk_ctx ctx) { return (int64_t)std::max(1, std::min(std::min(KK_FS_MAX_BLOCKS, ctx->fstep_blocks), ctx->num_cus)) * 8 * 256 * 2; }

template <int TPB, int NP>
static void fstep_launch(bool lowsync, int G, hipStream_t s, const kk_sparse_dev& M, double* V, int64_t ld, int m, const double* bprev_dev, double bprev,
                         int cgs_order, double* L, int cap, char* sync, int* err, unsigned epoch, long long ticks, double* ws_scal, double* host_out,
                         double token, int normalize, int fault, int arnoldi, int npass) {
    if (lowsync)
        hipLaunchKernelGGL((k_lanczos_fstep<TPB, NP, true>), dim3(G), dim3(TPB), 0, s, M.ell_col, M.ell_val, M.ell_ld, M.width, M.nrows, V, ld, m, bprev_dev, bprev,
                           cgs_order, L, cap, sync, err, epoch, ticks, ws_scal, host_out, token, normalize, fault, arnoldi, npass);
    else
        hipLaunchKernelGGL((k_lanczos_fstep<TPB, NP, false>), dim3(G), dim3(TPB), 0, s, M.ell_col, M.ell_val, M.ell_ld, M.width, M.nrows, V, ld, m, bprev_dev, bprev,
                           cgs_order, L, cap, sync, err, epoch, ticks, ws_scal, host_out, token, normalize, fault, arnoldi, npass);
}

// one fused Lanczos step on columns [0, m) of V (v = column m - 1, normalised), result in column m; scalars to ws + WS_SCAL and to host_out
int kk_launch_lanczos_fstep(kk_ctx ctx, const kk_sparse_dev& M, double* V, int64_t ld, int m, bool lowsync, bool cgs_order, const double* bprev_dev,
                            double bprev, double* L, int cap, double* host_out, double token, bool normalize, bool arnoldi, int npass) {
    KK_CHECK(ctx->d_fsync && m >= 2 && m <= KK_FS_MAX_M && npass >= 1 && npass <= 2 && M.format == 0 && M.nrows <= kk_fstep_capacity_rows(ctx), KK_ERR_UNSUPPORTED,
             "kk_launch_lanczos_fstep: not eligible (m = %d, %lld rows)", m, (long long)M.nrows);
    // as many blocks as the chip offers CUs (all resident at once: the grid reductions need every block), at most "fstep_blocks".  256-thread
    // blocks while one row pair per thread covers the vector (<= 512 x blocks rows), else 1024-thread blocks with 1 / 2 / 4 pairs per thread
    const int gmax = std::max(1, std::min(std::min(KK_FS_MAX_BLOCKS, ctx->fstep_blocks), ctx->num_cus));
    // (1024-thread blocks -- four waves per SIMD -- were measured SLOWER at every length, 34.7 vs 20.0 us at 1 k rows, 44.1 vs 26.2 at 1e5: sixteen
    //  waves per block make the block-level steps of the two reductions and the solve's barriers that much longer; kept behind option "fstep_threads")
    const int tpb = ctx->fstep_threads == 1024 ? 1024 : (ctx->fstep_threads == 512 ? 512 : 256);
    const int64_t chunks = (M.nrows + tpb * 2 - 1) / (tpb * 2);
    const int np_need = (int)((chunks + gmax - 1) / gmax);
    const int NP = np_need <= 1 ? 1 : (np_need <= 2 ? 2 : (np_need <= 4 ? 4 : 8));
    const int G = (int)((chunks + NP - 1) / NP);
    KK_CHECK(G >= 1 && G <= gmax && np_need <= (tpb == 1024 ? 4 : 8), KK_ERR_UNSUPPORTED, "kk_launch_lanczos_fstep: %lld rows do not fit %d blocks", (long long)M.nrows, gmax);
    KK_HIP(hipSetDevice(ctx->device));
    if (ctx->fs_epoch > 0xffffffffu - 8u) {   // tags are unique over the life of the context: re-zero the area before the 32-bit counter wraps
        KK_HIP(hipMemsetAsync(ctx->d_fsync, 0, KK_FS_SYNC_BYTES, ctx->stream));
        ctx->fs_epoch = 0;
    }
    const unsigned epoch = ctx->fs_epoch + 1u;   // tags epoch .. epoch + npass (passes, then the norm)
    ctx->fs_epoch += 4;
    int fault = 0;
    if (ctx->fstep_fault > 0) { --ctx->fstep_fault; fault = 1; }
    int* err = (int*)((char*)ctx->d_fsync + KK_FS_SYNC_BYTES);
    const long long ticks = (long long)((ctx->persist_timeout_ms > 0 ? ctx->persist_timeout_ms : 20.0) * 1e5);   // 100 MHz wall clock
    kk_prof_scope ps(ctx, "k_lanczos_fstep");
    double* ws_scal = ctx->ws + WS_SCAL;
    const int nrm = normalize ? 1 : 0, cg = cgs_order ? 1 : 0;
#define FS_ARGS lowsync, G, ctx->stream, M, V, ld, m, bprev_dev, bprev, cg, L, cap, (char*)ctx->d_fsync, err, epoch, ticks, ws_scal, host_out, token, nrm, fault, arnoldi ? 1 : 0, npass
    if (tpb == 256) {
        switch (NP) {
            case 1: fstep_launch<256, 1>(FS_ARGS); break;
            case 2: fstep_launch<256, 2>(FS_ARGS); break;
            case 4: fstep_launch<256, 4>(FS_ARGS); break;
            default: fstep_launch<256, 8>(FS_ARGS); break;
        }
    } else if (tpb == 512) {
        switch (NP) {
            case 1: fstep_launch<512, 1>(FS_ARGS); break;
            case 2: fstep_launch<512, 2>(FS_ARGS); break;
            case 4: fstep_launch<512, 4>(FS_ARGS); break;
            default: fstep_launch<512, 8>(FS_ARGS); break;
        }
    } else {
        switch (NP) {
            case 1: fstep_launch<1024, 1>(FS_ARGS); break;
            case 2: fstep_launch<1024, 2>(FS_ARGS); break;
            default: fstep_launch<1024, 4>(FS_ARGS); break;
        }
    }
#undef FS_ARGS
    KK_HIP(hipGetLastError());
    ++ctx->fstep_launches;
    return KK_OK;
}
