// libkrylov_hip.so, C ABI part 5: the fused L3 steps -- initialize / expand! of Lanczos, Arnoldi and GKL with the
// speculative next-step apply (src/factorizations/{lanczos,arnoldi,gkl}.jl).
#include "kk_host.h"
#include <atomic>
#include <chrono>

// ------------------------------------------------------------------------------------------
// L3 fused expand! steps
// ------------------------------------------------------------------------------------------
int check_square_op(kk_op op, kk_basis b) {
    KK_CHECK(op && b, KK_ERR_INVALID, "null arg");
    KK_CHECK(op->ctx == b->ctx, KK_ERR_INVALID, "operator and basis belong to different contexts");
    const int64_t in = op->A.n_ghost > 0 ? op->A.n_local : op->ncols;
    KK_CHECK(op->nrows == b->n && in == b->n, KK_ERR_DIM, "operator is %lldx%lld but vectors have %lld rows",
             (long long)op->nrows, (long long)op->ncols, (long long)b->n);
    ctx_foreign_touch(b);   // (a step of ANOTHER factorization of this context: that one's run-ahead state in the shared scalars is void)
    return KK_OK;
}

// initialize (factorizations/lanczos.jl:180-222 == arnoldi.jl:135-175): col c0 = x0 -> v ; col c0+1 = r
static int krylov_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta, double* alpha, double* beta) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(c0 >= 0 && c0 + 2 <= b->cap, KK_ERR_INVALID, "initialize: need columns %d..%d", c0, c0 + 1);
    KK_TRY(norm_flush(b));
    KK_CHECK(alpha && beta, KK_ERR_INVALID, "null output");
    kk_ctx c = b->ctx;
    double* x0 = b->col(c0);
    double* r = b->col(c0 + 1);
    gram_touch(b, c0);
    // beta0 = norm(x0); Ax0 = A x0 with fused <x0, Ax0>
    KK_TRY(kk_launch_nrm2(c, x0, b->ld, SCP(c, SC_NRM2)));
    kk_spmv_fuse f;
    f.dot_mode = 1; f.dot_out = SCP(c, SC_ALPHA0);
    KK_TRY(kk_launch_spmv(c, op->A, x0, r, b->ld, f));
    KK_TRY(ws_fetch_async(c, WS_SCAL, 16, 0));
    KK_TRY(stream_sync(c));
    const double beta0 = pin(c, WS_SCAL + SC_NRM)[0];
    if (beta0 == 0.0) {
        kk_set_error("initial vector should not have norm zero");
        return KK_ERR_ZERO_NORM;
    }
    double a = pin(c, WS_SCAL + SC_ALPHA0)[0] / (beta0 * beta0);
    KK_TRY(kk_launch_scal(c, x0, b->ld, 1.0 / beta0, nullptr));   // v = x0/beta0      :190
    KK_TRY(kk_launch_scal(c, r, b->ld, 1.0 / beta0, nullptr));    // r = Ax0/beta0     :194
    const bool ir = (orth == KK_CGSIR || orth == KK_MGSIR);
    double beta_old = 0;
    if (ir) {
        KK_TRY(kk_launch_nrm2(c, r, b->ld, SCP(c, SC_NRM2B)));  // beta_old = norm(r) :196
        KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2B, 2, 1));
    }
    // r -= alpha v ; beta = norm(r)
    KK_TRY(kk_launch_axpby(c, r, x0, b->ld, -a, 1.0, nullptr, 1.0, 0));
    KK_TRY(kk_launch_nrm2(c, r, b->ld, SCP(c, SC_NRM2)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
    KK_TRY(stream_sync(c));
    double bt = pin(c, WS_SCAL + SC_NRM2)[1];
    if (ir) beta_old = pin(c, WS_SCAL + SC_NRM2B, 1)[1];
    auto correct = [&]() -> int {  // dalpha = <v,r>; alpha += dalpha; r -= dalpha v; beta = |r|   :201-204
        KK_TRY(kk_launch_mgs_step(c, r, b->ld, nullptr, nullptr, x0, WSP(c, WS_S), nullptr));
        KK_TRY(kk_launch_mgs_step(c, r, b->ld, x0, c->ws + WS_S, nullptr, nullptr, SCP(c, SC_NRM2)));
        KK_TRY(ws_fetch_async(c, WS_S, 1, 0));
        KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
        KK_TRY(stream_sync(c));
        a += pin(c, WS_S)[0];
        bt = pin(c, WS_SCAL + SC_NRM2)[1];
        return KK_OK;
    };
    if (orth == KK_CGS2 || orth == KK_MGS2) {
        KK_TRY(correct());
    } else if (ir) {
        while (KK_EPS < bt && bt < eta * beta_old) {
            beta_old = bt;
            KK_TRY(correct());
        }
    }
    *alpha = a;
    *beta = bt;
    return KK_OK;
}

KK_API int kk_lanczos_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta, double* alpha,
                                     double* beta) {
    return krylov_initialize(op, b, c0, orth, eta, alpha, beta);
}
KK_API int kk_arnoldi_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta, double* alpha,
                                     double* beta) {
    return krylov_initialize(op, b, c0, orth, eta, alpha, beta);
}

// Speculative first half of the NEXT expand!: w' = A (r/beta) - beta v  with the scale applied on
// the fly from the device-resident beta, so the GPU keeps working while the host reads back
// (alpha, beta), returns to the caller and re-enters.  r itself is NOT modified; the next expand
// call normalises it in place (after this read) and skips its SpMV if (op, c0, k, beta) match.
// Bit-identical to the non-speculative order: r*(1/beta) is formed with the same operands.
static int speculate_next(kk_op op, kk_basis b, int c0, int k_next, int dot_mode, bool with_prev, double beta_host, double* dot_out = nullptr,
                          bool inside_sweep = false /* the sweep launch that follows forms A v itself (c->sweep_apply): nothing is launched here */) {
    kk_ctx c = b->ctx;
    b->spec_valid = false;
    if (!c->speculate || c0 + k_next + 2 > b->cap || k_next + 1 > KK_MAX_M) return KK_OK;
    if (inside_sweep) {
        kk_spmv_fuse fi;
        fi.xscale_dev = c->persist_norm_done ? SCP(c, SC_XS) : SCP(c, SC_INVNRM);
        if (with_prev) { fi.vprev = b->col(c0 + k_next - 1); fi.bprev_dev = SCP(c, SC_NRM); }
        fi.dot_mode = dot_mode;
        fi.dot_out = dot_out ? dot_out : SCP(c, SC_SPECA);
        c->sweep_apply.on = true; c->sweep_apply.M = &op->A; c->sweep_apply.x = b->col(c0 + k_next); c->sweep_apply.f = fi;
        b->spec_dot_ptr = fi.dot_mode ? fi.dot_out : nullptr;
        b->spec_valid = true; b->spec_op = op; b->spec_c0 = c0; b->spec_k = k_next; b->spec_dot_mode = dot_mode;
        b->spec_beta = beta_host;
        c->spec_owner = b;
        b->spec_gen = c->foreign_gen;
        return KK_OK;
    }
    kk_spmv_fuse f;
    // a sweep that went through a normalising persistent launch left r / |r| in the column (or r itself when the norm was
    // zero): the kernel wrote the factor that is still to be applied -- 1 or 1/|r| -- to SC_XS
    f.xscale_dev = c->persist_norm_done ? SCP(c, SC_XS) : SCP(c, SC_INVNRM);
    if (with_prev) { f.vprev = b->col(c0 + k_next - 1); f.bprev_dev = SCP(c, SC_NRM); }
    f.dot_mode = dot_mode;
    // (where the speculative <v, w> lands: its own slot, or -- when the caller knows that every read-back of the current step
    // is in the stream already -- directly the slot the next call's sweep reads: no device-to-device copy at the next call)
    f.dot_out = dot_out ? dot_out : SCP(c, SC_SPECA);
    KK_TRY(kk_launch_spmv(c, op->A, b->col(c0 + k_next), b->col(c0 + k_next + 1), b->ld, f));
    b->spec_dot_ptr = f.dot_out;
    b->spec_valid = true; b->spec_op = op; b->spec_c0 = c0; b->spec_k = k_next; b->spec_dot_mode = dot_mode;
    b->spec_beta = beta_host;
    c->spec_owner = b;
    b->spec_gen = c->foreign_gen;
    return KK_OK;
}
// true if the previous expand on this basis already enqueued exactly this step's SpMV; moves the
// speculative alpha into the regular slot
static int spec_take(kk_op op, kk_basis b, int c0, int k, int dot_mode, double beta_old, double* a0_slot, bool* hit) {
    kk_ctx c = b->ctx;
    *hit = b->spec_valid && c->spec_owner == b && b->spec_gen == c->foreign_gen && b->spec_op == op && b->spec_c0 == c0 && b->spec_k == k &&
           b->spec_dot_mode == dot_mode && b->spec_beta == beta_old;
    if (*hit && dot_mode && b->spec_dot_ptr != a0_slot)
        KK_HIP(hipMemcpyAsync(a0_slot, b->spec_dot_ptr, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return KK_OK;
}
// the last synchronisation of an expand: a pending speculation request (Arnoldi) is enqueued first
// The host waits only for the read-backs queued so far (event), NOT for the speculative SpMV that
// is enqueued behind them -- that one keeps the GPU busy during the host round trip.
static int la_enqueue(kk_op op, kk_basis b, int c0, int j, int nsweeps, bool lanczos_carry);
// the run-ahead of an Arnoldi step (la_enqueue below) will go through the panel kernel, which can apply this operator itself: same conditions as
// la_enqueue's own + kk_sweep_apply_ok
static bool la_apply_inside(kk_op op, kk_basis b, int c0, int j, int nsweeps) {
    kk_ctx c = b->ctx;
    const int m = j + 1;
    if (!c->lookahead || !c->fold_scale || !c->speculate || !c->persist_norm_done || kk_sharded(c) || c->persist_skip > 0) return false;
    if (m > KK_MAX_M || c0 + j + 3 > b->cap || c0 + j + 2 > b->cap) return false;
    if (!kk_mgs_panel_eligible(c, b->ld) || kk_mgs_lowsync(c, b->ld, m)) return false;
    const int stride = nsweeps > 1 ? (int)(WS_G - WS_S) : KK_MAX_M;
    return stride >= m && kk_sweep_apply_ok(c, op->A, b->ld);
}
// ... and of a Lanczos step through k_mgs_persist, which can form w = A v - beta v_prev and alpha0 itself (kk_sweep_apply_ok_persist)
static bool la_apply_inside_lanczos(kk_op op, kk_basis b, int c0, int j) {
    kk_ctx c = b->ctx;
    const int m = j + 1;
    if (!c->lookahead || !c->fold_scale || !c->speculate || !c->persist_norm_done || kk_sharded(c) || c->persist_skip > 0) return false;
    if (m > KK_MAX_M || c0 + j + 3 > b->cap) return false;
    if (kk_mgs_lowsync(c, b->ld, m)) return false;
    return kk_sweep_apply_ok_persist(c, op->A, b->ld, m);
}
int fetch_mark(kk_ctx c) {
    KK_HIP(hipEventRecord(c->ev_fetch, c->stream));
    return KK_OK;
}
int fetch_wait(kk_ctx c) {
    KK_HIP(hipEventSynchronize(c->ev_fetch));
    return KK_OK;
}
int final_sync(kk_ctx c) {
    if (c->spec_req.active) {
        c->spec_req.active = false;
        KK_TRY(fetch_mark(c));
        const bool inside = c->spec_req.la_nsweeps > 0 && la_apply_inside(c->spec_req.op, c->spec_req.b, c->spec_req.c0, c->spec_req.k_next, c->spec_req.la_nsweeps);
        KK_TRY(speculate_next(c->spec_req.op, c->spec_req.b, c->spec_req.c0, c->spec_req.k_next, 0, false, 0.0, nullptr, inside));
        if (c->spec_req.la_nsweeps > 0)   // Arnoldi with a persistent MGS / MGS2 sweep: the whole next step behind its apply (see la_enqueue)
            KK_TRY(la_enqueue(c->spec_req.op, c->spec_req.b, c->spec_req.c0, c->spec_req.k_next, c->spec_req.la_nsweeps, false));
        if (c->sweep_apply.on) {   // (la_enqueue declined after all: the apply goes out as its own launch -- nobody is left to consume the request)
            c->sweep_apply.on = false;
            KK_TRY(speculate_next(c->spec_req.op, c->spec_req.b, c->spec_req.c0, c->spec_req.k_next, 0, false, 0.0));
        }
        return fetch_wait(c);
    }
    return stream_sync(c);
}

// Run-ahead of a whole step (Lanczos, MGS2 in the strict / panel order, one GPU): once the speculative apply of step j is in
// the stream, everything its sweep needs is on the device too -- the normalised v_j (the previous sweep's commit), alpha0
// (SC_SPECA), the basis -- so the sweep and its read-back are enqueued as well, BEFORE the host waits for the scalars of the
// step before.  The next call finds its results on the way (or there), enqueues the step after, and returns: the GPU never
// waits for the host round trip between two expand! calls (~45 us of a 970 us iteration at 10 M rows, ~40 of 250 at 2 M).
// Same kernels, same operands, same order as the call-by-call route: bit-identical results.  Not enqueued when the step after
// it would no longer fit the slab (c0 + j + 3 > capacity): with the usual capacity of krylovdim + 2 the run-ahead stops at the
// last step of a Krylov cycle instead of wasting a sweep the caller will never ask for.
static int la_enqueue(kk_op op, kk_basis b, int c0, int j, int nsweeps, bool lanczos_carry) {
    kk_ctx c = b->ctx;
    b->la_valid = false;
    const int m = j + 1;
    if (!c->lookahead || !c->fold_scale || !b->spec_valid || !c->persist_norm_done || (kk_sharded(c) && !kk_xs_on(c)) || c->persist_skip > 0) return KK_OK;
    if (m > KK_MAX_M || c0 + j + 3 > b->cap) return KK_OK;
    if (!(kk_mgs_panel_eligible(c, b->ld) || kk_mgs_persist_eligible(c, b->ld, m, nsweeps)) || kk_mgs_lowsync(c, b->ld, m)) return KK_OK;
    if (lanczos_carry && b->spec_dot_ptr != SCP(c, SC_ALPHA0)) return KK_OK;   // (alpha0 of step j must sit where the sweep reads it)
    const int slot = 2 + (j & 1);
    const int64_t offs[2] = {WS_S, WS_G};
    c->persist_norm_req = true;
    const bool was_done = c->persist_norm_done;
    // The CURRENT step's launch may still be waiting for its check (persist_pending, slot 0, its token): the nested sweep below
    // must not take its place -- the caller's persist_check has to look at THAT launch (with the bookkeeping
    // overwritten here a barrier timeout of the first launch of a run-ahead chain went unnoticed).  The launch enqueued ahead
    // is identified by (la_slot, la_token) and checked by the next call.
    const bool cur_pending = c->persist_pending;
    const int cur_slot = c->persist_slot;
    const double cur_token = c->persist_check_token;
    c->persist_pending = false;
    const int st = pass_mgs_strict_sweeps(c, b->col(c0), b->ld, m, nsweeps, b->col(c0 + j + 1), offs, true, slot,
                                          lanczos_carry ? b->col(c0 + j) : nullptr, lanczos_carry ? c->ws + WS_SCAL + SC_ALPHA0 : nullptr);
    const bool ahead_persistent = c->persist_pending;
    const double ahead_token = c->persist_check_token;
    c->persist_pending = cur_pending; c->persist_slot = cur_slot; c->persist_check_token = cur_token;
    c->persist_norm_req = false;
    c->persist_norm_done = was_done;   // (describes the sweep of the CURRENT step until the caller has read it)
    KK_TRY(st);
    if (!ahead_persistent) return KK_OK;   // (took the launch-per-vector route: results are simply not used ahead)
    KK_HIP(hipEventRecord(c->ev_la[slot & 1], c->stream));
    b->la_valid = true; b->la_k = j; b->la_slot = slot; b->la_token = ahead_token; b->la_nsweeps = nsweeps; b->la_kind = 0;
    b->la_inside = c->sweep_apply_fused;
    return KK_OK;
}

// The projection-based Lanczos step (CGS2, or MGS2 in its low-synchronisation form; lanczos.jl:318-322 / 329-336) for basis size
// m = k + 1, enqueued WITHOUT any host data: w = col(c0 + k + 1) holds A v - beta v_prev (the apply is in the stream), alpha0 sits in
// SC_ALPHA0, v = col(c0 + k) is normalised.  Coefficients, [Gram row,] scalars are fetched into pinned slot `slot`.
static int lanczos_proj_enqueue(kk_basis b, int c0, int k, kk_orth_t orth, int slot, int rows_in_stream, bool* rode) {
    kk_ctx c = b->ctx;
    const int m = k + 1;
    const int64_t ld = b->ld;
    double* V = b->col(c0);
    double* v = b->col(c0 + k);
    double* w = b->col(c0 + k + 1);
    const double* a0_dev = c->ws + WS_SCAL + SC_ALPHA0;
    *rode = false;
    if (orth == KK_CGS2) {
        // one projection pass with "w -= alpha0 v" folded in (read V twice in total):
        //   s = V'(w - alpha0 v) ; w <- w - V (s + alpha0 e_m) ; beta = |w|
        KK_TRY(kk_launch_project(c, V, ld, m, w, v, a0_dev, nullptr, WSP(c, WS_S), WSP(c, WS_G)));
        KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, c->ws + WS_S, -1.0, 1.0, m - 1, a0_dev, SCP(c, SC_NRM2)));
        // ONE read-back for coefficients and scalars: a D2H copy costs ~4.5 us on the stream whatever its size, and the
        // workspace areas in between travel along for free (WS_S .. WS_SCAL+4 = 10 KB)
        return ws_fetch_async(c, WS_S, WS_SCAL + 4 - WS_S, slot);
    }
    // low-sync MGS2: project (Gram row riding along) -> triangular solve ON THE DEVICE (alpha0 folded into the last
    // coefficient) -> update
    KK_TRY(lowsync_project_dev(b, m, w, v, a0_dev, a0_dev, WS_X, WS_Y, rode, rows_in_stream));
    KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, WSP(c, WS_X), -1.0, 1.0, -1, nullptr, SCP(c, SC_NRM2)));
    // ONE read-back (Gram row, last MGS coefficient, scalars): WS_G .. WS_SCAL+4 = 8 KB
    if (*rode) return ws_fetch_async(c, WS_G, WS_SCAL + 4 - WS_G, slot);
    return ws_fetch_async(c, WS_Y + m - 1, WS_SCAL + 4 - (WS_Y + m - 1), slot);
}
// Run-ahead of that step (VERDICT r4 item 5 for the route the SHORT vectors take: below ~1.4 M rows the auto mode runs the projection
// pair, and an expand! there was bound by the host round trip between two calls): once the speculative apply of step j is in the
// stream, the scale of its new basis vector (by the device's 1 / beta, guarded like the persistent kernels' commit), its projection
// step and its read-back are enqueued as well, BEFORE the host waits for the scalars of step j - 1.  Same kernels, operands and
// order as call by call: bit-identical.  The column the scale touched is noted as normalised (norm_col) once beta is known, so any
// other access settles it exactly as after a persistent sweep.
static int la_enqueue_proj(kk_op op, kk_basis b, int c0, int j, kk_orth_t orth) {
    kk_ctx c = b->ctx;
    b->la_valid = false;
    const int m = j + 1;
    if (!c->lookahead || !c->fold_scale || !b->spec_valid || kk_sharded(c)) return KK_OK;
    if (m > KK_MAX_M || c0 + j + 3 > b->cap || b->spec_dot_ptr != SCP(c, SC_ALPHA0)) return KK_OK;
    if (orth == KK_MGS2 && c0 != 0) return KK_OK;
    const int slot = 2 + (j & 1);
    // v_j = r / beta in place, beta still on the device only
    KK_TRY(kk_launch_scal(c, b->col(c0 + j), b->ld, 0.0, SCP(c, SC_INVNRM), 2));
    bool rode = false;
    KK_TRY(lanczos_proj_enqueue(b, c0, j, orth, slot, orth == KK_MGS2 ? j : 0, &rode));   // (Gram row j - 1 is being produced by the step in front)
    KK_HIP(hipEventRecord(c->ev_la[slot & 1], c->stream));
    b->la_valid = true; b->la_k = j; b->la_slot = slot; b->la_nsweeps = 0; b->la_kind = 1; b->la_orth = (int)orth; b->la_rode = rode;
    return KK_OK;
}

// ---- the whole step in one launch (short vectors: kk_kernels_fstep.hip).  Single-rank context, operator in the ELL format without
// ghost columns, factorization starting at column 0, at most KK_FS_MAX_M basis vectors, CGS2 or MGS2 in its low-synchronisation form.
static bool fstep_ok_common(kk_ctx c, kk_op op, kk_basis b, int c0, int k);
static bool fstep_ok(kk_ctx c, kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, bool lowsync) {
    if (!(orth == KK_CGS2 || (orth == KK_MGS2 && lowsync))) return false;
    return fstep_ok_common(c, op, b, c0, k);
}
// Arnoldi: the same launch without the three-term part, one (CGS, MGS) or two (CGS2, MGS2) orthogonalisation passes; MGS family in its low-sync form
static bool fstep_ok_arnoldi(kk_ctx c, kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, bool lowsync) {
    if (!(orth == KK_CGS || orth == KK_CGS2 || ((orth == KK_MGS || orth == KK_MGS2) && lowsync))) return false;
    return fstep_ok_common(c, op, b, c0, k);
}
static bool fstep_ok_common(kk_ctx c, kk_op op, kk_basis b, int c0, int k) {
    if (!c->fused_step || !c->d_fsync || kk_sharded(c) || c->allreduce || (c->comm && c->comm->active)) return false;
    if (c0 != 0 || k < 1 || k + 1 > KK_FS_MAX_M) return false;
    // The one launch wins while the basis is SHORT as well: its fixed cost is ~8-15 us against ~28 us of the projection pair's ten stream
    // operations, but every basis vector costs it 0.3-0.8 us (two wave reductions per column, a granule per value and block, a row of the
    // solve) against 0.05-0.4 us in the pair, whose passes use the whole chip.  Measured cross-over (tools/fstep_probe.py, krylovdim 30 and
    // 100, profiles/r06_fstep_probe.jsonl): m ~ 80-107 at 1 k rows, ~60 at 40 k, 27-35 at 102 k.  Beyond it the step takes the ordinary route
    // (same slab state: the Gram rows and the normalised column are kept by both).
    if (c->fused_step_m_limit >= 0) {
        const double lim = c->fused_step_m_limit > 0 ? (double)c->fused_step_m_limit : std::max(16.0, 96.0 - 64.0 * (double)b->n / 1.0e5);
        if ((double)(k + 1) > lim) return false;
    }
    const kk_sparse_dev& M = op->A;
    if (M.format != 0 || M.n_ghost != 0 || M.halo || M.plan) return false;
    return b->n <= c->fused_step_max_rows && b->n <= kk_fstep_capacity_rows(c) && b->ld * 8 < ((int64_t)1 << 31);
}
static inline double* fstep_slot(kk_ctx c, int slot) { return c->h_pin + (int64_t)slot * WS_TOTAL + WS_USER; }   // (WS_USER: unused on single-rank contexts)
// enqueue the step for basis size m = k + 1 (v = column k, normalised IN THE STREAM); the kernel reads beta of the step in front from
// the device (bprev_dev) or takes the host's value
static int fstep_enqueue(kk_op op, kk_basis b, int k, kk_orth_t orth, const double* bprev_dev, double bprev, int rows_in_stream, int slot, double* token_out,
                         bool arnoldi = false) {
    kk_ctx c = b->ctx;
    const int m = k + 1;
    const bool ls = orth == KK_MGS2 || orth == KK_MGS;
    const int npass = arnoldi && (orth == KK_CGS2 || orth == KK_MGS2) ? 2 : 1;
    if (ls) {   // Gram rows of the basis columns below the newest one (lowsync_project_dev): known, on their way in the stream, or recomputed (restart)
        const int newest = m - 1;
        if (std::max(b->gram_rows, rows_in_stream) < newest) KK_TRY(gram_ensure(b, newest));
        if (b->gram_rows < 1) b->gram_rows = 1;
        KK_TRY(gram_device(b));
    }
    c->fs_token += 1.0;
    *token_out = c->fs_token;
    return kk_launch_lanczos_fstep(c, op->A, b->col(0), b->ld, m, ls, orth == KK_CGS2, bprev_dev, bprev, ls ? b->d_gram : nullptr, b->cap, fstep_slot(c, slot),
                                   c->fs_token, c->fold_scale != 0, arnoldi, npass);
}
// wait for the token of a launch in its pinned slot (the kernel's LAST store): a spin on host memory -- no copy, no event
static bool fstep_wait(kk_ctx c, int slot, double token) {
    const volatile double* h = fstep_slot(c, slot);
    const auto t0 = std::chrono::steady_clock::now();
    for (long it = 0;; ++it) {
        if (h[0] == token) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
        if ((it & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) break;
    }
    (void)hipStreamSynchronize(c->stream);   // (a launch that is merely late has arrived now; one that gave up never will)
    std::atomic_thread_fence(std::memory_order_acquire);
    return h[0] == token;
}

KK_API int kk_lanczos_expand(kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, double eta, double beta_old,
                                 double* alpha, double* beta, int* npasses) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(k >= 1 && c0 >= 0 && c0 + k + 2 <= b->cap, KK_ERR_INVALID,
             "kk_lanczos_expand: need k >= 1 and columns %d..%d within capacity %d", c0, c0 + k + 1, b->cap);
    KK_CHECK(alpha && beta, KK_ERR_INVALID, "null output");
    KK_CHECK(beta_old != 0.0, KK_ERR_ZERO_NORM, "kk_lanczos_expand: residual norm is zero");
    kk_ctx c = b->ctx;
    KK_TRY(route_agree(b));
    const int64_t ld = b->ld;
    const int m = k + 1;                  // basis size after the push
    double* V = b->col(c0);
    double* v = b->col(c0 + k);           // holds r on entry
    const double* vprev = b->col(c0 + k - 1);
    double* w = b->col(c0 + k + 1);
    const bool cgs_order = (orth == KK_CGS || orth == KK_CGS2 || orth == KK_CGSIR);
    const bool wide = m > KK_MAX_M;   // more basis vectors than one panel: base recurrence + whole passes through orth_run (panel by panel)
    const bool lowsync = !wide && kk_mgs_lowsync(c, ld, m);
    // Row-sharded run, projection-based orthogonaliser: the alpha0 partial of the SpMV and the two projection panels
    // stay LOCAL, land side by side in ws[WS_SHBUF ..] = [alpha0 | V'w | V'v] and are summed by ONE all-reduce; the
    // second (and last) one is |w|^2.  SURVEY.md 8(e).
    const bool sh_fused = !wide && kk_sharded(c) && (orth == KK_CGS2 || (orth == KK_MGS2 && lowsync && c0 == 0));
    double* a0_slot = sh_fused ? WSP(c, WS_SHBUF) : SCP(c, SC_ALPHA0);
    // the previous expand! of this factorization may have left r already normalised (persistent kernel): then the column IS
    // the new basis vector; any other pending normalised column is settled first
    const bool v_ready = b->norm_col == c0 + k && b->norm_beta == beta_old;
    if (!v_ready) {
        norm_discard(b, c0 + k + 1);                    // (w is overwritten as a whole by the apply)
        KK_TRY(norm_flush_range(b, c0, k + 2));         // a normalised column among the ones this step reads; one beyond them (the dead
    }                                                   // residual of the factorization a restart shrank) stays as it is until somebody looks
    // the previous call may have enqueued this WHOLE step already (apply, sweep and read-back: la_enqueue below)
    const bool strict_branch = orth == KK_MGS2 && !wide && !sh_fused && !(lowsync && c0 == 0);
    const bool proj_branch = !wide && !sh_fused && (orth == KK_CGS2 || (orth == KK_MGS2 && lowsync && c0 == 0));
    const bool la_same = b->la_valid && b->spec_valid && c->spec_owner == b && b->spec_gen == c->foreign_gen && b->spec_op == op && b->spec_c0 == c0 && b->spec_k == k &&
                         b->la_k == k && b->spec_beta == beta_old && v_ready;
    const bool fs_route = proj_branch && fstep_ok(c, op, b, c0, k, orth, lowsync);   // the whole step in one launch (short vectors)
    const bool la_hit = la_same && b->la_kind == 0 && b->la_nsweeps == 1 && strict_branch;
    const bool la_proj_hit = la_same && b->la_kind == 1 && b->la_orth == (int)orth && proj_branch && !kk_sharded(c) && !fs_route;
    const bool la_fs_hit = la_same && b->la_kind == 2 && b->la_orth == (int)orth && fs_route;
    const int la_slot = b->la_slot;
    const double la_token = b->la_token;
    if (b->la_valid && !la_hit && !la_proj_hit && !la_fs_hit) b->spec_valid = false;   // a sweep enqueued ahead has consumed the speculative apply's output column: nothing to take over
    const int la_proj_slot = b->la_slot;
    const bool la_proj_rode = b->la_rode;
    bool hit = la_hit || la_proj_hit || la_fs_hit;
    if (!hit && !fs_route) KK_TRY(spec_take(op, b, c0, k, cgs_order ? 1 : 2, beta_old, a0_slot, &hit));
    gram_touch(b, c0 + k);
    int passes = 0;
    c->persist_norm_done = la_hit;   // (a step enqueued ahead always asked for the normalised commit)
    // V = push!(V, scale!!(r, 1/beta_old))   lanczos.jl:257
    if (v_ready) b->norm_col = -1;
    else KK_TRY(kk_launch_scal(c, v, ld, 1.0 / beta_old, nullptr));
    // (strict MGS2 on the register-resident kernel, value-free stencil: the sweep launch forms w and alpha0 itself -- k_mgs_persist<.., APPLY>;
    //  pass_mgs_strict_sweeps sends the apply out as its own launch on any route that cannot)
    const bool apply_inside = !hit && !fs_route && strict_branch && a0_slot == SCP(c, SC_ALPHA0) && c->persist_skip == 0 && kk_sweep_apply_ok_persist(c, op->A, ld, m);
    kk_spmv_fuse f_step;
    f_step.vprev = vprev; f_step.bprev = beta_old;
    f_step.dot_mode = cgs_order ? 1 : 2;
    f_step.dot_out = a0_slot;
    if (apply_inside) {
        c->sweep_apply.on = true; c->sweep_apply.M = &op->A; c->sweep_apply.x = v; c->sweep_apply.f = f_step;
    } else if (!hit && !fs_route) {
        // w = A v - beta_old v_prev with the fused alpha dot   lanczos.jl:297-299 / 306-308
        const bool prev_suspend = c->ar_suspend;
        if (sh_fused) c->ar_suspend = true;
        const int st = kk_launch_spmv(c, op->A, v, w, ld, f_step);
        c->ar_suspend = prev_suspend;
        KK_TRY(st);
    }  // else: the previous expand already enqueued exactly this SpMV (speculate_next)
    const double* a0_dev = c->ws + WS_SCAL + SC_ALPHA0;
    double a = 0, bt = 0;
    if (orth == KK_CGS || orth == KK_MGS || orth == KK_CGSIR || orth == KK_MGSIR || wide) {
        // w -= alpha v ; beta = |w|
        KK_TRY(kk_launch_mgs_step(c, w, ld, v, a0_dev, nullptr, nullptr, SCP(c, SC_NRM2)));
        KK_TRY(ws_fetch_async(c, WS_SCAL, 4, 0));
        KK_TRY(stream_sync(c));
        a = pin(c, WS_SCAL + SC_ALPHA0)[0];
        bt = pin(c, WS_SCAL + SC_NRM)[0];
        if (orth == KK_CGSIR || orth == KK_MGSIR) {  // lanczos.jl:346-354 / 363-374
            const double ab2 = a * a + beta_old * beta_old;
            double nold = std::sqrt(bt * bt + ab2);
            std::vector<double> s(m);
            while (KK_EPS < bt && bt < eta * nold) {
                nold = bt;
                double nn = 0;
                int p1 = 0;
                KK_TRY(orth_run(b, c0, m, w, orth == KK_CGSIR ? KK_CGS : KK_MGS, eta, s.data(), &nn, &p1, true));
                a += s[m - 1];
                bt = nn;
                ++passes;
            }
        } else if (orth == KK_CGS2 || orth == KK_MGS2) {   // wide: the one unconditional re-orthogonalisation pass of :320-322 / :331-336
            std::vector<double> s(m);
            double nn = 0;
            int p1 = 0;
            KK_TRY(orth_run(b, c0, m, w, orth == KK_CGS2 ? KK_CGS : KK_MGS, eta, s.data(), &nn, &p1, true));
            a += s[m - 1];
            bt = nn;
            passes = 1;
        }
    } else if (sh_fused) {
        double* buf = WSP(c, WS_SHBUF);
        const bool ls = (orth == KK_MGS2);
        if (ls) {  // low-sync MGS: the strictly-lower Gram rows of the basis, V'v being the newest one (rides along)
            if (b->gram_rows < m - 1) KK_TRY(gram_ensure(b, m - 1));  // only after the basis was transformed (restart)
            if (b->gram_rows < 1) b->gram_rows = 1;
            KK_TRY(gram_device(b));
        }
        {
            kk_ar_suspend local_only(c);
            KK_TRY(kk_launch_project(c, V, ld, m, w, nullptr, nullptr, v, buf + 1, buf + 1 + m));
        }
        KK_TRY(kk_allreduce(c, buf, 1 + 2 * m));                      // all-reduce 1 of 2: alpha0, V'w, V'v
        // rhs = V'w - alpha0 V'v = V'(w - alpha0 v) [exact low-sync triangular solve], alpha0 folded into the last coefficient
        KK_TRY(kk_launch_lanczos_coef(c, buf, ls ? b->d_gram : nullptr, b->cap, m, ls ? 1 : 0, WSP(c, WS_X), SCP(c, SC_TMP0)));
        KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, WSP(c, WS_X), -1.0, 1.0, -1, nullptr,
                                   SCP(c, SC_NRM2)));                  // all-reduce 2 of 2 inside: |w|^2
        // ONE read-back: the scalars and, right behind them in the workspace, the all-reduced Gram row V'v
        KK_TRY(ws_fetch_async(c, WS_SCAL, ls ? (WS_SHBUF + 1 + 2 * m) - WS_SCAL : 8, 0));
        KK_TRY(fetch_mark(c));
        {
            kk_ar_suspend local_only(c);   // the speculative alpha0 partial joins the next step's all-reduce
            KK_TRY(speculate_next(op, b, c0, k + 1, cgs_order ? 1 : 2, true, 0.0));
        }
        KK_TRY(fetch_wait(c));
        a = pin(c, WS_SCAL + SC_TMP0)[0] + pin(c, WS_SCAL + SC_TMP1)[0];
        bt = pin(c, WS_SCAL + SC_NRM)[0];
        if (ls) lowsync_commit_row(b, m, pin(c, WS_SHBUF + 1 + m, 0));
        passes = 1;
    } else if (fs_route) {
        // ONE launch per step (k_lanczos_fstep): apply, both grid reductions, the small solve, the update and the normalised commit; the scalars
        // arrive in a pinned slot the host polls.  The NEXT step's launch goes out before the host looks at this one's token.
        const int m1 = k + 1;
        int slot = 2 + (k & 1);
        double token = 0;
        if (la_fs_hit) { slot = la_slot; token = la_token; }
        else KK_TRY(fstep_enqueue(op, b, k, orth, nullptr, beta_old, 0, slot, &token));
        b->spec_valid = false; b->la_valid = false;
        if (c->lookahead && c->fold_scale && c0 + k + 3 <= b->cap && fstep_ok(c, op, b, c0, k + 1, orth, lowsync)) {
            double tk = 0;
            KK_TRY(fstep_enqueue(op, b, k + 1, orth, SCP(c, SC_NRM), 0.0, m1, 2 + ((k + 1) & 1), &tk));   // (beta of THIS step: on the device when that launch starts)
            b->spec_valid = true; b->spec_op = op; b->spec_c0 = c0; b->spec_k = k + 1; b->spec_dot_mode = -1; b->spec_beta = 0.0; b->spec_dot_ptr = nullptr;
            c->spec_owner = b; b->spec_gen = c->foreign_gen;
            b->la_valid = true; b->la_k = k + 1; b->la_slot = 2 + ((k + 1) & 1); b->la_token = tk; b->la_nsweeps = 0; b->la_kind = 2; b->la_orth = (int)orth; b->la_rode = false;
        }
        if (!fstep_wait(c, slot, token)) {
            // the launch gave up (a block that never became resident: GPU shared with another job).  Column k holds v = r / beta_old, nothing else of
            // the factorization was touched: the route is switched off for this context and the step runs again on the ordinary one
            KK_HIP(hipMemsetAsync((char*)c->d_fsync + KK_FS_SYNC_BYTES, 0, sizeof(int), c->stream));
            ++c->fstep_failures;
            c->fused_step = 0;
            b->spec_valid = false; b->la_valid = false;
            b->norm_col = c0 + k; b->norm_beta = beta_old;   // (v is normalised in place already: the repeated call must not scale it again)
            return kk_lanczos_expand(op, b, c0, k, orth, eta, beta_old, alpha, beta, npasses);
        }
        const double* h = fstep_slot(c, slot);
        a = h[1] + h[2];
        bt = h[4];
        if (orth == KK_MGS2) lowsync_commit_row(b, m1, h + 8);
        passes = 1;
        if (h[6] != 0.0 && kk_persist_norm_applies(bt)) { b->norm_col = c0 + k + 1; b->norm_beta = bt; }
        else { b->la_valid = false; b->spec_valid = false; }   // (zero / overflowing norm: the column holds w itself, the step enqueued behind it is void)
    } else if (orth == KK_CGS2 || (orth == KK_MGS2 && lowsync && c0 == 0)) {
        // projection-based step (lanczos_proj_enqueue): s = V'(w - alpha0 v) [exact triangular solve for MGS2], w <- w - V (s + alpha0 e_m),
        // beta = |w|; one host synchronisation.  With the run-ahead (la_enqueue_proj) this call finds its step in the stream already.
        const int dm = cgs_order ? 1 : 2;
        int slot = 0;
        bool rode = false;
        if (la_proj_hit) {
            slot = la_proj_slot;
            rode = la_proj_rode;
        } else {
            KK_TRY(lanczos_proj_enqueue(b, c0, k, orth, 0, 0, &rode));
            KK_TRY(fetch_mark(c));
        }
        // the NEXT step: its apply (|w| and 1 / |w| are on the device; alpha0 straight to its slot: every read-back of this step is in the
        // stream) and, where the run-ahead applies, its scale + projection step + read-back
        KK_TRY(speculate_next(op, b, c0, k + 1, dm, true, 0.0, kk_sharded(c) ? nullptr : SCP(c, SC_ALPHA0)));
        KK_TRY(la_enqueue_proj(op, b, c0, k + 1, orth));
        if (la_proj_hit) KK_HIP(hipEventSynchronize(c->ev_la[slot & 1]));
        else KK_TRY(fetch_wait(c));
        const int64_t off_last = orth == KK_CGS2 ? WS_S : WS_Y;
        a = pin(c, WS_SCAL + SC_ALPHA0, slot)[0] + pin(c, off_last, slot)[m - 1];
        if (rode) lowsync_commit_row(b, m, pin(c, WS_G, slot));
        bt = pin(c, WS_SCAL + SC_NRM, slot)[0];
        passes = 1;
        // the step enqueued ahead has scaled the new residual column in place: logically it still holds r = bt * column
        if (b->la_valid && b->la_kind == 1) {
            if (kk_persist_norm_applies(bt)) { b->norm_col = c0 + k + 1; b->norm_beta = bt; }
            else { b->la_valid = false; b->spec_valid = false; }   // (zero / overflowing norm: the kernel left the column alone; the step behind it is void)
        }
    } else if (orth == KK_MGS2) {
        // strict: w -= alpha0 v fused with the first dot of the sweep   lanczos.jl:329-334
        const int64_t offs[1] = {WS_S};
        int slot = 0;
        bool inside_lost = la_hit && b->la_inside;   // the launch in question (this call's, or the one enqueued ahead) formed w and alpha0 itself
        for (int attempt = 0;; ++attempt) {
            const bool ahead = la_hit && attempt == 0;   // this step's sweep and read-back are in the stream already
            if (!ahead) {
                c->persist_norm_req = c->fold_scale != 0;   // the kernel holds |w| before it writes w back: store r / beta (next step's scale)
                const int st_sw = pass_mgs_strict_sweeps(c, V, ld, m, 1, w, offs, true, 0, v, a0_dev);
                c->persist_norm_req = false;
                KK_TRY(st_sw);
                if (attempt == 0) inside_lost = c->sweep_apply_fused;   // (the launch was to form w and alpha0 itself: if it is lost, so are they)
                if (!c->persist_pending) KK_TRY(ws_fetch_async(c, WS_SCAL, 1, 0));   // (the persistent route fetched the scalars already)
                KK_TRY(fetch_mark(c));
                slot = 0;
            } else {
                slot = la_slot;
            }
            if (attempt == 0 && (!kk_sharded(c) || kk_xs_on(c))) {   // (row-sharded: only when the sweep reduces over the ranks inside its launch)
                const bool inside = la_apply_inside_lanczos(op, b, c0, k + 1);
                KK_TRY(speculate_next(op, b, c0, k + 1, 2, true, 0.0, SCP(c, SC_ALPHA0), inside));   // |w| and 1/|w| are on the device; alpha0 straight to its slot
                KK_TRY(la_enqueue(op, b, c0, k + 1, 1, true));                                   // ... and, where it pays, the whole next step behind it
                if (c->sweep_apply.on) {   // (la_enqueue declined after all: the apply as its own launch)
                    c->sweep_apply.on = false;
                    KK_TRY(speculate_next(op, b, c0, k + 1, 2, true, 0.0, SCP(c, SC_ALPHA0)));
                }
            }
            if (ahead) KK_HIP(hipEventSynchronize(c->ev_la[slot & 1]));
            else KK_TRY(fetch_wait(c));
            bool redo = false;
            if (ahead) KK_TRY(persist_check_at(c, slot, la_token, &redo));
            else KK_TRY(persist_check(c, &redo));
            if (!redo) break;
            // The grid barrier of the persistent kernel timed out (GPU shared with another job): no block wrote w back, so w
            // = A v - beta_old v_prev, the pending pair (v, alpha0) and the basis are exactly what the sweep started from.  v has
            // already been scaled and the SpMV is done -- neither is repeated; only the sweep runs again, on the
            // launch-per-vector route, and everything enqueued ahead (which consumed the norm of the failed launch) is dropped.
            b->spec_valid = false;
            b->la_valid = false;
            KK_CHECK(attempt == 0, KK_ERR_HIP, "kk_lanczos_expand: the launch-per-vector MGS route reported a grid-barrier timeout (internal error)");
            // (the step enqueued ahead has replaced alpha0 on the device by its own: put this step's back -- it came home with
            // the read-back of the failed launch -- once the stream has run dry)
            KK_TRY(stream_sync(c));
            if (inside_lost) {
                // ... unless the lost launch was to form them itself (k_mgs_persist<.., APPLY>): then neither w nor alpha0 exists -- the repeated sweep is
                // handed the apply again and, on the launch-per-vector route it takes now, sends it out as a launch of its own
                c->sweep_apply.on = true; c->sweep_apply.M = &op->A; c->sweep_apply.x = v; c->sweep_apply.f = f_step;
            } else {
                const double a0_host = pin(c, WS_SCAL + SC_ALPHA0, slot)[0];
                KK_HIP(hipMemcpyAsync(SCP(c, SC_ALPHA0), &a0_host, sizeof(double), hipMemcpyHostToDevice, c->stream));
                KK_TRY(stream_sync(c));
            }
        }
        a = pin(c, WS_SCAL + SC_ALPHA0, slot)[0] + pin(c, WS_S, slot)[m - 1];
        bt = pin(c, WS_SCAL + SC_NRM2, slot)[1];
        passes = 1;
        if (c->persist_norm_done && kk_persist_norm_applies(bt)) { b->norm_col = c0 + k + 1; b->norm_beta = bt; }
    } else {
        kk_set_error("unknown orthogonalizer %d", (int)orth);
        return KK_ERR_INVALID;
    }
    *alpha = a;
    *beta = bt;
    if (npasses) *npasses = passes;
    if (b->spec_valid) b->spec_beta = bt;  // the caller must come back with exactly this beta
    return KK_OK;
}

KK_API int kk_arnoldi_expand(kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, double eta, double beta_old,
                                 double* h, double* beta, int* npasses) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(k >= 1 && c0 >= 0 && c0 + k + 2 <= b->cap, KK_ERR_INVALID,
             "kk_arnoldi_expand: need k >= 1 and columns %d..%d within capacity %d", c0, c0 + k + 1, b->cap);
    KK_CHECK(h && beta, KK_ERR_INVALID, "null output");
    KK_CHECK(beta_old != 0.0, KK_ERR_ZERO_NORM, "kk_arnoldi_expand: residual norm is zero");
    kk_ctx c = b->ctx;
    KK_TRY(route_agree(b));
    const int m = k + 1;
    double* v = b->col(c0 + k);
    double* w = b->col(c0 + k + 1);
    const bool v_ready = b->norm_col == c0 + k && b->norm_beta == beta_old;   // see kk_lanczos_expand
    if (!v_ready) {
        norm_discard(b, c0 + k + 1);
        KK_TRY(norm_flush_range(b, c0, k + 2));
    }
    const int la_sweeps = orth == KK_MGS ? 1 : (orth == KK_MGS2 ? 2 : 0);
    const bool strict_route = la_sweeps > 0 && m <= KK_MAX_M && !kk_mgs_lowsync(c, b->ld, m) && (!kk_sharded(c) || kk_xs_on(c));
    // the previous call may have enqueued this WHOLE step already (apply, sweeps and read-back: la_enqueue)
    const bool la_hit = b->la_valid && b->spec_valid && c->spec_owner == b && b->spec_gen == c->foreign_gen && b->spec_op == op && b->spec_c0 == c0 && b->spec_k == k &&
                        b->spec_dot_mode == 0 && b->la_kind == 0 && b->la_k == k && b->la_nsweeps == la_sweeps && b->spec_beta == beta_old && strict_route && v_ready;
    const int la_slot = b->la_slot;
    const double la_token = b->la_token;
    // the whole step in one launch (short vectors and short bases: kk_kernels_fstep.hip; round 6)
    const bool fs_route = m <= KK_MAX_M && fstep_ok_arnoldi(c, op, b, c0, k, orth, la_sweeps > 0 && kk_mgs_lowsync(c, b->ld, m));
    const bool la_fs_hit = b->la_valid && b->spec_valid && c->spec_owner == b && b->spec_gen == c->foreign_gen && b->spec_op == op && b->spec_c0 == c0 && b->spec_k == k &&
                           b->la_kind == 2 && b->la_orth == 100 + (int)orth && b->la_k == k && b->spec_beta == beta_old && v_ready && fs_route;
    if (b->la_valid && !la_hit && !la_fs_hit) b->spec_valid = false;   // the sweep enqueued ahead has consumed the speculative apply's column
    bool hit = la_hit || la_fs_hit;
    if (!hit && !fs_route) KK_TRY(spec_take(op, b, c0, k, 0, beta_old, SCP(c, SC_ALPHA0), &hit));
    gram_touch(b, c0 + k);
    c->persist_norm_done = la_hit;
    if (v_ready) b->norm_col = -1;
    else KK_TRY(kk_launch_scal(c, v, b->ld, 1.0 / beta_old, nullptr));  // push!(V, scale(r, 1/beta))   arnoldi.jl:209
    if (fs_route) {
        // w = A v, one or two orthogonalisation passes, the norm and the normalised commit in ONE launch; column of H and beta through the pinned slot
        int slot = 2 + (k & 1);
        double token = 0;
        if (la_fs_hit) { slot = la_slot; token = la_token; }
        else KK_TRY(fstep_enqueue(op, b, k, orth, nullptr, 0.0, 0, slot, &token, true));
        b->spec_valid = false; b->la_valid = false;
        if (c->lookahead && c->fold_scale && c0 + k + 3 <= b->cap && fstep_ok_arnoldi(c, op, b, c0, k + 1, orth, la_sweeps > 0 && kk_mgs_lowsync(c, b->ld, m + 1))) {
            double tk = 0;
            KK_TRY(fstep_enqueue(op, b, k + 1, orth, nullptr, 0.0, m, 2 + ((k + 1) & 1), &tk, true));
            b->spec_valid = true; b->spec_op = op; b->spec_c0 = c0; b->spec_k = k + 1; b->spec_dot_mode = -1; b->spec_beta = 0.0; b->spec_dot_ptr = nullptr;
            c->spec_owner = b; b->spec_gen = c->foreign_gen;
            b->la_valid = true; b->la_k = k + 1; b->la_slot = 2 + ((k + 1) & 1); b->la_token = tk; b->la_nsweeps = 0; b->la_kind = 2; b->la_orth = 100 + (int)orth; b->la_rode = false;
        }
        if (!fstep_wait(c, slot, token)) {   // the launch gave up: route off for this context, the step again on the ordinary one (see kk_lanczos_expand)
            KK_HIP(hipMemsetAsync((char*)c->d_fsync + KK_FS_SYNC_BYTES, 0, sizeof(int), c->stream));
            ++c->fstep_failures;
            c->fused_step = 0;
            b->spec_valid = false; b->la_valid = false;
            b->norm_col = c0 + k; b->norm_beta = beta_old;
            return kk_arnoldi_expand(op, b, c0, k, orth, eta, beta_old, h, beta, npasses);
        }
        const double* hs = fstep_slot(c, slot);
        for (int j = 0; j < m; ++j) h[j] = hs[8 + KK_FS_MAX_M + j];
        *beta = hs[4];
        if (orth == KK_MGS || orth == KK_MGS2) lowsync_commit_row(b, m, hs + 8);
        if (npasses) *npasses = (orth == KK_CGS2 || orth == KK_MGS2) ? 2 : 1;
        if (hs[6] != 0.0 && kk_persist_norm_applies(*beta)) { b->norm_col = c0 + k + 1; b->norm_beta = *beta; }
        else { b->la_valid = false; b->spec_valid = false; }
        if (b->spec_valid) b->spec_beta = *beta;
        return KK_OK;
    }
    if (!hit) {
        kk_spmv_fuse f;
        KK_TRY(kk_launch_spmv(c, op->A, v, w, b->ld, f));           // w = apply(operator, last(V))  :242
    }
    if (la_hit) {
        // this step is in the stream (or done): enqueue the NEXT one behind it, then collect
        KK_TRY(speculate_next(op, b, c0, k + 1, 0, false, 0.0, nullptr, la_apply_inside(op, b, c0, k + 1, la_sweeps)));
        KK_TRY(la_enqueue(op, b, c0, k + 1, la_sweeps, false));
        if (c->sweep_apply.on) {   // (la_enqueue declined after all: the apply as its own launch)
            c->sweep_apply.on = false;
            KK_TRY(speculate_next(op, b, c0, k + 1, 0, false, 0.0));
        }
        KK_HIP(hipEventSynchronize(c->ev_la[la_slot & 1]));
        bool redo = false;
        KK_TRY(persist_check_at(c, la_slot, la_token, &redo));
        if (!redo) {
            for (int j = 0; j < m; ++j) h[j] = pin(c, WS_S, la_slot)[j] + (la_sweeps > 1 ? pin(c, WS_G, la_slot)[j] : 0.0);
            *beta = pin(c, WS_SCAL + SC_NRM2, la_slot)[1];
            if (npasses) *npasses = la_sweeps;
            if (b->spec_valid) b->spec_beta = *beta;
            if (kk_persist_norm_applies(*beta)) { b->norm_col = c0 + k + 1; b->norm_beta = *beta; }
            return KK_OK;
        }
        // grid-barrier timeout of the launch enqueued ahead: w = A v is untouched -- drop what was enqueued behind it and run
        // this step's sweeps on the ordinary route (which now takes the launch-per-vector kernels)
        b->spec_valid = false; b->la_valid = false;
        KK_TRY(stream_sync(c));
        c->persist_norm_done = false;
        if (b->la_inside) {   // ... unless the lost launch was to form A v itself (k_mgs_panel<.., APPLY>): then nothing has written w yet
            kk_spmv_fuse f;
            KK_TRY(kk_launch_spmv(c, op->A, v, w, b->ld, f));
        }
    }
    // ask orth_run to enqueue the NEXT step's SpMV right before its final host sync (non-IR variants)
    c->spec_req.active = (orth != KK_CGSIR && orth != KK_MGSIR);
    c->spec_req.op = op; c->spec_req.b = b; c->spec_req.c0 = c0; c->spec_req.k_next = k + 1;
    c->spec_req.la_nsweeps = strict_route ? la_sweeps : 0;
    c->persist_norm_req = c->fold_scale != 0 && (orth == KK_MGS || orth == KK_MGS2);   // one launch ends the step: it may store r / beta
    int st = orth_run(b, c0, m, w, orth, eta, h, beta, npasses, true);  // orthogonalize!! + norm      :243-244
    c->persist_norm_req = false;
    c->spec_req.active = false;
    c->spec_req.la_nsweeps = 0;
    if (st == KK_OK && b->spec_valid) b->spec_beta = *beta;
    if (st == KK_OK && c->persist_norm_done && kk_persist_norm_applies(*beta)) { b->norm_col = c0 + k + 1; b->norm_beta = *beta; }
    return st;
}

// ---- GKL ----------------------------------------------------------------------------------
static int check_gkl(kk_op op, kk_basis bu, kk_basis bv) {
    KK_CHECK(op && bu && bv, KK_ERR_INVALID, "null arg");
    KK_CHECK(op->ctx == bu->ctx && op->ctx == bv->ctx, KK_ERR_INVALID, "objects belong to different contexts");
    const int64_t ncols = op->gather ? op->gather->n_local : op->ncols;   // row-sharded map: this rank's shard of the short vectors
    KK_CHECK(op->nrows == bu->n && ncols == bv->n, KK_ERR_DIM,
             "GKL: operator is %lldx%lld, U vectors have %lld rows, V vectors %lld", (long long)op->nrows,
             (long long)ncols, (long long)bu->n, (long long)bv->n);
    KK_CHECK(op->gather || op->A.n_ghost == 0, KK_ERR_UNSUPPORTED,
             "GKL on a row-sharded map needs an operator made by kk_csr_create_sharded_rect");
    KK_TRY(norm_flush(bu));
    KK_TRY(norm_flush(bv));
    return KK_OK;
}
// v = A'u - beta_old vlast with |v|^2 -> SC_NRM2 triple ; r = A v - alpha u (alpha = *alpha_dev) with |r|^2 -> SC_NRM2B triple.
// Single GPU: fused into the SpMV epilogue.  Row-sharded map: the reduce-scatter / all-gather sits between the sparse
// product and the vector update, so the update is a one-column unproject with the fused norm (all-reduced in its finalize).
static int gkl_apply_adjoint(kk_op op, const kk_sparse_dev* At, kk_basis bv, const double* u, double* v, const double* vlast,
                             double beta_old) {
    kk_ctx c = op->ctx;
    if (!op->gather) {
        kk_spmv_fuse f1;
        if (vlast) { f1.vprev = vlast; f1.bprev = beta_old; }
        f1.nrm_out = SCP(c, SC_NRM2);
        return kk_launch_spmv(c, *At, u, v, bv->ld, f1);
    }
    KK_TRY(rect_apply(op, 1, u, v));
    if (!vlast) return kk_launch_nrm2(c, v, bv->ld, SCP(c, SC_NRM2));
    kk_coef one;
    memset(&one, 0, sizeof(one));
    one.v[0] = beta_old;
    return kk_launch_unproject(c, vlast, bv->ld, 1, v, v, &one, nullptr, -1.0, 1.0, -1, nullptr, SCP(c, SC_NRM2));
}
static int gkl_apply_normal(kk_op op, kk_basis bu, const double* v, double* r, const double* u, const double* alpha_dev) {
    kk_ctx c = op->ctx;
    if (!op->gather) {
        kk_spmv_fuse f2;
        if (u) { f2.vprev = u; f2.bprev_dev = alpha_dev; }
        f2.nrm_out = SCP(c, SC_NRM2B);
        return kk_launch_spmv(c, op->A, v, r, bu->ld, f2);
    }
    KK_TRY(rect_apply(op, 0, v, r));
    if (!u) return KK_OK;
    return kk_launch_unproject(c, u, bu->ld, 1, r, r, nullptr, alpha_dev, -1.0, 1.0, -1, nullptr, SCP(c, SC_NRM2B));
}

KK_API int kk_gkl_initialize(kk_op op, kk_basis bu, kk_basis bv, double* alpha, double* beta) {
    KK_TRY(check_gkl(op, bu, bv));
    KK_CHECK(bu->cap >= 2 && bv->cap >= 1, KK_ERR_INVALID, "GKL initialize: capacity too small");
    KK_CHECK(alpha && beta, KK_ERR_INVALID, "null output");
    kk_ctx c = bu->ctx;
    const kk_sparse_dev* At;
    KK_TRY(get_matrix(op, 1, &At));
    double* u0 = bu->col(0);
    double* v0 = bv->col(0);
    double* r = bu->col(1);
    gram_touch(bu, 0); gram_touch(bv, 0);
    // beta0 = |u0| ; v0 = A' u0 (with |v0|^2) ; Av0 = A v0 (with <u0, A v0> computed separately)
    KK_TRY(kk_launch_nrm2(c, u0, bu->ld, SCP(c, SC_NRM2B)));
    KK_TRY(gkl_apply_adjoint(op, At, bv, u0, v0, nullptr, 0.0));
    if (op->gather) KK_TRY(rect_apply(op, 0, v0, r));
    else {
        kk_spmv_fuse f2;
        KK_TRY(kk_launch_spmv(c, op->A, v0, r, bu->ld, f2));
    }
    KK_TRY(kk_launch_dot(c, u0, r, bu->ld, SCP(c, SC_DOT)));
    KK_TRY(ws_fetch_async(c, WS_SCAL, 16, 0));
    KK_TRY(stream_sync(c));
    const double beta0 = pin(c, WS_SCAL + SC_NRMB)[0];
    if (beta0 == 0.0) {
        kk_set_error("initial vector should not have norm zero");
        return KK_ERR_ZERO_NORM;
    }
    const double a = pin(c, WS_SCAL + SC_NRM)[0] / beta0;                 // alpha = |v0|/beta0   gkl.jl:189
    const double a2 = pin(c, WS_SCAL + SC_DOT)[0] / (beta0 * beta0);      // alpha^2 check        :191-192
    if (!(std::fabs(a2 - a * a) <= std::sqrt(KK_EPS) * std::max(std::fabs(a2), a * a))) {
        kk_set_error("operator and its adjoint are not compatible");
        return KK_ERR_INVALID;
    }
    KK_TRY(kk_launch_scal(c, u0, bu->ld, 1.0 / beta0, nullptr));          // u = u0/beta0
    KK_TRY(kk_launch_scal(c, v0, bv->ld, 1.0 / (a * beta0), nullptr));    // v = v0/(alpha beta0)
    // r = Av0/(alpha beta0) - alpha u
    KK_TRY(kk_launch_axpby(c, r, u0, bu->ld, -a, 1.0 / (a * beta0), nullptr, 1.0, 0));
    KK_TRY(kk_launch_nrm2(c, r, bu->ld, SCP(c, SC_NRM2)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
    KK_TRY(stream_sync(c));
    *alpha = a;
    *beta = pin(c, WS_SCAL + SC_NRM)[0];
    return KK_OK;
}

KK_API int kk_gkl_expand(kk_op op, kk_basis bu, kk_basis bv, int k, kk_orth_t orth, double eta, double beta_old,
                             double* alpha, double* beta, int* npasses_v, int* npasses_u) {
    KK_TRY(check_gkl(op, bu, bv));
    KK_CHECK(k >= 1 && k + 2 <= bu->cap && k + 1 <= bv->cap, KK_ERR_INVALID,
             "kk_gkl_expand: k=%d does not fit capacities %d / %d", k, bu->cap, bv->cap);
    KK_CHECK(alpha && beta, KK_ERR_INVALID, "null output");
    KK_CHECK(beta_old != 0.0, KK_ERR_ZERO_NORM, "kk_gkl_expand: residual norm is zero");
    kk_ctx c = bu->ctx;
    const kk_sparse_dev* At;
    KK_TRY(get_matrix(op, 1, &At));
    double* u = bu->col(k);            // holds r on entry
    double* v = bv->col(k);
    double* r = bu->col(k + 1);
    const double* vlast = bv->col(k - 1);
    gram_touch(bu, k); gram_touch(bv, k);
    int pv = 0, pu = 0;
    double a = 0, bt = 0;
    std::vector<double> tmp(k + 1);
    // U = push!(U, scale!!(r, 1/beta_old))   gkl.jl:254
    KK_TRY(kk_launch_scal(c, u, bu->ld, 1.0 / beta_old, nullptr));
    // v = A'u - beta_old V[end]  (fused), alpha = |v| fused when no sweep follows
    const bool v_sweep = (orth == KK_MGS2 || orth == KK_CGSIR || orth == KK_MGSIR);
    KK_TRY(gkl_apply_adjoint(op, At, bv, u, v, vlast, beta_old));
    if (orth == KK_MGS2) {  // gkl.jl:330-336
        double nn = 0;
        KK_TRY(orth_run(bv, 0, k, v, KK_MGS, eta, tmp.data(), &nn, nullptr, true));
        a = nn;
        pv = 1;
        // alpha and 1 / alpha are on the device already: every route of the sweep ends with the norm triple {|v|^2, |v|, 1 / |v|} in
        // SC_NRM2 -- the same sqrt and the same division the host would repeat (rounds 1-4 copied them back up and waited: one host
        // round trip of the four per expand!)
    } else if (orth == KK_CGSIR || orth == KK_MGSIR) {  // gkl.jl:353-360 / 380-389
        KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
        KK_TRY(stream_sync(c));
        a = pin(c, WS_SCAL + SC_NRM)[0];
        double nold = std::sqrt(a * a + beta_old * beta_old);
        while ((orth == KK_CGSIR || KK_EPS < a) && a < eta * nold) {
            nold = a;
            double nn = 0;
            KK_TRY(orth_run(bv, 0, k, v, orth == KK_CGSIR ? KK_CGS : KK_MGS, eta, tmp.data(), &nn, nullptr, true));
            a = nn;
            ++pv;
            if (a == 0.0) break;
        }
        double hv[3] = {a * a, a, 1.0 / a};
        KK_HIP(hipMemcpyAsync(c->ws + WS_SCAL + SC_NRM2, hv, sizeof(hv), hipMemcpyHostToDevice, c->stream));
        KK_TRY(stream_sync(c));
    }
    (void)v_sweep;
    const double* alpha_dev = c->ws + WS_SCAL + SC_NRM;
    const double* inva_dev = c->ws + WS_SCAL + SC_INVNRM;
    // v = scale!!(v, inv(alpha))
    KK_TRY(kk_launch_scal(c, v, bv->ld, 0.0, inva_dev));
    // r = A v - alpha u (fused), beta = |r| fused when no sweep follows
    KK_TRY(gkl_apply_normal(op, bu, v, r, u, alpha_dev));
    if (orth == KK_CGS || orth == KK_MGS) {
        KK_TRY(ws_fetch_async(c, WS_SCAL, 16, 0));
        KK_TRY(stream_sync(c));
        a = pin(c, WS_SCAL + SC_NRM)[0];
        bt = pin(c, WS_SCAL + SC_NRMB)[0];
    } else if (orth == KK_CGS2 || orth == KK_MGS2) {  // gkl.jl:319-321 / 341-344
        KK_TRY(ws_fetch_async(c, WS_SCAL, 16, 2));
        double nn = 0;
        KK_TRY(orth_run(bu, 0, k + 1, r, orth == KK_CGS2 ? KK_CGS : KK_MGS, eta, tmp.data(), &nn, nullptr, true));
        a = pin(c, WS_SCAL + SC_NRM, 2)[0];
        bt = nn;
        pu = 1;
    } else {  // IR: gkl.jl:364-370 / 394-401
        KK_TRY(ws_fetch_async(c, WS_SCAL, 16, 0));
        KK_TRY(stream_sync(c));
        a = pin(c, WS_SCAL + SC_NRM)[0];
        bt = pin(c, WS_SCAL + SC_NRMB)[0];
        double nold = std::sqrt(a * a + bt * bt);
        while (KK_EPS < bt && bt < eta * nold) {
            nold = bt;
            double nn = 0;
            KK_TRY(orth_run(bu, 0, k + 1, r, orth == KK_CGSIR ? KK_CGS : KK_MGS, eta, tmp.data(), &nn, nullptr, true));
            bt = nn;
            ++pu;
        }
    }
    *alpha = a;
    *beta = bt;
    if (npasses_v) *npasses_v = pv;
    if (npasses_u) *npasses_u = pu;
    return KK_OK;
}

