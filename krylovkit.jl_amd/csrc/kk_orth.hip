// libkrylov_hip.so, C ABI part 4: L2 basis operations (project / unproject / rank-1 update / basis transform / Givens /
// Householder), the low-synchronisation Gram bookkeeping and the six orthogonalisers (src/orthonormal.jl).
#include "kk_host.h"

// ------------------------------------------------------------------------------------------
// L2 basis operations
// ------------------------------------------------------------------------------------------

KK_API int kk_project(kk_basis b, int c0, int m, kk_basis bx, int cx, double alpha, double beta, double* y) {
    CHECK_RANGE(b, c0, m); CHECK_COL(bx, cx); CHECK_SAME(b, bx);
    KK_CHECK(y || m == 0, KK_ERR_INVALID, "null y");
    if (m == 0) return KK_OK;
    kk_ctx c = b->ctx;
    KK_TRY(kk_launch_project(c, b->col(c0), b->ld, m, bx->col(cx), nullptr, nullptr, nullptr, WSP(c, WS_S), WSP(c, WS_G)));
    KK_TRY(ws_fetch_async(c, WS_S, m, 0));
    KK_TRY(stream_sync(c));
    const double* s = pin(c, WS_S);
    for (int j = 0; j < m; ++j) y[j] = (beta == 0.0) ? alpha * s[j] : beta * y[j] + alpha * s[j];
    return KK_OK;
}

KK_API int kk_unproject(kk_basis by, int cy, kk_basis b, int c0, int m, const double* x, double alpha, double beta) {
    CHECK_RANGE(b, c0, m); CHECK_COL(by, cy); CHECK_SAME(b, by);
    KK_CHECK(x || m == 0, KK_ERR_INVALID, "null x");
    KK_CHECK(!(by == b && cy >= c0 && cy < c0 + m), KK_ERR_INVALID, "kk_unproject: y aliases a basis column");
    gram_touch(by, cy);
    kk_coef ch;
    memset(&ch, 0, sizeof(ch));
    for (int j = 0; j < m; ++j) ch.v[j] = x[j];
    return kk_launch_unproject(b->ctx, b->col(c0), b->ld, m, by->col(cy), by->col(cy), &ch, nullptr, alpha, beta, -1,
                               nullptr, nullptr);
}

KK_API int kk_rank1update(kk_basis b, int c0, int m, kk_basis by, int cy, const double* x, double alpha, double beta) {
    CHECK_BLOCK(b, c0, m); CHECK_COL(by, cy); CHECK_SAME(b, by);   // any m: column-local, so wider ranges simply go panel by panel
    KK_CHECK(x || m == 0, KK_ERR_INVALID, "null x");
    KK_CHECK(!(by == b && cy >= c0 && cy < c0 + m), KK_ERR_INVALID, "kk_rank1update: y aliases a basis column");
    if (m == 0) return KK_OK;
    gram_touch(b, c0);
    for (int j0 = 0; j0 < m; j0 += KK_MAX_M) {
        const int mm = std::min(KK_MAX_M, m - j0);
        kk_coef ch;
        memset(&ch, 0, sizeof(ch));
        for (int j = 0; j < mm; ++j) ch.v[j] = x[j0 + j];
        KK_TRY(kk_launch_rank1(b->ctx, b->col(c0 + j0), b->ld, mm, by->col(cy), &ch, alpha, beta));
    }
    return KK_OK;
}

// basistransform!(b, U) (orthonormal.jl:291-354).  One kernel panel (m <= KK_MAX_M) is transformed in place by the MFMA
// kernel.  A wider basis -- krylovdim > 256 is legal in the reference and the thick restarts of eigsolve / svdsolve hand
// over the WHOLE basis (eigsolve/lanczos.jl:109) -- takes the rarely used wide route: the product is accumulated panel by
// panel of input columns into a temporary n-column slab (multi-column update kernel, 16 outputs per launch) and copied back.
static int basistransform_wide(kk_basis b, int c0, int m, int n, const double* U, int ldu) {
    kk_ctx c = b->ctx;
    double* tmp = nullptr;
    KK_HIP(hipSetDevice(c->device));
    const size_t bytes = (size_t)n * b->ld * sizeof(double);
    hipError_t e = hipMalloc(&tmp, bytes);
    if (e != hipSuccess) {
        kk_set_error("kk_basistransform: m = %d > %d needs a temporary of %zu bytes: %s", m, KK_MAX_M, bytes, hipGetErrorString(e));
        return KK_ERR_NOMEM;
    }
    int st = KK_OK;
    for (int k0 = 0; k0 < m && st == KK_OK; k0 += KK_MAX_M) {
        const int mk = std::min(KK_MAX_M, m - k0);
        st = block_update_run(c, b->col(c0 + k0), b->ld, mk, tmp, b->ld, n, U + k0, ldu, 1.0, k0 == 0 ? 0.0 : 1.0, nullptr);
    }
    if (st == KK_OK && hipMemcpyAsync(b->col(c0), tmp, bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) st = KK_ERR_HIP;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    return st;
}

KK_API int kk_basistransform(kk_basis b, int c0, int m, int n, const double* U, int ldu) {
    CHECK_BLOCK(b, c0, m);
    KK_CHECK(U && n >= 0 && n <= m && ldu >= m, KK_ERR_DIM, "kk_basistransform: U must be m x n with n <= m, ldu >= m");
    if (n == 0 || m == 0) return KK_OK;
    kk_ctx c = b->ctx;
    gram_touch(b, c0);
    if (m > KK_MAX_M) return basistransform_wide(b, c0, m, n, U, ldu);
    // pack U (m x n, leading dimension m) into the pinned staging area, then into device scratch
    KK_TRY(stream_sync(c));
    double* hp = c->h_U;  // pinned KK_MAX_M x KK_MAX_M staging
    for (int j = 0; j < n; ++j) memcpy(hp + (size_t)j * m, U + (size_t)j * ldu, m * sizeof(double));
    double* dU = c->partials;  // reuse the partial-sum buffer as U scratch (>= 2 MiB)
    KK_HIP(hipMemcpyAsync(dU, hp, (size_t)m * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return kk_launch_basistransform(c, b->col(c0), b->ld, m, n, dU);
}

KK_API int kk_givens_rmul(kk_basis b, int i1, int i2, double cc, double s) {
    CHECK_COL(b, i1); CHECK_COL(b, i2);
    KK_CHECK(i1 != i2, KK_ERR_INVALID, "kk_givens_rmul: i1 == i2");
    gram_touch(b, std::min(i1, i2));
    return kk_launch_givens(b->ctx, b->col(i1), b->col(i2), b->ld, cc, s);
}

KK_API int kk_householder_rmul(kk_basis b, int c0, int m, const double* v, double beta) {
    CHECK_BLOCK(b, c0, m);
    KK_CHECK(v || m == 0, KK_ERR_INVALID, "null v");
    if (m == 0 || beta == 0.0) return KK_OK;  // iszero(beta) && return b  (reflector.jl:147)
    gram_touch(b, c0);
    if (m > KK_MAX_M) {   // the reflector does not fit the kernarg block: read it from device memory (same row-local kernel)
        kk_ctx c = b->ctx;
        KK_CHECK((size_t)m <= (size_t)(2 * KK_MAX_M + 8) * KK_MAX_BLOCKS, KK_ERR_UNSUPPORTED, "kk_householder_rmul: m too large");
        KK_TRY(stream_sync(c));   // the partial-sum scratch doubles as staging: nothing may still be reading it
        KK_HIP(hipMemcpyAsync(c->partials, v, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
        return kk_launch_householder_dev(c, b->col(c0), b->ld, m, c->partials, beta);
    }
    kk_coef ch;
    memset(&ch, 0, sizeof(ch));
    for (int j = 0; j < m; ++j) ch.v[j] = v[j];
    return kk_launch_householder(b->ctx, b->col(c0), b->ld, m, &ch, beta);
}

// ---- Gram rows for the low-synchronisation MGS -------------------------------------------
// gram(i, j) = <b_i, b_j>, j < i, stored at b->gram[i*cap + j]; rows [0, gram_rows) valid.
// device mirror of the host Gram rows (used by the on-device low-sync solve)
int gram_device(kk_basis b) {
    if (b->gram.empty()) b->gram.assign((size_t)b->cap * b->cap, 0.0);
    if (!b->d_gram) {
        KK_HIP(hipSetDevice(b->ctx->device));   // lazy allocation: must land on the context's device, not the thread's current one
        KK_HIP(hipMalloc(&b->d_gram, (size_t)b->cap * b->cap * sizeof(double)));
        KK_HIP(hipMemcpy(b->d_gram, b->gram.data(), (size_t)b->cap * b->cap * sizeof(double), hipMemcpyHostToDevice));
        KK_HIP(hipMalloc(&b->d_gdiag, (size_t)b->cap * sizeof(double)));
        KK_HIP(hipMemset(b->d_gdiag, 0, (size_t)b->cap * sizeof(double)));
    }
    return KK_OK;
}
static int gram_upload_rows(kk_basis b, int lo, int hi) {
    if (!b->d_gram || hi <= lo) return KK_OK;
    // pageable source: the runtime stages it before returning, so the host vector may change afterwards
    KK_HIP(hipMemcpyAsync(b->d_gram + (size_t)lo * b->cap, b->gram.data() + (size_t)lo * b->cap,
                          (size_t)(hi - lo) * b->cap * sizeof(double), hipMemcpyHostToDevice, b->ctx->stream));
    // rows recomputed from the slab (after a restart transformed it): their norms were not measured, take them as 1
    const int z0 = lo <= 1 ? 0 : lo;   // (row 0 has no strictly-lower entries and is never uploaded: its norm goes with row 1)
    if (b->d_gdiag) KK_HIP(hipMemsetAsync(b->d_gdiag + z0, 0, (size_t)(hi - z0) * sizeof(double), b->ctx->stream));
    return KK_OK;
}
static int gram_ensure_host(kk_basis b, int upto);
int gram_ensure(kk_basis b, int upto /* exclusive */) {
    const int lo = std::max(b->gram_rows, 1);
    KK_TRY(gram_ensure_host(b, upto));
    return gram_upload_rows(b, std::min(lo, upto), std::max(upto, lo));
}
static int gram_ensure_host(kk_basis b, int upto /* exclusive */) {
    kk_ctx c = b->ctx;
    if (b->gram.empty()) b->gram.assign((size_t)b->cap * b->cap, 0.0);
    if (b->gram_rows < 1) b->gram_rows = 1;  // row 0 has no strictly-lower entries
    if (upto - b->gram_rows >= 4) {
        // many rows missing (after a thick restart): one MFMA Gram panel sweep instead of one
        // projection per row -- V is read ~upto/16 times instead of ~upto/2 times
        const int lo = b->gram_rows;
        std::vector<double> M;
        for (int j0 = 0; j0 < upto - 1; j0 += 16) {
            const int q = std::min(16, upto - 1 - j0);
            const int i0 = std::max(lo, j0 + 1);
            if (i0 >= upto) continue;
            const int p = upto - i0;
            M.assign((size_t)p * q, 0.0);
            const int saved_mode = c->block_mode;
            c->block_mode = 1;
            int st = block_inner_run(c, b->col(i0), b->ld, p, b->col(j0), b->ld, q, b->ld, M.data(), p);
            c->block_mode = saved_mode;
            KK_TRY(st);
            for (int jj = 0; jj < q; ++jj)
                for (int ii = 0; ii < p; ++ii)
                    if (j0 + jj < i0 + ii) b->gram[(size_t)(i0 + ii) * b->cap + j0 + jj] = M[ii + (size_t)p * jj];
        }
        b->gram_rows = upto;
        return KK_OK;
    }
    for (int i = b->gram_rows; i < upto; ++i) {
        for (int j0 = 0; j0 < i; j0 += KK_MAX_M) {
            const int mm = std::min(KK_MAX_M, i - j0);
            KK_TRY(kk_launch_project(c, b->col(j0), b->ld, mm, b->col(i), nullptr, nullptr, nullptr, WSP(c, WS_G), WSP(c, WS_G)));
            KK_TRY(ws_fetch_async(c, WS_G, mm, 1));
            KK_TRY(stream_sync(c));
            memcpy(&b->gram[(size_t)i * b->cap + j0], pin(c, WS_G, 1), mm * sizeof(double));
        }
        b->gram_rows = i + 1;
    }
    return KK_OK;
}
// solve (I + L) s = p in place, L = strictly lower Gram block of columns [c0, c0+m)
static void gram_solve(kk_basis b, int c0, int m, double* p) {
    for (int i = 1; i < m; ++i) {
        const double* row = &b->gram[(size_t)(c0 + i) * b->cap + c0];
        double t = p[i];
        for (int j = 0; j < i; ++j) t -= row[j] * p[j];
        p[i] = t;
    }
}

// ONE read-back from workspace offset `lo` through the named scalars: a D2H copy costs ~4.5 us on the stream whatever its
// size (<= 10 KB here), so coefficient areas, Gram row and norms travel in a single copy instead of two to four
static int fetch_through_scalars(kk_ctx c, int64_t lo, int slot) { return ws_fetch_async(c, lo, WS_SCAL + 8 - lo, slot); }

// ---- one orthogonalisation pass; coefficient results land in pinned slot `slot` ------------
// CGS pass:  s = V'w ; w -= V s ; optional |w| (orthonormal.jl:378-384)
static int pass_cgs(kk_ctx c, const double* V, int64_t ld, int m, double* w, bool want_norm, int slot) {
    KK_TRY(kk_launch_project(c, V, ld, m, w, nullptr, nullptr, nullptr, WSP(c, WS_S), WSP(c, WS_G)));
    KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, c->ws + WS_S, -1.0, 1.0, -1, nullptr,
                               want_norm ? SCP(c, SC_NRM2) : nullptr));
    if (want_norm) KK_TRY(fetch_through_scalars(c, WS_S, slot));
    else KK_TRY(ws_fetch_async(c, WS_S, m, slot));
    return KK_OK;
}
// strict MGS sweep (orthonormal.jl:414-423): `carry` = pending axpy (q, &s) left over from a
// previous sweep whose last subtraction is fused into this sweep's first dot.
int pass_mgs_strict(kk_ctx c, const double* V, int64_t ld, int m, double* w, int64_t ws_s, bool want_norm,
                           int slot, const double* carry_q, const double* carry_s, bool leave_carry) {
    const double* qp = carry_q;
    const double* sp = carry_s;
    for (int j = 0; j < m; ++j) {
        const double* q = V + (int64_t)j * ld;
        KK_TRY(kk_launch_mgs_step(c, w, ld, qp, sp, q, WSP(c, ws_s + j), nullptr));
        qp = q;
        sp = c->ws + ws_s + j;
    }
    if (!leave_carry) {
        KK_TRY(kk_launch_mgs_step(c, w, ld, qp, sp, nullptr, nullptr, want_norm ? SCP(c, SC_NRM2) : nullptr));
        if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, slot));
    }
    KK_TRY(ws_fetch_async(c, ws_s, m, slot));
    return KK_OK;
}
// `nsweeps` (1 or 2) strict MGS sweeps; sweep i leaves its coefficients at ws[ws_s[i] ..] (fetched into pinned slot
// `slot`).  When the work vector fits the register file of the chip the whole thing is ONE launch of the persistent
// kernel (w resident on chip, every basis vector read once from HBM per sweep); otherwise one fused axpy+dot launch
// per basis vector.  The order of operations -- and therefore every bit of the result -- is the same on both routes
// up to the summation order inside an inner product.
int pass_mgs_strict_sweeps(kk_ctx c, const double* V, int64_t ld, int m, int nsweeps, double* w, const int64_t* ws_s,
                           bool want_norm, int slot, const double* carry_q, const double* carry_s) {
    c->persist_norm_done = false;
    const bool panel = m > 0 && kk_mgs_panel_eligible(c, ld);   // short vectors: P basis vectors per grid reduction (P = 1 in strict mode)
    // a pending "apply inside the sweep launch" request (run-ahead of an Arnoldi step, kk_krylov.hip): honoured by the panel kernel; on any other
    // route the apply goes out as the separate launch it would have been
    kk_sweep_apply ap = c->sweep_apply;
    c->sweep_apply.on = false;
    c->sweep_apply_fused = false;
    {
        const int stride0 = nsweeps > 1 ? (int)(ws_s[1] - ws_s[0]) : KK_MAX_M;
        const bool plain = !ap.f.vprev && ap.f.dot_mode == 0 && !ap.f.nrm_out && ap.f.a0 == 0.0 && ap.f.a1 == 1.0;
        const bool lanczos = ap.f.vprev && ap.f.dot_mode == 2 && !ap.f.nrm_out && ap.f.a0 == 0.0 && ap.f.a1 == 1.0 && carry_q == ap.x && ap.f.dot_out == carry_s && nsweeps == 1;
        const bool fused = ap.on && c->persist_skip == 0 && stride0 >= m &&
                           ((panel && plain && !carry_q && !c->panel_lag && kk_sweep_apply_ok(c, *ap.M, ld)) ||
                            (!panel && lanczos && kk_sweep_apply_ok_persist(c, *ap.M, ld, m)));
        if (ap.on && !fused) {
            KK_TRY(kk_launch_spmv(c, *ap.M, ap.x, w, ld, ap.f));
            ap.on = false;
        }
    }
    if (panel || (m > 0 && kk_mgs_persist_eligible(c, ld, m, nsweeps))) {
        // both sweeps' coefficient areas must be addressable as out_s + sweep * stride
        const int stride = nsweeps > 1 ? (int)(ws_s[1] - ws_s[0]) : KK_MAX_M;
        if (c->persist_skip > 0) {
            --c->persist_skip;   // recovering from a grid-barrier timeout: this sweep takes the launch-per-vector route (same order)
        } else if (stride >= m) {
            const bool normalize = c->persist_norm_req && want_norm;
            if (panel) {
                KK_TRY(kk_launch_mgs_panel(c, V, ld, m, nsweeps, w, carry_q, carry_s, WSP(c, ws_s[0]), stride,
                                           want_norm ? SCP(c, SC_NRM2) : nullptr, normalize, c->mgs_mode == 0, ap.on ? &ap : nullptr));
                c->sweep_apply_fused = ap.on;
            } else {
                KK_TRY(kk_launch_mgs_persist(c, V, ld, m, nsweeps, w, carry_q, carry_s, WSP(c, ws_s[0]), stride,
                                             want_norm ? SCP(c, SC_NRM2) : nullptr, normalize, ap.on ? &ap : nullptr));
                c->sweep_apply_fused = ap.on;
            }
            c->persist_pending = true;
            c->persist_slot = slot;
            c->persist_check_token = c->persist_token;
            c->persist_norm_done = normalize;
            // ONE read-back from the first coefficient area through the named scalars (alpha0, |w|^2, |w|, 1/|w|, ... and the
            // completion token of the launch): a D2H copy costs ~4.5 us on the stream whatever its size, and the areas in
            // between travel along for free (<= 10 KB)
            int64_t lo = ws_s[0];
            for (int i = 1; i < nsweeps; ++i) lo = std::min(lo, ws_s[i]);
            KK_TRY(ws_fetch_async(c, lo, WS_SCAL + 16 - lo, slot));
            return KK_OK;
        }
    }
    const double* cq = carry_q;
    const double* cs = carry_s;
    for (int i = 0; i < nsweeps; ++i) {
        const bool last = (i == nsweeps - 1);
        KK_TRY(pass_mgs_strict(c, V, ld, m, w, ws_s[i], last && want_norm, slot, cq, cs, !last));
        cq = V + (int64_t)(m - 1) * ld;
        cs = c->ws + ws_s[i] + m - 1;
    }
    return KK_OK;
}
// after the host synchronisation that follows a persistent launch: did its grid barrier time out?  Then no block wrote w
// back (HBM holds the input of the sweep, the pending axpy operands are untouched), the persistent route is suspended
// for the next few sweeps and *timed_out tells the caller to repeat the sweep -- pass_mgs_strict_sweeps now takes the
// launch-per-vector route.  Callers that enqueued dependent work behind the launch (a speculative next-step apply)
// must cancel it first: it consumed scalars the failed launch never wrote.
int persist_check_at(kk_ctx c, int slot, double token, bool* timed_out) {
    *timed_out = false;
    // a launch that committed wrote its token next to the scalars of the sweep (same read-back); anything else -- the flag
    // was raised by a block whose spin ran out, blocks left without writing w back -- leaves the previous launch's token
    if (pin(c, WS_SCAL + SC_PERSIST_OK, slot)[0] != token) {
        kk_xs_postmortem(c, "persistent launch did not commit");
        KK_HIP(hipMemsetAsync((char*)c->d_sync + KK_SYNC_ERR_OFFSET, 0, sizeof(int), c->stream));
        c->persist_norm_done = false;
        ++c->persist_timeouts;
        if (c->comm && c->comm->xs_active) c->comm->xs_clear_word = true;   // (the lost launch's id sits in this rank's abort word: kk_xs_launch_args clears it)
        // the route is NOT switched off for good -- a transient timeout (GPU shared with another job) used to move
        // an auto-mode context from the reference's strict order to the low-sync form for the rest of its life, silently
        // changing the rounding of every later sweep.  The sweep that failed is repeated on the launch-per-vector route
        // (same strict order); the persistent route is retried after `persist_backoff` further strict sweeps, the interval
        // doubling with every timeout in a row.
        c->persist_skip = c->persist_backoff;
        c->persist_backoff = std::min(c->persist_backoff * 2, 1 << 20);
        // ordinary launches ASSUME that all blocks become resident.  Where that keeps failing (CUs masked or held by
        // another queue for good) the retries go through the cooperative API from now on: the runtime then guarantees the
        // residency (at ~35 us per launch) -- on a single-rank context; across ranks no launch API can, the back-off stays.
        if (++c->persist_timeouts_row >= 3 && !c->persist_coop && !kk_xs_on(c)) c->persist_coop = 1;
        *timed_out = true;
    } else {
        c->persist_timeouts_row = 0;
        if (c->persist_backoff > 4) c->persist_backoff = 4;   // a clean launch: back to the short retry interval
    }
    return KK_OK;
}
int persist_check(kk_ctx c, bool* timed_out) {
    *timed_out = false;
    if (!c->persist_pending) return KK_OK;
    c->persist_pending = false;
    return persist_check_at(c, c->persist_slot, c->persist_check_token, timed_out);
}
// strict sweeps + the synchronisation that ends them, with the recovery above folded in (no speculation involved)
static int strict_sweeps_synced(kk_ctx c, const double* V, int64_t ld, int m, int nsweeps, double* w, const int64_t* ws_s,
                                bool want_norm, int slot, bool last_sync /* final_sync: a requested speculative apply goes out */) {
    const auto req = c->spec_req;
    for (int attempt = 0; attempt < 2; ++attempt) {
        KK_TRY(pass_mgs_strict_sweeps(c, V, ld, m, nsweeps, w, ws_s, want_norm, slot, nullptr, nullptr));
        KK_TRY(last_sync ? final_sync(c) : stream_sync(c));
        bool redo = false;
        KK_TRY(persist_check(c, &redo));
        if (!redo) return KK_OK;
        if (last_sync && req.active) {   // the speculative apply final_sync enqueued read a norm the failed launch never wrote
            req.b->spec_valid = false;
            c->spec_req = req;
        }
    }
    kk_set_error("strict MGS sweep: the launch-per-vector route reported a grid-barrier timeout (internal error)");
    return KK_ERR_HIP;
}

// Projection for the low-sync MGS: p = V'(w - a*pre) into pinned slot `slot` (synchronised on
// return).  The Gram row of the newest basis vector (column c0+m-1) rides along as a second
// right-hand side of the same kernel when it is the only row missing -- no extra pass over V.
static int lowsync_project(kk_basis b, int c0, int m, const double* w, const double* pre_vec, const double* pre_a,
                           int slot) {
    kk_ctx c = b->ctx;
    KK_TRY(gram_ensure(b, c0 + m - 1));
    const int newest = c0 + m - 1;
    const bool ride = (b->gram_rows == newest && newest > 0 && c0 == 0);
    if (!ride) KK_TRY(gram_ensure(b, c0 + m));
    KK_TRY(kk_launch_project(c, b->col(c0), b->ld, m, w, pre_vec, pre_a, ride ? b->col(newest) : nullptr, WSP(c, WS_S), WSP(c, WS_G)));
    KK_TRY(ws_fetch_async(c, WS_S, m, slot));
    if (ride) KK_TRY(ws_fetch_async(c, WS_G, m, slot));
    KK_TRY(stream_sync(c));
    if (ride) {
        memcpy(&b->gram[(size_t)newest * b->cap], pin(c, WS_G, slot), (m - 1) * sizeof(double));
        b->gram_rows = newest + 1;
        KK_TRY(gram_upload_rows(b, newest, newest + 1));
    }
    return KK_OK;
}
// Device-side variant (c0 == 0): p = V'(w - a*pre) [+ Gram row of the newest vector riding along],
// then (I + L) s = p solved ON THE DEVICE; coefficients (s, with *a0_dev added to the last one) land in
// ws[ws_coef..], plain s in ws[ws_s..].  No host synchronisation.  If *rode, the caller must fetch
// ws[WS_G .. WS_G+m-1) with its final read-back and hand it to lowsync_commit_row().
int lowsync_project_dev(kk_basis b, int m, const double* w, const double* pre_vec, const double* pre_a,
                               const double* a0_dev, int64_t ws_coef, int64_t ws_s, bool* rode, int rows_in_stream) {
    kk_ctx c = b->ctx;
    const int newest = m - 1;
    // rows_in_stream (run-ahead of a whole step): Gram rows below this index are valid ON THE DEVICE -- the solve kernels of the steps in
    // the stream store them -- although the host mirror has not received them yet; they arrive with those steps' read-backs
    const int have = std::max(b->gram_rows, rows_in_stream);
    if (have < newest) KK_TRY(gram_ensure(b, newest));  // only after the basis was transformed (restart)
    if (b->gram_rows < 1) b->gram_rows = 1;
    const bool ride = (std::max(b->gram_rows, rows_in_stream) == newest && newest > 0);
    KK_TRY(gram_device(b));
    KK_TRY(kk_launch_project(c, b->col(0), b->ld, m, w, pre_vec, pre_a, ride ? b->col(newest) : nullptr, WSP(c, WS_S),
                             WSP(c, WS_G)));
    KK_TRY(kk_launch_lowsync_solve(c, WSP(c, WS_S), ride ? WSP(c, WS_G) : nullptr, b->d_gram, b->cap, m, newest, a0_dev,
                                   WSP(c, ws_coef), WSP(c, ws_s)));
    *rode = ride;
    return KK_OK;
}
void lowsync_commit_row(kk_basis b, int m, const double* g_host) {
    const int newest = m - 1;
    memcpy(&b->gram[(size_t)newest * b->cap], g_host, (m - 1) * sizeof(double));
    b->gram_rows = newest + 1;
}
// low-sync MGS sweep: p = V'w (one pass), s = (I+L)^-1 p on the host, w -= V s.
static int pass_mgs_lowsync(kk_basis b, int c0, int m, double* w, double* s_out, bool want_norm, int slot) {
    kk_ctx c = b->ctx;
    KK_TRY(lowsync_project(b, c0, m, w, nullptr, nullptr, slot));
    kk_coef ch;
    memset(&ch, 0, sizeof(ch));
    memcpy(ch.v, pin(c, WS_S, slot), m * sizeof(double));
    gram_solve(b, c0, m, ch.v);
    memcpy(s_out, ch.v, m * sizeof(double));
    KK_TRY(kk_launch_unproject(c, b->col(c0), b->ld, m, w, w, &ch, nullptr, -1.0, 1.0, -1, nullptr,
                               want_norm ? SCP(c, SC_NRM2) : nullptr));
    if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, slot));
    return KK_OK;
}

// orthogonalize!!(w, b[c0:c0+m), x, alg) -- all six algorithms (orthonormal.jl:378-452).
// On return x[0..m) holds the accumulated coefficients; *nrm = |w| if want_norm.
// orthogonalize!! against MORE than KK_MAX_M basis vectors (the reference knows no limit: krylovdim = 300 is legal): the
// kernels take panels of <= KK_MAX_M columns, so a classical pass is projected panel by panel (the same inner products) and
// subtracted panel by panel, a modified sweep runs the strict order through the panels one after the other.  One host
// synchronisation per panel -- this is the rarely used wide route, not the tuned one.
static int orth_run_wide(kk_basis b, int c0, int m, double* w, kk_orth_t alg, double eta, double* x, double* nrm, int* npasses,
                         bool want_norm) {
    kk_ctx c = b->ctx;
    const int64_t ld = b->ld;
    const bool classical = (alg == KK_CGS || alg == KK_CGS2 || alg == KK_CGSIR);
    const bool ir = (alg == KK_CGSIR || alg == KK_MGSIR);
    double nn = 0;
    auto one_pass = [&](double* s, bool norm) -> int {   // s[0..m) = coefficients of ONE pass over all panels; nn = |w| if norm
        if (classical) {
            for (int j0 = 0; j0 < m; j0 += KK_MAX_M) {
                const int mm = std::min(KK_MAX_M, m - j0);
                KK_TRY(kk_launch_project(c, b->col(c0 + j0), ld, mm, w, nullptr, nullptr, nullptr, WSP(c, WS_S), WSP(c, WS_G)));
                KK_TRY(ws_fetch_async(c, WS_S, mm, 0));
                KK_TRY(stream_sync(c));
                memcpy(s + j0, pin(c, WS_S, 0), mm * sizeof(double));
            }
            for (int j0 = 0; j0 < m; j0 += KK_MAX_M) {
                const int mm = std::min(KK_MAX_M, m - j0);
                kk_coef ch;
                memset(&ch, 0, sizeof(ch));
                memcpy(ch.v, s + j0, mm * sizeof(double));
                const bool last = j0 + mm >= m;
                KK_TRY(kk_launch_unproject(c, b->col(c0 + j0), ld, mm, w, w, &ch, nullptr, -1.0, 1.0, -1, nullptr,
                                           (last && norm) ? SCP(c, SC_NRM2) : nullptr));
            }
            if (norm) {
                KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
                KK_TRY(stream_sync(c));
                nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            }
        } else {
            for (int j0 = 0; j0 < m; j0 += KK_MAX_M) {
                const int mm = std::min(KK_MAX_M, m - j0);
                const bool last = j0 + mm >= m;
                const int64_t offs[1] = {WS_S};
                KK_TRY(strict_sweeps_synced(c, b->col(c0 + j0), ld, mm, 1, w, offs, last && norm, 0, false));
                memcpy(s + j0, pin(c, WS_S, 0), mm * sizeof(double));
                if (last && norm) nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            }
        }
        return KK_OK;
    };
    std::vector<double> tmp(m);
    int passes = 0;
    if (!ir) {
        const bool two = (alg == KK_CGS2 || alg == KK_MGS2);
        KK_TRY(one_pass(x, want_norm && !two));
        passes = 1;
        if (two) {
            KK_TRY(one_pass(tmp.data(), want_norm));
            for (int j = 0; j < m; ++j) x[j] += tmp[j];
            passes = 2;
        }
    } else {   // orthonormal.jl:400-412 / :440-452
        KK_TRY(kk_launch_nrm2(c, w, ld, SCP(c, SC_NRM2B)));
        KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2B, 2, 1));
        KK_TRY(stream_sync(c));
        double nold = pin(c, WS_SCAL + SC_NRM2B, 1)[1];
        KK_TRY(one_pass(x, true));
        passes = 1;
        while (KK_EPS < nn && nn < eta * nold) {
            nold = nn;
            KK_TRY(one_pass(tmp.data(), true));
            for (int j = 0; j < m; ++j) x[j] += tmp[j];
            ++passes;
        }
    }
    if (nrm) *nrm = nn;
    if (npasses) *npasses = passes;
    return KK_OK;
}

int orth_run(kk_basis b, int c0, int m, double* w, kk_orth_t alg, double eta, double* x, double* nrm,
                    int* npasses, bool want_norm) {
    kk_ctx c = b->ctx;
    const double* V = b->col(c0);
    const int64_t ld = b->ld;
    int passes = 0;
    double nn = 0;
    KK_TRY(route_agree(b));   // (cross-rank context: the route of the sweeps below is decided with the longest shard of this slab)
    if (m > KK_MAX_M) return orth_run_wide(b, c0, m, w, alg, eta, x, nrm, npasses, want_norm);
    if (m == 0) {
        if (want_norm || alg == KK_CGSIR || alg == KK_MGSIR) {
            KK_TRY(kk_launch_nrm2(c, w, ld, SCP(c, SC_NRM2)));
            KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
            KK_TRY(stream_sync(c));
            nn = pin(c, WS_SCAL + SC_NRM2)[1];
        }
        if (nrm) *nrm = nn;
        if (npasses) *npasses = 0;
        return KK_OK;
    }
    const bool lowsync = c0 == 0 && kk_mgs_lowsync(c, ld, m);
    std::vector<double> tmp(m);
    switch (alg) {
        case KK_CGS: {
            KK_TRY(pass_cgs(c, V, ld, m, w, want_norm, 0));
            KK_TRY(final_sync(c));
            memcpy(x, pin(c, WS_S, 0), m * sizeof(double));
            nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            passes = 1;
        } break;
        case KK_CGS2: {  // :394-399
            if (c->fuse_passes && m <= 128) {
                // s1 = V'w ; [w1 = w - V s1 ; s2 = V'w1] fused (V read once) ; w2 = w1 - V s2 (+ norm)
                KK_TRY(kk_launch_project(c, V, ld, m, w, nullptr, nullptr, nullptr, WSP(c, WS_S), WSP(c, WS_G)));
                KK_TRY(kk_launch_unproj_proj(c, V, ld, m, w, w, nullptr, WSP(c, WS_S), WSP(c, WS_G), nullptr));
                KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, WSP(c, WS_G), -1.0, 1.0, -1, nullptr,
                                           want_norm ? SCP(c, SC_NRM2) : nullptr));
                KK_TRY(fetch_through_scalars(c, WS_S, 0));   // s1 (WS_S), s2 (WS_G), norm
                KK_TRY(final_sync(c));
                for (int j = 0; j < m; ++j) x[j] = pin(c, WS_S, 0)[j] + pin(c, WS_G, 0)[j];
                nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            } else {
                KK_TRY(pass_cgs(c, V, ld, m, w, false, 0));
                KK_TRY(pass_cgs(c, V, ld, m, w, want_norm, 1));
                KK_TRY(final_sync(c));
                for (int j = 0; j < m; ++j) x[j] = pin(c, WS_S, 0)[j] + pin(c, WS_S, 1)[j];
                nn = pin(c, WS_SCAL + SC_NRM2, 1)[1];
            }
            passes = 2;
        } break;
        case KK_CGSIR: {  // :400-412
            KK_TRY(kk_launch_nrm2(c, w, ld, SCP(c, SC_NRM2B)));
            KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2B, 2, 1));
            KK_TRY(pass_cgs(c, V, ld, m, w, true, 0));
            KK_TRY(stream_sync(c));
            double nold = pin(c, WS_SCAL + SC_NRM2B, 1)[1];
            memcpy(x, pin(c, WS_S, 0), m * sizeof(double));
            nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            passes = 1;
            while (KK_EPS < nn && nn < eta * nold) {
                nold = nn;
                KK_TRY(pass_cgs(c, V, ld, m, w, true, 0));
                KK_TRY(stream_sync(c));
                for (int j = 0; j < m; ++j) x[j] += pin(c, WS_S, 0)[j];
                nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
                ++passes;
            }
        } break;
        case KK_MGS: {
            if (lowsync && c->fuse_passes) {
                bool rode = false;
                KK_TRY(lowsync_project_dev(b, m, w, nullptr, nullptr, nullptr, WS_X, WS_Y, &rode));
                KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, WSP(c, WS_X), -1.0, 1.0, -1, nullptr,
                                           want_norm ? SCP(c, SC_NRM2) : nullptr));
                KK_TRY(fetch_through_scalars(c, rode ? WS_G : WS_Y, 0));   // [Gram row,] coefficients (WS_Y), norm
                KK_TRY(final_sync(c));
                memcpy(x, pin(c, WS_Y, 0), m * sizeof(double));
                if (rode) lowsync_commit_row(b, m, pin(c, WS_G, 0));
            } else if (lowsync) {
                KK_TRY(pass_mgs_lowsync(b, c0, m, w, x, want_norm, 0));
                KK_TRY(final_sync(c));
            } else {
                const int64_t offs[1] = {WS_S};
                KK_TRY(strict_sweeps_synced(c, V, ld, m, 1, w, offs, want_norm, 0, true));
                memcpy(x, pin(c, WS_S, 0), m * sizeof(double));
            }
            nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            passes = 1;
        } break;
        case KK_MGS2: {  // :434-439
            if (lowsync && c->fuse_passes && m <= 128) {
                // p1 = V'w -> s1 = (I+L)^-1 p1 ; [w1 = w - V s1 ; p2 = V'w1] fused ; s2 = (I+L)^-1 p2 ; w2 = w1 - V s2
                // -- both triangular solves on the device: ONE host synchronisation for the whole 2-pass step
                bool rode = false;
                KK_TRY(lowsync_project_dev(b, m, w, nullptr, nullptr, nullptr, WS_X, WS_Y, &rode));
                KK_TRY(kk_launch_unproj_proj(c, V, ld, m, w, w, nullptr, WSP(c, WS_X), WSP(c, WS_S), nullptr));
                KK_TRY(kk_launch_lowsync_solve(c, WSP(c, WS_S), nullptr, b->d_gram, b->cap, m, m - 1, nullptr, WSP(c, WS_X),
                                               WSP(c, WS_Z)));
                KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, WSP(c, WS_X), -1.0, 1.0, -1, nullptr,
                                           want_norm ? SCP(c, SC_NRM2) : nullptr));
                KK_TRY(fetch_through_scalars(c, rode ? WS_G : WS_Y, 0));   // [Gram row,] s1 (WS_Y), s2 (WS_Z), norm
                KK_TRY(final_sync(c));
                for (int j = 0; j < m; ++j) x[j] = pin(c, WS_Y, 0)[j] + pin(c, WS_Z, 0)[j];
                if (rode) lowsync_commit_row(b, m, pin(c, WS_G, 0));
            } else if (lowsync) {
                KK_TRY(pass_mgs_lowsync(b, c0, m, w, x, false, 0));
                KK_TRY(pass_mgs_lowsync(b, c0, m, w, tmp.data(), want_norm, 0));
                KK_TRY(final_sync(c));
                for (int j = 0; j < m; ++j) x[j] += tmp[j];
            } else {
                // the last axpy of sweep 1 is fused with the first dot of sweep 2
                const int64_t offs[2] = {WS_S, WS_G};
                KK_TRY(strict_sweeps_synced(c, V, ld, m, 2, w, offs, want_norm, 0, true));
                for (int j = 0; j < m; ++j) x[j] = pin(c, WS_S, 0)[j] + pin(c, WS_G, 0)[j];
            }
            nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            passes = 2;
        } break;
        case KK_MGSIR: {  // :440-452
            KK_TRY(kk_launch_nrm2(c, w, ld, SCP(c, SC_NRM2B)));
            KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2B, 2, 1));
            const int64_t ir_offs[1] = {WS_S};
            if (lowsync) {
                KK_TRY(pass_mgs_lowsync(b, c0, m, w, x, true, 0));
                KK_TRY(stream_sync(c));
            } else {
                KK_TRY(strict_sweeps_synced(c, V, ld, m, 1, w, ir_offs, true, 0, false));
            }
            double nold = pin(c, WS_SCAL + SC_NRM2B, 1)[1];
            if (!lowsync) memcpy(x, pin(c, WS_S, 0), m * sizeof(double));
            nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            passes = 1;
            while (KK_EPS < nn && nn < eta * nold) {
                nold = nn;
                if (lowsync) {
                    KK_TRY(pass_mgs_lowsync(b, c0, m, w, tmp.data(), true, 0));
                    KK_TRY(stream_sync(c));
                } else {
                    KK_TRY(strict_sweeps_synced(c, V, ld, m, 1, w, ir_offs, true, 0, false));
                }
                for (int j = 0; j < m; ++j) x[j] += lowsync ? tmp[j] : pin(c, WS_S, 0)[j];
                nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
                ++passes;
            }
        } break;
        default:
            kk_set_error("unknown orthogonalizer %d", (int)alg);
            return KK_ERR_INVALID;
    }
    if (nrm) *nrm = nn;
    if (npasses) *npasses = passes;
    return KK_OK;
}

KK_API int kk_orthogonalize(kk_basis b, int c0, int m, kk_basis bw, int cw, kk_orth_t alg, double eta, double* x,
                                double* nrm, int* npasses) {
    CHECK_BLOCK(b, c0, m); CHECK_COL(bw, cw); CHECK_SAME(b, bw);   // any m: more than KK_MAX_M vectors go panel by panel
    KK_CHECK(x || m == 0, KK_ERR_INVALID, "null x");
    KK_CHECK(!(bw == b && cw >= c0 && cw < c0 + m), KK_ERR_INVALID, "kk_orthogonalize: w aliases a basis column");
    gram_touch(bw, cw);
    return orth_run(b, c0, m, bw->col(cw), alg, eta, x, nrm, npasses, nrm != nullptr);
}

KK_API int kk_orthonormalize(kk_basis b, int c0, int m, kk_basis bw, int cw, kk_orth_t alg, double eta, double* x,
                                 double* nrm, int* npasses) {
    CHECK_BLOCK(b, c0, m); CHECK_COL(bw, cw); CHECK_SAME(b, bw);
    KK_CHECK(x || m == 0, KK_ERR_INVALID, "null x");
    KK_CHECK(!(bw == b && cw >= c0 && cw < c0 + m), KK_ERR_INVALID, "kk_orthonormalize: w aliases a basis column");
    gram_touch(bw, cw);
    kk_ctx c = b->ctx;
    double nn = 0;
    // MGS / MGS2 end with one sweep launch: the persistent kernel can store w / |w| itself
    c->persist_norm_done = false;
    c->persist_norm_req = c->fold_scale != 0 && (alg == KK_MGS || alg == KK_MGS2);
    const int st = orth_run(b, c0, m, bw->col(cw), alg, eta, x, &nn, npasses, true);
    c->persist_norm_req = false;
    KK_TRY(st);
    if (nrm) *nrm = nn;
    if (c->persist_norm_done && kk_persist_norm_applies(nn)) return KK_OK;   // already normalised at the kernel's commit
    return kk_launch_scal(c, bw->col(cw), bw->ld, 1.0 / nn, nullptr);  // scale!!(v, inv(beta))  :525
}

// _orthogonalize!!(v, q, alg) (orthonormal.jl:455-489)
int orth_vec_run(kk_ctx c, const double* q, double* w, int64_t ld, kk_orth_t alg, double eta, double* s_out,
                        double* nrm, bool want_norm) {
    double s = 0, nn = 0;
    if (alg == KK_CGS || alg == KK_MGS) {
        KK_TRY(kk_launch_mgs_step(c, w, ld, nullptr, nullptr, q, WSP(c, WS_S), nullptr));
        KK_TRY(kk_launch_mgs_step(c, w, ld, q, c->ws + WS_S, nullptr, nullptr, want_norm ? SCP(c, SC_NRM2) : nullptr));
        KK_TRY(ws_fetch_async(c, WS_S, 1, 0));
        if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
        KK_TRY(stream_sync(c));
        s = pin(c, WS_S)[0];
        nn = pin(c, WS_SCAL + SC_NRM2)[1];
    } else if (alg == KK_CGS2 || alg == KK_MGS2) {
        KK_TRY(kk_launch_mgs_step(c, w, ld, nullptr, nullptr, q, WSP(c, WS_S), nullptr));
        KK_TRY(kk_launch_mgs_step(c, w, ld, q, c->ws + WS_S, q, WSP(c, WS_S + 1), nullptr));
        KK_TRY(kk_launch_mgs_step(c, w, ld, q, c->ws + WS_S + 1, nullptr, nullptr, want_norm ? SCP(c, SC_NRM2) : nullptr));
        KK_TRY(ws_fetch_async(c, WS_S, 2, 0));
        if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
        KK_TRY(stream_sync(c));
        s = pin(c, WS_S)[0] + pin(c, WS_S)[1];
        nn = pin(c, WS_SCAL + SC_NRM2)[1];
    } else {
        KK_TRY(kk_launch_nrm2(c, w, ld, SCP(c, SC_NRM2B)));
        KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2B, 2, 1));
        KK_TRY(kk_launch_mgs_step(c, w, ld, nullptr, nullptr, q, WSP(c, WS_S), nullptr));
        KK_TRY(kk_launch_mgs_step(c, w, ld, q, c->ws + WS_S, nullptr, nullptr, SCP(c, SC_NRM2)));
        KK_TRY(ws_fetch_async(c, WS_S, 1, 0));
        KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
        KK_TRY(stream_sync(c));
        double nold = pin(c, WS_SCAL + SC_NRM2B, 1)[1];
        s = pin(c, WS_S)[0];
        nn = pin(c, WS_SCAL + SC_NRM2)[1];
        while (KK_EPS < nn && nn < eta * nold) {
            nold = nn;
            KK_TRY(kk_launch_mgs_step(c, w, ld, nullptr, nullptr, q, WSP(c, WS_S), nullptr));
            KK_TRY(kk_launch_mgs_step(c, w, ld, q, c->ws + WS_S, nullptr, nullptr, SCP(c, SC_NRM2)));
            KK_TRY(ws_fetch_async(c, WS_S, 1, 0));
            KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
            KK_TRY(stream_sync(c));
            s += pin(c, WS_S)[0];
            nn = pin(c, WS_SCAL + SC_NRM2)[1];
        }
    }
    if (s_out) *s_out = s;
    if (nrm) *nrm = nn;
    return KK_OK;
}

KK_API int kk_orthogonalize_vec(kk_basis bq, int cq, kk_basis bw, int cw, kk_orth_t alg, double eta, double* s,
                                    double* nrm) {
    CHECK_COL(bq, cq); CHECK_COL(bw, cw); CHECK_SAME(bq, bw);
    KK_CHECK(!(bq == bw && cq == cw), KK_ERR_INVALID, "kk_orthogonalize_vec: q and w must differ");
    gram_touch(bw, cw);
    return orth_vec_run(bq->ctx, bq->col(cq), bw->col(cw), bw->ld, alg, eta, s, nrm, nrm != nullptr);
}

