// libkrylov_hip.so, C ABI part 3: fused iteration bodies of the short-recurrence solvers (CG, BiCGStab, LSMR).
#include "kk_host.h"

// One CG iteration body (linsolve/cg.jl:60-66) with ONE host synchronisation:
//   [p = r + beta p]  (skipped when beta_is_first)   q = a0 p + a1 A p with fused <p,q> (stays on the device)
//   alpha = rho / <p,q> formed inside the update kernel ; x += alpha p ; r -= alpha q ; |r|
// columns of `b`: cx, cr, cp, cq.  Returns <p,q> and |r|.
KK_API int kk_cg_iterate(kk_op op, kk_basis b, int cx, int cr, int cp, int cq, double a0, double a1, double beta,
                             int first, double rho, double* pq, double* rnorm) {
    KK_TRY(check_square_op(op, b));
    CHECK_COL(b, cx); CHECK_COL(b, cr); CHECK_COL(b, cp); CHECK_COL(b, cq);
    KK_CHECK(pq && rnorm, KK_ERR_INVALID, "null output");
    kk_ctx c = b->ctx;
    gram_touch(b, std::min(std::min(cx, cr), std::min(cp, cq)));
    if (!first) KK_TRY(kk_launch_axpby(c, b->col(cp), b->col(cr), b->ld, 1.0, beta, nullptr, 1.0, 0));   // p = add!!(p, r, 1, beta)
    kk_spmv_fuse f;
    f.a0 = a0; f.a1 = a1;
    f.dot_mode = 2;
    f.dot_out = SCP(c, SC_DOT);
    KK_TRY(kk_launch_spmv(c, op->A, b->col(cp), b->col(cq), b->ld, f));
    KK_TRY(kk_launch_cg_update(c, b->col(cx), b->col(cp), b->col(cr), b->col(cq), b->ld, rho, SCP(c, SC_DOT), SCP(c, SC_NRM2)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 4, 0));   // NRM2, NRM, INVNRM, DOT
    KK_TRY(stream_sync(c));
    *pq = pin(c, WS_SCAL + SC_DOT)[0];
    *rnorm = pin(c, WS_SCAL + SC_NRM)[0];
    return KK_OK;
}
// x += alpha p ; r -= alpha q ; *rnorm = |r|   (linsolve/cg.jl:63-66 in one pass)
KK_API int kk_cg_update(kk_basis bx, int cx, kk_basis bp, int cp, kk_basis br, int cr, kk_basis bq, int cq, double alpha,
                            double* rnorm) {
    CHECK_COL(bx, cx); CHECK_COL(bp, cp); CHECK_COL(br, cr); CHECK_COL(bq, cq);
    CHECK_SAME(bx, bp); CHECK_SAME(bx, br); CHECK_SAME(bx, bq);
    KK_CHECK(rnorm, KK_ERR_INVALID, "null output");
    kk_ctx c = bx->ctx;
    gram_touch(bx, cx); gram_touch(br, cr);
    KK_TRY(kk_launch_cg_update(c, bx->col(cx), bp->col(cp), br->col(cr), bq->col(cq), bx->ld, alpha, nullptr, SCP(c, SC_NRM2)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
    KK_TRY(stream_sync(c));
    *rnorm = pin(c, WS_SCAL + SC_NRM2)[1];
    return KK_OK;
}

// ---- BiCGStab (linsolve/bicgstab.jl:118-199), one call per half step.  cols = {x, r, r_shadow, p, v, s, t}.
// The recurrence scalars rho, sigma, alpha, omega never leave the device; the host reads only the two norms the
// reference tests against tol (and alpha / rho for the rare explicit-residual branches).
static int bicg_apply_t(kk_op op, kk_basis b, const int* cols, double a0, double a1) {
    kk_ctx c = b->ctx;
    kk_spmv_fuse f;                       // t = (a0 + a1 A) s with <t,s> and <t,t>     (:157-160)
    f.a0 = a0; f.a1 = a1;
    f.dot_mode = 2; f.dot_out = SCP(c, SC_BICG + 5);
    f.nrm_out = SCP(c, SC_BICG + 6);
    return kk_launch_spmv(c, op->A, b->col(cols[5]), b->col(cols[6]), b->ld, f);
}
// BiCG half: [p = r + beta (p_prev - omega v_prev)] ; v = (a0 + a1 A) p ; sigma = <r_shadow, v> ; alpha = rho/sigma ;
// s = r - alpha v, then (touching nothing else) the stabiliser's t = (a0 + a1 A) s.  Everything is enqueued; the
// read-back of the scalars goes to pinned slot `slot` and is marked by event `ev`.
// cols = {x, r, r_shadow, p, v, s, t, p_prev, v_prev}; p_prev/v_prev may equal p/v (in place) or be the other half of
// a double buffer, so that a run-ahead half can be discarded without having destroyed p and v.
static int bicg_half_enqueue(kk_op op, kk_basis b, const int* cols, double a0, double a1, int mode, double rho, int slot,
                             hipEvent_t ev) {
    kk_ctx c = b->ctx;
    gram_touch(b, *std::min_element(cols, cols + 7));
    double* sc = SCP(c, SC_BICG);
    double *r = b->col(cols[1]), *rs = b->col(cols[2]), *pp = b->col(cols[3]), *v = b->col(cols[4]), *sv = b->col(cols[5]);
    if (mode != 0) KK_TRY(kk_launch_set_scalar(c, sc, rho));
    if (mode != 1) KK_TRY(kk_launch_bicg_p(c, pp, b->col(cols[7]), r, b->col(cols[8]), b->ld, sc));
    kk_spmv_fuse f;
    f.a0 = a0; f.a1 = a1;
    f.dot_mode = 3; f.dot_vec = rs; f.dot_out = SCP(c, SC_BICG + 2);
    KK_TRY(kk_launch_spmv(c, op->A, pp, v, b->ld, f));
    KK_TRY(kk_launch_bicg_s(c, sv, r, v, b->ld, sc, SCP(c, SC_BICG_SN)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_BICG, 16, slot));
    KK_HIP(hipEventRecord(ev, c->stream));
    return bicg_apply_t(op, b, cols, a0, a1);
}
static int bicg_check_cols(kk_basis b, const int* cols, int n) {
    KK_CHECK(cols, KK_ERR_INVALID, "null cols");
    for (int i = 0; i < n; ++i) CHECK_COL(b, cols[i]);
    return KK_OK;
}
// mode 0: rho is the device value left by kk_bicgstab_full; 1 (first iteration, :34-52): p already equals r, rho comes
// from the host; 2: rho comes from the host (r was replaced by the explicit residual, :175-179); 3: collect the half
// that the previous kk_bicgstab_full already enqueued (ahead_cols) -- nothing is launched, the host only waits.
KK_API int kk_bicgstab_half(kk_op op, kk_basis b, const int* cols, double a0, double a1, int mode, double rho,
                                double* snorm, double* alpha) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(snorm && alpha, KK_ERR_INVALID, "null arg");
    KK_CHECK(mode >= 0 && mode <= 3, KK_ERR_INVALID, "kk_bicgstab_half: mode must be 0..3");
    KK_TRY(bicg_check_cols(b, cols, 9));
    kk_ctx c = b->ctx;
    int slot = 0;
    if (mode == 3) {
        KK_CHECK(c->bicg_ahead, KK_ERR_INVALID, "kk_bicgstab_half: mode 3 without a run-ahead half");
        slot = 1;
        KK_HIP(hipEventSynchronize(c->ev_fetch2));
    } else {
        KK_TRY(bicg_half_enqueue(op, b, cols, a0, a1, mode, rho, 0, c->ev_fetch));
        KK_HIP(hipEventSynchronize(c->ev_fetch));
    }
    c->bicg_ahead = false;
    *snorm = pin(c, WS_SCAL + SC_BICG_SN, slot)[1];
    *alpha = pin(c, WS_SCAL + SC_BICG, slot)[3];
    return KK_OK;
}
// stabiliser half: omega = <t,s>/<t,t> ; x += alpha p + omega s ; r = s - omega t ; returns |r|, rho = <r_shadow, r>
// and omega.  redo_t: s was replaced by the host (explicit residual, :143-146) -> recompute t first.
// ahead_cols (9 columns, or NULL): enqueue the NEXT BiCG half on those columns before waiting, so that the GPU keeps
// working through the host round trip; the next kk_bicgstab_half(mode 3) collects it, any other mode discards it.
KK_API int kk_bicgstab_full(kk_op op, kk_basis b, const int* cols, double a0, double a1, int redo_t,
                                const int* ahead_cols, double* rnorm, double* rho, double* omega) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(rnorm && rho && omega, KK_ERR_INVALID, "null arg");
    KK_TRY(bicg_check_cols(b, cols, 7));
    if (ahead_cols) KK_TRY(bicg_check_cols(b, ahead_cols, 9));
    kk_ctx c = b->ctx;
    gram_touch(b, *std::min_element(cols, cols + 7));
    double* sc = SCP(c, SC_BICG);
    if (redo_t) KK_TRY(bicg_apply_t(op, b, cols, a0, a1));
    KK_TRY(kk_launch_bicg_xr(c, b->col(cols[0]), b->col(cols[3]), b->col(cols[5]), b->col(cols[6]), b->col(cols[1]),
                             b->col(cols[2]), b->ld, sc, SCP(c, SC_BICG_RN), sc));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_BICG, 16, 0));
    KK_TRY(fetch_mark(c));
    if (ahead_cols) {
        KK_TRY(bicg_half_enqueue(op, b, ahead_cols, a0, a1, 0, 0.0, 1, c->ev_fetch2));
        c->bicg_ahead = true;
    }
    KK_TRY(fetch_wait(c));
    *rnorm = pin(c, WS_SCAL + SC_BICG_RN)[1];
    *rho = pin(c, WS_SCAL + SC_BICG)[0];
    *omega = pin(c, WS_SCAL + SC_BICG)[4];
    return KK_OK;
}

// ---- LSMR (lssolve/lsmr.jl:61-128) fused vector updates.
// Ah = Av - c Ah ; u = Av - alpha u ; returns beta = |u|     (columns of one basis in the row space of A, :64-68)
KK_API int kk_lsmr_step_u(kk_basis b, int c_av, int c_ah, int c_u, double c, double alpha, double* beta) {
    CHECK_COL(b, c_av); CHECK_COL(b, c_ah); CHECK_COL(b, c_u);
    KK_CHECK(beta, KK_ERR_INVALID, "null output");
    KK_CHECK(c_av != c_ah && c_av != c_u && c_ah != c_u, KK_ERR_INVALID, "kk_lsmr_step_u: columns must differ");
    kk_ctx c_ = b->ctx;
    gram_touch(b, std::min(c_ah, c_u));
    KK_TRY(kk_launch_lsmr_u(c_, b->col(c_av), b->col(c_ah), b->col(c_u), b->ld, c, alpha, SCP(c_, SC_NRM2)));
    KK_TRY(ws_fetch_async(c_, WS_SCAL + SC_NRM2, 2, 0));
    KK_TRY(stream_sync(c_));
    *beta = pin(c_, WS_SCAL + SC_NRM2)[1];
    return KK_OK;
}
// hbar = h - c1 hbar ; x += c2 hbar ; h = v - c3 h (skipped when cv < 0).  Used for (h, hbar, x, v) in the domain of A
// and, with cv < 0, for (Ah, Ahbar, r) with c2 negated in the row space (:121-128).  Stream-ordered, no host sync.
KK_API int kk_lsmr_update(kk_basis b, int ch, int chbar, int cx, kk_basis bv, int cv, double c1, double c2, double c3) {
    CHECK_COL(b, ch); CHECK_COL(b, chbar); CHECK_COL(b, cx);
    KK_CHECK(ch != chbar && ch != cx && chbar != cx, KK_ERR_INVALID, "kk_lsmr_update: columns must differ");
    const double* v = nullptr;
    if (cv >= 0) {
        CHECK_COL(bv, cv); CHECK_SAME(b, bv);
        v = bv->col(cv);
    }
    gram_touch(b, std::min(std::min(ch, chbar), cx));
    return kk_launch_lsmr_hx(b->ctx, b->col(ch), b->col(chbar), b->col(cx), v, b->ld, c1, c2, c3);
}

