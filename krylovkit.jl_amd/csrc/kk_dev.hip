// libkrylov_hip.so, C ABI part 7: split-phase entry points whose scalar results stay in caller-owned device memory
// (row-sharded iterators of krylovkit_hip/dist.py).
#include "kk_host.h"

// ------------------------------------------------------------------------------------------
// split-phase API (row-sharded multi-GPU runs): local partials into caller-owned device buffers
// ------------------------------------------------------------------------------------------
KK_API int kk_apply_fused_dev(kk_op op, kk_basis b, int col_v, int col_prev, int col_w, double beta_old,
                                  int dot_mode, void* dev_dot) {
    KK_TRY(check_square_op(op, b));
    CHECK_COL(b, col_v); CHECK_COL(b, col_w);
    KK_CHECK(col_prev < b->cap && col_v != col_w, KK_ERR_INVALID, "kk_apply_fused_dev: bad columns");
    KK_CHECK(dot_mode == 0 || dev_dot, KK_ERR_INVALID, "kk_apply_fused_dev: dot requested without output buffer");
    gram_touch(b, col_w);
    kk_spmv_fuse f;
    if (col_prev >= 0) { f.vprev = b->col(col_prev); f.bprev = beta_old; }
    f.dot_mode = dot_mode;
    f.dot_out = (double*)dev_dot;
    return kk_launch_spmv(op->ctx, op->A, b->col(col_v), b->col(col_w), b->ld, f);
}
KK_API int kk_apply_fused_dev2(kk_op op, kk_basis b, int col_v, int col_prev, int col_w, const void* dev_xscale,
                                   const void* dev_bprev, double beta_old, int dot_mode, void* dev_dot) {
    KK_TRY(check_square_op(op, b));
    CHECK_COL(b, col_v); CHECK_COL(b, col_w);
    KK_CHECK(col_prev < b->cap && col_v != col_w, KK_ERR_INVALID, "kk_apply_fused_dev2: bad columns");
    KK_CHECK(dot_mode == 0 || dev_dot, KK_ERR_INVALID, "kk_apply_fused_dev2: dot requested without output buffer");
    gram_touch(b, col_w);
    kk_spmv_fuse f;
    f.xscale_dev = (const double*)dev_xscale;
    if (col_prev >= 0) { f.vprev = b->col(col_prev); f.bprev = beta_old; f.bprev_dev = (const double*)dev_bprev; }
    f.dot_mode = dot_mode;
    f.dot_out = (double*)dev_dot;
    return kk_launch_spmv(op->ctx, op->A, b->col(col_v), b->col(col_w), b->ld, f);
}
KK_API int kk_unproject_devcoef(kk_basis by, int cy, kk_basis b, int c0, int m, const void* dev_coef, double alpha,
                                    double beta, void* dev_nrm) {
    CHECK_RANGE(b, c0, m); CHECK_COL(by, cy); CHECK_SAME(b, by);
    KK_CHECK(dev_coef || m == 0, KK_ERR_INVALID, "null coef");
    KK_CHECK(!(by == b && cy >= c0 && cy < c0 + m), KK_ERR_INVALID, "kk_unproject_devcoef: y aliases a basis column");
    gram_touch(by, cy);
    return kk_launch_unproject(b->ctx, b->col(c0), b->ld, m, by->col(cy), by->col(cy), nullptr, (const double*)dev_coef, alpha,
                               beta, -1, nullptr, (double*)dev_nrm);
}
KK_API int kk_project_dev(kk_basis b, int c0, int m, kk_basis bx, int cx, int col_rhs2, void* dev_out) {
    CHECK_RANGE(b, c0, m); CHECK_COL(bx, cx); CHECK_SAME(b, bx);
    KK_CHECK(dev_out || m == 0, KK_ERR_INVALID, "null output");
    KK_CHECK(col_rhs2 < bx->cap, KK_ERR_INVALID, "kk_project_dev: rhs2 column out of range");
    if (m == 0) return KK_OK;
    double* o = (double*)dev_out;
    return kk_launch_project(b->ctx, b->col(c0), b->ld, m, bx->col(cx), nullptr, nullptr,
                             col_rhs2 >= 0 ? bx->col(col_rhs2) : nullptr, o, o + m);
}
KK_API int kk_unproject_dev(kk_basis by, int cy, kk_basis b, int c0, int m, const double* coef, double alpha,
                                double beta, void* dev_nrm) {
    CHECK_RANGE(b, c0, m); CHECK_COL(by, cy); CHECK_SAME(b, by);
    KK_CHECK(coef || m == 0, KK_ERR_INVALID, "null coef");
    KK_CHECK(!(by == b && cy >= c0 && cy < c0 + m), KK_ERR_INVALID, "kk_unproject_dev: y aliases a basis column");
    gram_touch(by, cy);
    kk_coef ch;
    memset(&ch, 0, sizeof(ch));
    for (int j = 0; j < m; ++j) ch.v[j] = coef[j];
    return kk_launch_unproject(b->ctx, b->col(c0), b->ld, m, by->col(cy), by->col(cy), &ch, nullptr, alpha, beta, -1,
                               nullptr, (double*)dev_nrm);
}
KK_API int kk_dot_dev(kk_basis bx, int cx, kk_basis by, int cy, void* dev_out) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    KK_CHECK(dev_out, KK_ERR_INVALID, "null output");
    return kk_launch_dot(bx->ctx, bx->col(cx), by->col(cy), bx->ld, (double*)dev_out);
}
KK_API int kk_nrm2_dev(kk_basis bx, int cx, void* dev_out3) {
    CHECK_COL(bx, cx);
    KK_CHECK(dev_out3, KK_ERR_INVALID, "null output");
    return kk_launch_nrm2(bx->ctx, bx->col(cx), bx->ld, (double*)dev_out3);
}
KK_API int kk_axpy_dev(kk_basis by, int cy, kk_basis bx, int cx, const void* dev_a, double sign) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    KK_CHECK(dev_a, KK_ERR_INVALID, "null scalar");
    gram_touch(by, cy);
    return kk_launch_axpby(by->ctx, by->col(cy), bx->col(cx), by->ld, 0.0, 1.0, (const double*)dev_a, sign, 1);
}
KK_API int kk_scal_rsqrt_dev(kk_basis bx, int cx, const void* dev_nrm2) {
    CHECK_COL(bx, cx);
    KK_CHECK(dev_nrm2, KK_ERR_INVALID, "null scalar");
    gram_touch(bx, cx);
    return kk_launch_scal(bx->ctx, bx->col(cx), bx->ld, 0.0, (const double*)dev_nrm2, 1);
}

// coefficient algebra of a row-sharded Lanczos step on the device, between its two all-reduces: see k_lanczos_coef.
// buf_dev = [alpha0 | V'w (m) | V'v (m)], L_dev = cap x cap row-major strictly-lower Gram matrix (lowsync only),
// coef_out_dev (m), res_dev (>= 2).  Stream-ordered, no host synchronisation.
KK_API int kk_lanczos_coef_dev(kk_ctx c, const void* buf_dev, void* L_dev, int cap, int m, int lowsync, void* coef_out_dev,
                               void* res_dev) {
    KK_CHECK(c && buf_dev && coef_out_dev && res_dev, KK_ERR_INVALID, "null arg");
    KK_CHECK(m >= 1 && m <= KK_MAX_M, KK_ERR_INVALID, "kk_lanczos_coef_dev: m = %d out of range", m);
    KK_CHECK(!lowsync || (L_dev && cap >= m), KK_ERR_INVALID, "kk_lanczos_coef_dev: Gram matrix missing or too small");
    return kk_launch_lanczos_coef(c, (const double*)buf_dev, (double*)L_dev, cap, m, lowsync, (double*)coef_out_dev, (double*)res_dev);
}
// sc_dev = {1/sqrt(*nrm2_dev), sqrt(*nrm2_dev)}, *res2_dev = *nrm2_dev (device scalars of the speculative next apply)
KK_API int kk_norm_scalars_dev(kk_ctx c, const void* nrm2_dev, void* sc_dev, void* res2_dev) {
    KK_CHECK(c && nrm2_dev && sc_dev && res2_dev, KK_ERR_INVALID, "null arg");
    return kk_launch_norm_scalars(c, (const double*)nrm2_dev, (double*)sc_dev, (double*)res2_dev);
}

