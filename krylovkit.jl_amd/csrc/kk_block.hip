// libkrylov_hip.so, C ABI part 6: block (multi-vector) operations and the BlockLanczos steps
// (src/factorizations/blocklanczos.jl).
#include "kk_host.h"
#include <cstring>

// ------------------------------------------------------------------------------------------
// BlockLanczos (src/factorizations/blocklanczos.jl)
// ------------------------------------------------------------------------------------------

// M_host (p x q, ldm) = X' Y.  Panel mode: MFMA gram tiles into device scratch, ONE D2H + sync.
int block_inner_run(kk_ctx c, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
                           double* M, int ldm) {
    if (p == 0 || q == 0) return KK_OK;
    KK_CHECK((int64_t)p * q <= KK_BLK_SCRATCH, KK_ERR_UNSUPPORTED, "block_inner: %d x %d too large", p, q);
    if (c->block_mode == 0) {  // strict: p*q scalar inner calls (blocklanczos.jl:47-50)
        for (int j = 0; j < q; ++j)
            for (int i = 0; i < p; ++i)
                KK_TRY(kk_launch_dot(c, X + (int64_t)i * ldx, Y + (int64_t)j * ldy, ld, c->blk + i + (int64_t)p * j));
    } else {
        for (int j0 = 0; j0 < q; j0 += 16)
            for (int i0 = 0; i0 < p; i0 += 128)
                KK_TRY(kk_launch_block_gram(c, X + (int64_t)i0 * ldx, ldx, std::min(128, p - i0), Y + (int64_t)j0 * ldy, ldy,
                                            std::min(16, q - j0), ld, c->blk + i0 + (int64_t)p * j0, p));
    }
    if (c->block_mode != 0) KK_TRY(kk_allreduce(c, c->blk, (int64_t)p * q));
    KK_HIP(hipMemcpyAsync(c->h_blk, c->blk, (size_t)p * q * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    KK_TRY(stream_sync(c));
    for (int j = 0; j < q; ++j) memcpy(M + (size_t)j * ldm, c->h_blk + (size_t)j * p, p * sizeof(double));
    return KK_OK;
}

KK_API int kk_block_inner(kk_basis bx, int cx, int p, kk_basis by, int cy, int q, double* M, int ldm) {
    CHECK_BLOCK(bx, cx, p); CHECK_BLOCK(by, cy, q); CHECK_SAME(bx, by);
    KK_CHECK(M && ldm >= p, KK_ERR_INVALID, "kk_block_inner: bad output");
    return block_inner_run(bx->ctx, bx->col(cx), bx->ld, p, by->col(cy), by->ld, q, bx->ld, M, ldm);
}

KK_API int kk_block_apply(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, int nb) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    CHECK_BLOCK(bx, cx, nb); CHECK_BLOCK(by, cy, nb);
    KK_TRY(check_apply(op, 0, bx, by));
    KK_CHECK(!(bx == by && cx < cy + nb && cy < cx + nb), KK_ERR_INVALID, "kk_block_apply: blocks overlap");
    gram_touch(by, cy);
    if (op->gather) {   // row-sharded rectangular map: its ghost-only matrix reads the all-gathered buffer, one column at a time
        for (int j = 0; j < nb; ++j) KK_TRY(rect_apply(op, 0, bx->col(cx + j), by->col(cy + j)));
        return KK_OK;
    }
    return kk_launch_spmm(op->ctx, op->A, bx->col(cx), bx->ld, by->col(cy), by->ld, nb);
}

// stage the m x nb coefficient panel S[:, j0 : j0+nb] (host, column-major, leading dimension lds) for
// kk_launch_block_update: row-major on the device with the row stride padded to the kernel's width (4 / 8 / 16,
// zeros in the pad) so that the kernel reads whole rows with wide scalar loads and needs no j < nb branches
static int stage_coef(kk_ctx c, const double* S, int lds, int m, int j0, int nb, const double** dev_out) {
    // ring of KK_STAGE_SLOTS staging slots (same offset in the pinned and the device scratch): the host only waits
    // for the stream when the ring wraps, not before every launch
    const int st = kk_bu_stride(nb);
    KK_CHECK((int64_t)m * st <= KK_STAGE_DOUBLES, KK_ERR_UNSUPPORTED, "block update: coefficient panel %d x %d too large", m, st);
    if (c->stage_slot >= KK_STAGE_SLOTS) {
        KK_TRY(stream_sync(c));
        c->stage_slot = 0;
    }
    const size_t off = (size_t)c->stage_slot * KK_STAGE_DOUBLES;
    c->stage_slot++;
    for (int i = 0; i < m; ++i) {
        double* row = c->h_blk + off + (size_t)i * st;
        for (int j = 0; j < nb; ++j) row[j] = S[i + (size_t)lds * (j0 + j)];
        for (int j = nb; j < st; ++j) row[j] = 0.0;
    }
    if (m > 0) KK_HIP(hipMemcpyAsync(c->blk + off, c->h_blk + off, (size_t)m * st * sizeof(double), hipMemcpyHostToDevice, c->stream));
    *dev_out = c->blk + off;
    return KK_OK;
}
// W[:, j] = beta W[:, j] + alpha V S[:, j]  (S host, m x q col-major lds); norms_host optional
int block_update_run(kk_ctx c, const double* V, int64_t ld, int m, double* W, int64_t ldw, int q, const double* S,
                     int lds, double alpha, double beta, double* norms) {
    if (q == 0) return KK_OK;
    KK_CHECK((int64_t)m * 16 + 64 <= KK_BLK_SCRATCH / 2, KK_ERR_UNSUPPORTED, "block_update: m=%d too large", m);
    double* nrm_dev = c->blk + KK_BLK_SCRATCH / 2;  // q doubles
    for (int j0 = 0; j0 < q; j0 += 16) {
        const int nb = std::min(16, q - j0);
        const double* Sd = nullptr;
        KK_TRY(stage_coef(c, S, lds, m, j0, nb, &Sd));
        KK_TRY(kk_launch_block_update(c, V, ld, m, W + (int64_t)j0 * ldw, W + (int64_t)j0 * ldw, ldw, ldw, nb, Sd, alpha,
                                      beta, norms ? nrm_dev + j0 : nullptr));
    }
    if (norms) {
        KK_HIP(hipMemcpyAsync(c->h_blk + KK_BLK_SCRATCH / 2, nrm_dev, q * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        KK_TRY(stream_sync(c));
        for (int j = 0; j < q; ++j) norms[j] = std::sqrt(c->h_blk[KK_BLK_SCRATCH / 2 + j]);
    }
    return KK_OK;
}

KK_API int kk_block_update(kk_basis bw, int cw, int q, kk_basis b, int c0, int m, const double* S, int lds,
                               double alpha, double beta, double* norms) {
    CHECK_BLOCK(bw, cw, q); CHECK_BLOCK(b, c0, m); CHECK_SAME(bw, b);
    KK_CHECK(S || m == 0, KK_ERR_INVALID, "null S");
    KK_CHECK(lds >= m, KK_ERR_DIM, "kk_block_update: lds < m");
    KK_CHECK(!(bw == b && cw < c0 + m && c0 < cw + q), KK_ERR_INVALID, "kk_block_update: W overlaps the basis range");
    gram_touch(bw, cw);
    return block_update_run(b->ctx, b->col(c0), b->ld, m, bw->col(cw), bw->ld, q, S, lds, alpha, beta, norms);
}

// block_reorthogonalize!(W, V) (blocklanczos.jl:277-284)
static int block_reorth_run(kk_basis b, int c0, int m, int cw, int q) {
    kk_ctx c = b->ctx;
    if (m == 0 || q == 0) return KK_OK;
    if (c->block_mode == 0) {  // strict: every W[i] swept against every basis vector, sequentially
        for (int i = 0; i < q; ++i) {
            for (int j0 = 0; j0 < m; j0 += KK_MAX_M) {
                const int mm = std::min(KK_MAX_M, m - j0);
                KK_TRY(pass_mgs_strict(c, b->col(c0 + j0), b->ld, mm, b->col(cw + i), WS_S, false, 0, nullptr, nullptr, false));
            }
        }
        return stream_sync(c);
    }
    // panel: P = V' W (MFMA), W -= V P.  Differs from the sequential sweep by L*P with L the
    // strictly-lower Gram matrix of V (O(eps)) -- far below roundoff of the update itself.
    std::vector<double> P((size_t)m * q);
    KK_TRY(block_inner_run(c, b->col(c0), b->ld, m, b->col(cw), b->ld, q, b->ld, P.data(), m));
    return block_update_run(c, b->col(c0), b->ld, m, b->col(cw), b->ld, q, P.data(), m, -1.0, 1.0, nullptr);
}

KK_API int kk_block_reorthogonalize(kk_basis b, int c0, int m, int cw, int q) {
    CHECK_BLOCK(b, c0, m); CHECK_BLOCK(b, cw, q);
    KK_CHECK(!(cw < c0 + m && c0 < cw + q), KK_ERR_INVALID, "kk_block_reorthogonalize: W overlaps the basis range");
    gram_touch(b, cw);
    return block_reorth_run(b, c0, m, cw, q);
}

// ---- small dense helpers for the CholQR2 fast path (p <= 64, host) -------------------------
// upper Cholesky G = R'R (column-major p x p); returns false if a pivot is not safely positive:
// pivot^2 must exceed rel^2 * G_jj and abs_min^2
static bool chol_upper_safe(const std::vector<double>& G, int p, std::vector<double>& R, double rel, double abs_min) {
    R.assign((size_t)p * p, 0.0);
    for (int j = 0; j < p; ++j) {
        for (int i = 0; i < j; ++i) {
            double t = G[i + (size_t)p * j];
            for (int k = 0; k < i; ++k) t -= R[k + (size_t)p * i] * R[k + (size_t)p * j];
            R[i + (size_t)p * j] = t / R[i + (size_t)p * i];
        }
        double d2 = G[j + (size_t)p * j];
        for (int k = 0; k < j; ++k) d2 -= R[k + (size_t)p * j] * R[k + (size_t)p * j];
        const double gjj = G[j + (size_t)p * j];
        if (!(d2 > rel * rel * gjj) || !(d2 > abs_min * abs_min) || !std::isfinite(d2)) return false;
        R[j + (size_t)p * j] = std::sqrt(d2);
    }
    return true;
}
static void triu_inverse(const std::vector<double>& R, int p, std::vector<double>& Ri) {
    Ri.assign((size_t)p * p, 0.0);
    for (int j = 0; j < p; ++j) {
        Ri[j + (size_t)p * j] = 1.0 / R[j + (size_t)p * j];
        for (int i = j - 1; i >= 0; --i) {
            double t = 0;
            for (int k = i + 1; k <= j; ++k) t += R[i + (size_t)p * k] * Ri[k + (size_t)p * j];
            Ri[i + (size_t)p * j] = -t / R[i + (size_t)p * i];
        }
    }
}

static int block_qr_strict(kk_basis b, int c_in, int p, int c_out, double tol, double* R, int ldr, int* good_idx, int* ngood,
                           int* is_drift);

// block_qr! (blocklanczos.jl:312-353).  Panel mode, out of place: CholQR2 on the MFMA Gram panel
//   G = B'B -> R1 = chol(G) -> Q1 = B R1^-1 -> G2 = Q1'Q1 -> R2 = chol(G2) -> Q = Q1 R2^-1, R = R2 R1
// (768 N bytes instead of ~240 p N for the column-by-column sweep).  It is taken only when every
// Cholesky pivot is safely away from the reference's rank / DGKS thresholds (beta_j > 1000 tol and
// beta_j > 1e-5 |b_j|), in which case the reference's block_qr! keeps every column and does not
// drift; otherwise the faithful column-by-column path below runs on the untouched input.
static int block_qr_run(kk_basis b, int c_in, int p, int c_out, double tol, double* R, int ldr, int* good_idx, int* ngood,
                        int* is_drift, const double* G_known = nullptr /* B'B (p x p, column-major) when the caller has it already */) {
    kk_ctx c = b->ctx;
    if (c->block_mode == 1 && c_out != c_in && p <= 64 && p >= 2) {
        const int64_t ld = b->ld;
        std::vector<double> G((size_t)p * p), R1, R2, Ri;
        if (G_known) G.assign(G_known, G_known + (size_t)p * p);
        else KK_TRY(block_inner_run(c, b->col(c_in), ld, p, b->col(c_in), ld, p, ld, G.data(), p));
        if (chol_upper_safe(G, p, R1, 1e-5, 1000.0 * tol)) {
            triu_inverse(R1, p, Ri);
            // Q1 = B * R1^-1 (out of place)
            for (int j0 = 0; j0 < p; j0 += 16) {
                const int nb = std::min(16, p - j0);
                const double* Sd = nullptr;
                KK_TRY(stage_coef(c, Ri.data(), p, p, j0, nb, &Sd));
                KK_TRY(kk_launch_block_update(c, b->col(c_in), ld, p, nullptr, b->col(c_out + j0), ld, ld, nb, Sd, 1.0, 0.0,
                                              nullptr));
            }
            KK_TRY(block_inner_run(c, b->col(c_out), ld, p, b->col(c_out), ld, p, ld, G.data(), p));
            double dev = 0;
            for (int j = 0; j < p; ++j)
                for (int i = 0; i < p; ++i) dev = std::max(dev, std::fabs(G[i + (size_t)p * j] - (i == j ? 1.0 : 0.0)));
            if (dev < 1e-3 && chol_upper_safe(G, p, R2, 1e-2, 0.0)) {
                triu_inverse(R2, p, Ri);
                // Q = Q1 * R2^-1: needs all p input columns per row before any write -> via scratch columns
                // is avoided by processing in ONE launch per 16 output columns reading the old values:
                // output columns j0.. only depend on input columns <= j0+15 (upper-triangular), and
                // are written after the kernel has read them (row-local), so go right-to-left.
                for (int j0 = ((p - 1) / 16) * 16; j0 >= 0; j0 -= 16) {
                    const int nb = std::min(16, p - j0);
                    const int mm = j0 + nb;  // rows of R2^-1 that can be non-zero for these columns
                    const double* Sd = nullptr;
                    KK_TRY(stage_coef(c, Ri.data(), p, mm, j0, nb, &Sd));
                    KK_TRY(kk_launch_block_update(c, b->col(c_out), ld, mm, nullptr, b->col(c_out + j0), ld, ld, nb, Sd, 1.0,
                                                  0.0, nullptr));
                }
                // R = R2 * R1
                for (int j = 0; j < p; ++j)
                    for (int i = 0; i < p; ++i) {
                        double t = 0;
                        for (int k = i; k <= j; ++k) t += R2[i + (size_t)p * k] * R1[k + (size_t)p * j];
                        R[i + (size_t)ldr * j] = (i <= j) ? t : 0.0;
                    }
                for (int j = 0; j < p; ++j) good_idx[j] = j;
                *ngood = p;
                if (is_drift) *is_drift = 0;
                return KK_OK;
            }
        }
    }
    return block_qr_strict(b, c_in, p, c_out, tol, R, ldr, good_idx, ngood, is_drift);
}

// faithful column-by-column MGS with DGKS and rank detection (blocklanczos.jl:312-353)
static int block_qr_strict(kk_basis b, int c_in, int p, int c_out, double tol, double* R, int ldr, int* good_idx, int* ngood,
                           int* is_drift) {
    kk_ctx c = b->ctx;
    const int64_t ld = b->ld;
    std::vector<double> Rf((size_t)p * p, 0.0);  // full p x p, column-major
    std::vector<char> idx(p, 1);
    bool drift = false;
    if (c_out != c_in)
        for (int j = 0; j < p; ++j) KK_TRY(kk_launch_copy_scal(c, b->col(c_out + j), b->col(c_in + j), ld, 1.0));
    double* Q = b->col(c_out);
    auto finish_col = [&](int j, double beta) -> int {
        if ((j == 0 && beta > tol) || (j > 0 && !(beta < tol))) {  // :319 uses beta > tol, :343 uses beta < tol
            Rf[j + (size_t)p * j] = beta;
            return kk_launch_scal(c, Q + (int64_t)j * ld, ld, 1.0 / beta, nullptr);
        }
        idx[j] = 0;
        KK_HIP(hipMemsetAsync(Q + (int64_t)j * ld, 0, ld * sizeof(double), c->stream));
        return KK_OK;
    };
    // column 1
    KK_TRY(kk_launch_nrm2(c, Q, ld, SCP(c, SC_NRM2)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
    KK_TRY(stream_sync(c));
    KK_TRY(finish_col(0, pin(c, WS_SCAL + SC_NRM2)[1]));
    for (int j = 1; j < p; ++j) {
        double* w = Q + (int64_t)j * ld;
        KK_TRY(pass_mgs_strict(c, Q, ld, j, w, WS_S, true, 0, nullptr, nullptr, false));  // first MGS :328-332
        KK_TRY(stream_sync(c));
        for (int i = 0; i < j; ++i) Rf[i + (size_t)p * j] = pin(c, WS_S)[i];
        double beta = pin(c, WS_SCAL + SC_NRM2)[1];
        if (tol < beta && beta < 100 * tol) {  // DGKS :334-342
            drift = true;
            KK_TRY(pass_mgs_strict(c, Q, ld, j, w, WS_S, true, 0, nullptr, nullptr, false));
            KK_TRY(stream_sync(c));
            for (int i = 0; i < j; ++i) Rf[i + (size_t)p * j] += pin(c, WS_S)[i];
            beta = pin(c, WS_SCAL + SC_NRM2)[1];
        }
        KK_TRY(finish_col(j, beta));
    }
    // compact the good vectors to the front (push!(V, R[good_idx]), blocklanczos.jl:219)
    int ng = 0;
    for (int j = 0; j < p; ++j) {
        if (!idx[j]) continue;
        if (ng != j) KK_TRY(kk_launch_copy_scal(c, Q + (int64_t)ng * ld, Q + (int64_t)j * ld, ld, 1.0));
        good_idx[ng] = j;
        for (int jj = 0; jj < p; ++jj) R[ng + (size_t)ldr * jj] = Rf[j + (size_t)p * jj];
        ++ng;
    }
    *ngood = ng;
    if (is_drift) *is_drift = drift ? 1 : 0;
    return KK_OK;
}

KK_API int kk_block_qr(kk_basis b, int c_in, int p, int c_out, double tol, double* R, int ldr, int* good_idx,
                           int* ngood, int* is_drift) {
    CHECK_BLOCK(b, c_in, p); CHECK_BLOCK(b, c_out, p);
    KK_CHECK(p >= 1 && p <= KK_MAX_M, KK_ERR_INVALID, "kk_block_qr: p=%d", p);
    KK_CHECK(R && good_idx && ngood && ldr >= p, KK_ERR_INVALID, "kk_block_qr: bad output arguments");
    KK_CHECK(c_out == c_in || c_out + p <= c_in || c_in + p <= c_out, KK_ERR_INVALID, "kk_block_qr: partial overlap");
    gram_touch(b, std::min(c_in, c_out));
    return block_qr_run(b, c_in, p, c_out, tol, R, ldr, good_idx, ngood, is_drift);
}

// device-side layout of the asynchronous block step inside the second half of the block scratch (doubles)
#define AB_BASE (KK_BLK_SCRATCH / 2 + 64)
#define AB_FLAG (AB_BASE)            // [4]   0 = fine, 1 / 2 = a CholQR2 safety test failed
#define AB_NRM (AB_BASE + 4)         // [16]  squared column norms of the new residual block
#define AB_B (AB_BASE + 20)          // [256] B = R2 R1, column-major ld 16
#define AB_M (AB_BASE + 276)         // [256] M = X' A X, column-major ld 16
#define AB_CF (AB_BASE + 532)        // [4]   commit flag of the step: 0 = the update wrote T = W R1^-1 (k_blk_commit_prep), else the plain block
#define AB_R1 (AB_BASE + 536)        // [256] first CholQR2 factor: of this step (k_blk_chol1) until k_blk_chol2 has read it, then of the NEXT step's commit
#define AB_READBACK 792              // flag + norms + B + M + commit flag + R1 travel to the host in ONE copy
#define AB_G (AB_BASE + 792)         // [256] Gram panels of the two CholQR2 rounds
#define AB_S1 (AB_BASE + 1048)       // [256] staged R1^-1
#define AB_S2 (AB_BASE + 1304)       // [256] staged R2^-1
#define AB_S3 (AB_BASE + 1560)       // [512] three-term panel [B' ; M]
#define AB_P (AB_BASE + 2072)        // [3 * KK_MAX_M * 16] re-orthogonalisation panel V'(AX), row-major; the one-pass step appends
                                     // the ride-along Gram panel V'X and the corrected panel (kn * st doubles each)
#define AB_GYY (AB_P + 3 * KK_MAX_M * 16)   // [256] (A X)'(A X), column-major ld 16 (one-pass step)
#define AB_GW (AB_GYY + 256)                // [256] Gram matrix of the residual block left behind, column-major ld p
#define AB_END (AB_GW + 256)
static_assert(AB_END <= KK_BLK_SCRATCH, "block scratch too small for the asynchronous block step");

// The one-pass recurrence of a block step, enqueued without a host round trip (used by expand! for k >= p and by initialize for
// k = 0, where the panel has no three-term rows and P = X1'(A X1) IS M1 -- blocklanczos.jl:181-192):
//   AX = A X (X = columns k .. k+p-1)  ->  P = V'(AX) against the whole basis [0, kn), Gram rows of the new block riding along
//   ->  Pc = (I - E) P  ->  W = AX - V Pc, written as the plain block, or (try_tc) as T = W R1^-1 into the next basis slot kn ..
// Leaves M in AB_M, the squared column norms in AB_NRM, the commit flag in AB_CF, R1 in AB_R1, T'T in AB_G, the residual Gram
// matrix in AB_GW, the safety flag in AB_FLAG, and copies the ride-along Gram panel to the pinned mirror.
static int onepass_enqueue(kk_op op, kk_basis b, int k, int p, int c_rnext, double qr_tol, bool try_tc) {
    kk_ctx c = b->ctx;
    const int64_t ld = b->ld;
    const int st = kk_bu_stride(p);
    double* D = c->blk;
    const int kn = k + p;
    double* AX = try_tc ? b->col(kn) : b->col(c_rnext);
    KK_TRY(kk_launch_spmm(c, op->A, b->col(k), ld, AX, ld, p));
    // one pass: P = V'(A X) against the whole basis, AX -= V P.  Rows k-p .. k+p-1 of P are the three-term coefficients
    // [B' ; M] (blocklanczos.jl:253-260), the other rows the re-orthogonalisation (:277-284); the reference subtracts the
    // three-term part first and projects the remainder once more ("twice is enough").  A single classical pass is NOT:
    // the orthogonality error E = V'V - I re-enters as E P and grows geometrically.  Here E is known -- the Gram rows of
    // every new block ride along in the panel kernel -- and the coefficients are corrected to first order,
    // P <- (I - E) P (k_blk_panel_correct), which leaves V'w = O(E^2 |P|) like the second pass does.  What remains is the
    // rounding of the panel itself, eps |A x_j| instead of eps |w_j|: a column that loses more than a factor 10 of its
    // norm raises flag 3 and the step is repeated on the two-pass route.  Saves the M panel, and one read + one write
    // of [Xprev X AX] per step.
    double* P = D + AB_P;
    double* G2 = P + (int64_t)kn * st;
    double* Pc = G2 + (int64_t)kn * st;
    // two accumulator sets: 80 basis columns per launch keep the kernel at two waves per SIMD (128 columns: one)
    const int chunk = c->gram2_chunk;
    const int nch = (kn + chunk - 1) / chunk;
    const int per = ((kn + nch - 1) / nch + 15) / 16 * 16;   // balanced chunks, whole 16-column groups
    const bool want_gw = c->resid_gram != 0;
    for (int i0 = 0; i0 < kn; i0 += per)
        KK_TRY(kk_launch_block_gram2(c, b->col(i0), ld, std::min(per, kn - i0), AX, ld, p, b->col(k), ld, p, ld, P + (int64_t)i0 * st,
                                     st, G2 + (int64_t)i0 * st, st, (want_gw && i0 == 0) ? D + AB_GYY : nullptr));
    KK_TRY(kk_allreduce(c, P, 2 * (int64_t)kn * st));
    if (want_gw) KK_TRY(kk_allreduce(c, D + AB_GYY, 256));
    KK_TRY(kk_launch_blk_gram_rows(c, G2, st, k, p, b->d_gram, b->cap, b->d_gdiag));
    KK_TRY(kk_launch_blk_panel_m(c, P, st, k, p, D + AB_M, 16));
    KK_TRY(kk_launch_blk_panel_correct(c, P, st, kn, p, b->d_gram, b->cap, Pc, b->d_gdiag));
    if (try_tc) {
        // first CholQR2 factor of the NEXT step from the predicted Gram matrix (AX)'(AX) - P'Pc, then the update writes
        // T = W R1^-1 into columns kn.. and accumulates T'T (AB_G); the residual area keeps A X
        KK_TRY(kk_launch_blk_resid_gram(c, P, Pc, st, kn, p, D + AB_GYY, nullptr, D + AB_GW));
        KK_TRY(kk_launch_blk_commit_prep(c, D + AB_GW, D + AB_GYY, p, 1000.0 * qr_tol, 1e-3, D + AB_R1, D + AB_S1, st, D + AB_CF));
        KK_TRY(kk_launch_block_update_commit(c, b->col(0), ld, kn, AX, AX, ld, AX, ld, p, Pc, D + AB_NRM, D + AB_CF, D + AB_S1,
                                             D + AB_G));
    } else {
        KK_TRY(kk_launch_block_update(c, b->col(0), ld, kn, AX, AX, ld, ld, p, Pc, -1.0, 1.0, D + AB_NRM));
    }
    KK_TRY(kk_launch_blk_onepass_check(c, P, st, kn, p, D + AB_NRM, 0.1, D + AB_FLAG));
    if (want_gw) KK_TRY(kk_launch_blk_resid_gram(c, P, Pc, st, kn, p, D + AB_GYY, D + AB_NRM, D + AB_GW));
    KK_HIP(hipMemcpyAsync(c->h_blk + (G2 - D), G2, (size_t)kn * st * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return KK_OK;
}

KK_API int kk_blocklanczos_initialize(kk_op op, kk_basis b, int c_x0, int bs0, int c_r, double qr_tol, int* bs,
                                          double* M1, int ldm, double* norm_R) {
    KK_TRY(check_square_op(op, b));
    CHECK_BLOCK(b, c_x0, bs0); CHECK_BLOCK(b, c_r, bs0); CHECK_BLOCK(b, 0, bs0);
    KK_CHECK(bs0 >= 1 && bs && M1 && norm_R && ldm >= bs0, KK_ERR_INVALID, "kk_blocklanczos_initialize: bad arguments");
    KK_CHECK(c_r >= bs0 && (c_x0 == 0 || c_x0 >= bs0) && !(c_x0 < c_r + bs0 && c_r < c_x0 + bs0), KK_ERR_INVALID,
             "kk_blocklanczos_initialize: column ranges overlap");
    kk_ctx c = b->ctx;
    gram_touch(b, 0);
    // beta0 = norm(X0) (Frobenius) must not vanish  :168-169
    std::vector<double> G((size_t)bs0 * bs0), R((size_t)bs0 * bs0);
    std::vector<int> good(bs0);
    double n2 = 0;
    if (c->block_mode == 1 && bs0 >= 2 && bs0 <= 64) {
        // |X0|_F^2 = trace(X0' X0): ONE Gram launch and one host round trip instead of bs0 of each (the 16 k_dot +
        // 16 synchronisations of a bs = 16 initialize).  The figure only decides whether the start block vanishes.
        KK_TRY(block_inner_run(c, b->col(c_x0), b->ld, bs0, b->col(c_x0), b->ld, bs0, b->ld, G.data(), bs0));
        for (int j = 0; j < bs0; ++j) n2 += G[j + (size_t)bs0 * j];
    } else {
        for (int j = 0; j < bs0; ++j) {
            KK_TRY(kk_launch_nrm2(c, b->col(c_x0 + j), b->ld, SCP(c, SC_NRM2)));
            KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 1, 0));
            KK_TRY(stream_sync(c));
            n2 += pin(c, WS_SCAL + SC_NRM2)[0];
        }
    }
    if (n2 == 0.0) {
        kk_set_error("initial vector should not have norm zero");
        return KK_ERR_ZERO_NORM;
    }
    int ng = 0, drift = 0;
    const bool have_G = c->block_mode == 1 && bs0 >= 2 && bs0 <= 64;
    KK_TRY(block_qr_run(b, c_x0, bs0, 0, qr_tol, R.data(), bs0, good.data(), &ng, &drift, have_G ? G.data() : nullptr));  // X1 = block_qr!(X0)[good]  :175-177
    KK_CHECK(ng >= 1, KK_ERR_ZERO_NORM, "kk_blocklanczos_initialize: start block has numerical rank 0");
    // Round 6: the recurrence of initialize in the ONE-PASS form of the block steps, ending in the same normalised
    // commit -- AX1 formed in the next basis slot, P = X1'(A X1) (= M1, with the Gram rows of X1 riding along), the residual written as
    // T = W R1^-1 with T'T accumulated.  The first expand! then finds its first CholQR2 round done, as every later one does: no Gram pass
    // over the residual block, no Q1 = W R1^-1 pass (0.8 ms of the 10 M-row sweep), and initialize itself loses the separate
    // block_inner / update pair.  Same conditions as the commit of a block step; anything else -- and a raised safety flag -- takes the
    // synchronous route below on the untouched X1.
    {
        const int p = ng, st = kk_bu_stride(ng), kn = ng;
        const bool can = c->block_mode == 1 && c->block_async && (c->block_fuse & 4) && c->block_commit && c->resid_gram && ng == bs0 && ng >= 2 &&
                         ng <= 16 && !drift && b->tc_skip == 0 && !b->tc_valid && 2 * ng <= c_r && 2 * ng <= c_x0 && 2 * ng <= b->cap &&
                         ((size_t)kn * st + 64 + st * st + 4 * 16 * 34) * sizeof(double) <= 64 * 1024;
        if (can) {
            double* D = c->blk;
            b->gram_c0 = 0; b->gram_rows = 0;
            KK_TRY(gram_device(b));
            if (st > p) KK_HIP(hipMemsetAsync(D + AB_P, 0, (size_t)3 * kn * st * sizeof(double), c->stream));   // pad columns of the row-major panels
            KK_HIP(hipMemsetAsync(D + AB_FLAG, 0, 4 * sizeof(double), c->stream));
            c->tc_owner = 0;
            c->gw_valid = false;
            KK_TRY(onepass_enqueue(op, b, 0, p, c_r, qr_tol, true));
            KK_HIP(hipMemcpyAsync(c->h_blk + AB_BASE, D + AB_BASE, AB_READBACK * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            KK_TRY(stream_sync(c));
            const double* H = c->h_blk + AB_BASE;
            if (H[0] == 0.0) {
                if (H[AB_CF - AB_BASE] != 0.0) {   // not committed (decided on the device): the plain residual block sits in the basis slot
                    KK_HIP(hipMemcpyAsync(b->col(c_r), b->col(kn), (size_t)p * b->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                    c->gw_valid = true; c->gw_basis = b->uid; c->gw_col = c_r; c->gw_p = p;
                } else {                           // committed: columns kn .. hold T, the residual area is formed on demand (blk_commit_flush)
                    b->tc_valid = true; b->tc_k = kn; b->tc_cr = c_r; b->tc_p = p;
                    memcpy(b->tc_R1, H + (AB_R1 - AB_BASE), (size_t)p * p * sizeof(double));
                    c->tc_owner = b->uid;
                }
                const double* G2h = c->h_blk + AB_P + (int64_t)kn * st;   // host mirror of the Gram rows of X1
                for (int i = 0; i < p; ++i)
                    for (int j = 0; j < i; ++j) b->gram[(size_t)i * b->cap + j] = G2h[(size_t)j * st + i];
                b->gram_rows = kn;
                for (int j = 0; j < p; ++j)
                    for (int i = 0; i < p; ++i) M1[i + (size_t)ldm * j] = H[276 + i + 16 * j];
                double f2 = 0;
                for (int j = 0; j < p; ++j) f2 += H[4 + j];
                *norm_R = std::sqrt(f2);
                *bs = ng;
                return KK_OK;
            }
            b->gram_rows = 0;   // (flag raised: the panel of this attempt is not the basis' Gram record)
        }
    }
    // AX1 = A X1 ; M1 = block_inner(X1, AX1) ; AX1[j] -= X1[i] M1[i,j]   :181-192
    KK_TRY(kk_launch_spmm(c, op->A, b->col(0), b->ld, b->col(c_r), b->ld, ng));
    KK_TRY(block_inner_run(c, b->col(0), b->ld, ng, b->col(c_r), b->ld, ng, b->ld, M1, ldm));
    std::vector<double> nr(ng);
    KK_TRY(block_update_run(c, b->col(0), b->ld, ng, b->col(c_r), b->ld, ng, M1, ldm, -1.0, 1.0, nr.data()));
    double f = 0;
    for (int j = 0; j < ng; ++j) f += nr[j] * nr[j];
    *norm_R = std::sqrt(f);
    *bs = ng;
    return KK_OK;
}

// settle a pending normalised commit: W = T R1 into the residual area (kk_host.h norm_flush calls this from every entry
// point that lets a caller see or change the slab)
int blk_commit_flush(kk_basis b) {
    if (!b || !b->tc_valid) return KK_OK;
    b->tc_valid = false;
    if (b->ctx->tc_owner == b->uid) b->ctx->tc_owner = 0;
    // not consumed.  Once is normal (a restart looks at the block); twice in a row means the caller looks after every step:
    // pause the commits, longer each time, until one is consumed again
    if (b->tc_streak >= 1) b->tc_skip = 2 << std::min(b->tc_streak, 5);
    ++b->tc_streak;
    const int p = b->tc_p;
    return block_update_run(b->ctx, b->col(b->tc_k), b->ld, p, b->col(b->tc_cr), b->ld, p, b->tc_R1, p, 1.0, 0.0, nullptr);
}

// expand!(::BlockLanczosIterator) without a host round trip between its kernels (panel mode, 2 <= block size <= 16, no rank
// drop, no DGKS drift): CholQR2 of the residual block with both Cholesky factorisations, the triangular inverses and
// every coefficient panel formed ON THE DEVICE, the Gram panels written straight in the layout the update kernel reads;
// ONE host synchronisation at the end returns B, M, the column norms and the safety flag.  The input block (c_r) and the
// basis are not modified, so a raised flag (*fine = false) simply sends the caller to the synchronous route below.
static int blocklanczos_expand_async(kk_op op, kk_basis b, int k, int p, int c_r, int c_rnext, double qr_tol, double* B,
                                     int ldb, double* M, int ldm, double* norm_R, bool* fine, bool use_gw, bool take_tc, int* qr_state) {
    kk_ctx c = b->ctx;
    const int64_t ld = b->ld;
    const int st = kk_bu_stride(p);
    double* D = c->blk;
    const int kn = k + p;
    KK_CHECK(kn <= KK_MAX_M, KK_ERR_UNSUPPORTED, "block step: %d basis vectors exceed the panel limit", kn);
    const bool onepass = (c->block_fuse & 4) != 0;
    // normalised commit (one-pass step): the next basis slot must be free (it is not on the last step before a restart) and
    // the panels small enough for the update kernel's LDS.  A X is then formed IN that slot and updated in place (the
    // read-modify-write of one 16-column area keeps the DRAM pages of the update's read and write streams together)
    const int st0 = kk_bu_stride(p);
    const bool tc_pause = b->tc_skip > 0;
    if (tc_pause) --b->tc_skip;
    const bool try_tc = !tc_pause && onepass && c->block_commit && c->resid_gram && kn + p <= std::min(c_r, c_rnext) && kn + p <= b->cap &&
                        ((size_t)kn * st0 + 64 + st0 * st0 + 4 * 16 * 34) * sizeof(double) <= 64 * 1024;
    if (onepass) {   // Gram rows of the basis columns [0, k): known from the previous steps, recomputed after a restart
        if (b->gram_c0 != 0) { b->gram_c0 = 0; b->gram_rows = 0; }
        KK_TRY(gram_device(b));
        if (b->gram_rows < k) KK_TRY(gram_ensure(b, k));
    }
    KK_HIP(hipMemsetAsync(D + AB_FLAG, 0, 4 * sizeof(double), c->stream));
    if (!take_tc) c->tc_owner = 0;   // this step rewrites the scratch a pending commit of another slab would need
    // ---- block_qr! as CholQR2, out of place: residual block (c_r) -> new basis block (columns k..k+p-1)
    // (after a normalised commit of the previous step columns k..k+p-1 hold T = W R1^-1 already, AB_R1 its factor and AB_G the
    //  not yet all-reduced Gram matrix T'T: the first CholQR2 round costs nothing)
    if (!take_tc) {
        if (use_gw) {   // the previous step left the Gram matrix of this very block behind (all-reduced already)
            KK_HIP(hipMemcpyAsync(D + AB_G, D + AB_GW, (size_t)p * p * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        } else {
            KK_TRY(kk_launch_block_gram(c, b->col(c_r), ld, p, b->col(c_r), ld, p, ld, D + AB_G, p));
            KK_TRY(kk_allreduce(c, D + AB_G, (int64_t)p * p));
        }
        KK_TRY(kk_launch_blk_chol1(c, D + AB_G, p, 1000.0 * qr_tol, D + AB_R1, D + AB_S1, st, D + AB_FLAG));
        if (c->block_fuse & 1) {   // Q1 = B R1^-1 written and G2 = Q1'Q1 accumulated in ONE pass over the block
            KK_TRY(kk_launch_block_gram_tile(c, nullptr, 0, p, nullptr, 0, b->col(c_r), ld, p, D + AB_S1, st, 1.0, 0.0, b->col(k), ld, p,
                                             ld, D + AB_G, 1, p));
        } else {
            KK_TRY(kk_launch_block_update(c, b->col(c_r), ld, p, nullptr, b->col(k), ld, ld, p, D + AB_S1, 1.0, 0.0, nullptr));
            KK_TRY(kk_launch_block_gram(c, b->col(k), ld, p, b->col(k), ld, p, ld, D + AB_G, p));
        }
    }
    KK_TRY(kk_allreduce(c, D + AB_G, (int64_t)p * p));
    KK_TRY(kk_launch_blk_chol2(c, D + AB_G, p, D + AB_R1, D + AB_B, 16, D + AB_S2, D + AB_S3, st, D + AB_FLAG));
    // Q = Q1 R2^-1 in place (row-local); not executed when k_blk_chol2 found Q1 orthonormal already (device flag)
    KK_TRY(kk_launch_block_update(c, b->col(k), ld, p, nullptr, b->col(k), ld, ld, p, D + AB_S2, 1.0, 0.0, nullptr, D + AB_FLAG + 1));
    // ---- block_lanczosrecurrence: AX = A X ; M = X' AX ; AX -= [Xprev X] [B' ; M]
    if (onepass) {
        KK_TRY(onepass_enqueue(op, b, k, p, c_rnext, qr_tol, try_tc));
    } else {
    double* AX = b->col(c_rnext);
    KK_TRY(kk_launch_spmm(c, op->A, b->col(k), ld, AX, ld, p));
    KK_TRY(kk_launch_block_gram(c, b->col(k), ld, p, AX, ld, p, ld, D + AB_M, 16));
    KK_TRY(kk_allreduce(c, D + AB_M, 256));
    KK_TRY(kk_launch_blk_fill_m(c, D + AB_M, 16, p, D + AB_S3, st));
    if (c->block_fuse & 2) {
        // P = V'(AX - [Xprev X] S3) with the three-term result formed on the fly (never written); the update below then
        // subtracts V (P + [0; S3]) from the original AX:  AX - [Xprev X] S3 - V P
        for (int i0 = 0; i0 < kn; i0 += 128)
            KK_TRY(kk_launch_block_gram_tile(c, b->col(i0), ld, std::min(128, kn - i0), AX, ld, b->col(k - p), ld, 2 * p, D + AB_S3, st,
                                             -1.0, 1.0, nullptr, 0, p, ld, D + AB_P + (int64_t)i0 * st, st, 1));
        KK_TRY(kk_allreduce(c, D + AB_P, (int64_t)kn * st));
        KK_TRY(kk_launch_blk_combine(c, D + AB_P, D + AB_S3, kn, 2 * p, st));
    } else {
        KK_TRY(kk_launch_block_update(c, b->col(k - p), ld, 2 * p, AX, AX, ld, ld, p, D + AB_S3, -1.0, 1.0, nullptr));
        // ---- block_reorthogonalize!(AX, V): P = V' AX written row-major = the update's coefficient panel; fused column norms
        for (int i0 = 0; i0 < kn; i0 += 128)
            KK_TRY(kk_launch_block_gram_rs(c, b->col(i0), ld, std::min(128, kn - i0), AX, ld, p, ld, D + AB_P + (int64_t)i0 * st, st, 1));
        KK_TRY(kk_allreduce(c, D + AB_P, (int64_t)kn * st));
    }
    KK_TRY(kk_launch_block_update(c, b->col(0), ld, kn, AX, AX, ld, ld, p, D + AB_P, -1.0, 1.0, D + AB_NRM));
    }
    // ---- the one read-back
    KK_HIP(hipMemcpyAsync(c->h_blk + AB_BASE, D + AB_BASE, AB_READBACK * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    KK_TRY(stream_sync(c));
    const double* H = c->h_blk + AB_BASE;
    c->last_qr_dev = H[2];
    *fine = (H[0] == 0.0);
    // where the block in columns k.. stands if this step has to be repeated: 0 = T (second round not applied), 1 = Q = T R2^-1
    if (qr_state) *qr_state = (H[0] == 2.0 || H[1] != 0.0) ? 0 : 1;
    if (!*fine) return KK_OK;
    if (take_tc) { c->block_commits++; b->tc_streak = 0; }
    if (try_tc && H[AB_CF - AB_BASE] != 0.0)   // not committed (decided on the device): the plain block sits in the basis slot
        KK_HIP(hipMemcpyAsync(b->col(c_rnext), b->col(kn), (size_t)p * ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (try_tc && H[AB_CF - AB_BASE] == 0.0) {   // committed: columns kn.. hold T, the residual area is not written at all
        b->tc_valid = true; b->tc_k = kn; b->tc_cr = c_rnext; b->tc_p = p;
        memcpy(b->tc_R1, H + (AB_R1 - AB_BASE), (size_t)p * p * sizeof(double));
        c->tc_owner = b->uid;
    } else if (onepass && c->resid_gram) {   // the residual block in c_rnext now has its Gram matrix in AB_GW
        c->gw_valid = true; c->gw_basis = b->uid; c->gw_col = c_rnext; c->gw_p = p;
    }
    if (onepass) {   // host mirror of the new Gram rows (strictly-lower storage), as the device kernel wrote them
        const double* G2h = c->h_blk + AB_P + (int64_t)kn * st;
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < k + i; ++j) b->gram[(size_t)(k + i) * b->cap + j] = G2h[(size_t)j * st + i];
        b->gram_rows = kn;
    }
    for (int j = 0; j < p; ++j)
        for (int i = 0; i < p; ++i) {
            B[i + (size_t)ldb * j] = H[20 + i + 16 * j];
            M[i + (size_t)ldm * j] = H[276 + i + 16 * j];
        }
    double f = 0;
    for (int j = 0; j < p; ++j) f += H[4 + j];
    *norm_R = std::sqrt(f);
    return KK_OK;
}

KK_API int kk_blocklanczos_expand(kk_op op, kk_basis b, int k, int bs_r, int c_r, int c_rnext, double qr_tol,
                                      int* bs_next, double* B, int ldb, double* M, int ldm, double* norm_R, int* is_drift) {
    KK_TRY(check_square_op(op, b));
    // a pending normalised commit of exactly this residual block is consumed here; anything else settles it first (CHECK_*)
    bool take_tc = false;
    if (b && b->tc_valid) {
        kk_ctx cc = b->ctx;
        take_tc = cc->tc_owner == b->uid && b->tc_k == k && b->tc_cr == c_r && b->tc_p == bs_r && cc->block_mode == 1 && cc->block_async &&
                  (cc->block_fuse & 4) && cc->block_commit && bs_r >= 2 && bs_r <= 16 && k + bs_r <= KK_MAX_M;
    }
    // argument validation FIRST, without side effects: an early error return must leave a pending commit pending (ADVICE r4 --
    // with the flag cleared before the checks a rejected call lost the commit: the residual area still held A X and nothing
    // formed W = T R1 afterwards)
    KK_CHECK(b && k >= 0 && bs_r >= 0 && k + bs_r <= b->cap && c_r >= 0 && c_r + bs_r <= b->cap && c_rnext >= 0 && c_rnext + bs_r <= b->cap,
             KK_ERR_INVALID, "kk_blocklanczos_expand: block [%d,%d), [%d,%d) or [%d,%d) outside capacity %d", 0, k + bs_r, c_r, c_r + bs_r,
             c_rnext, c_rnext + bs_r, b ? b->cap : 0);
    KK_CHECK(k >= bs_r && bs_r >= 1 && bs_next && B && M && norm_R && ldb >= bs_r && ldm >= bs_r, KK_ERR_INVALID,
             "kk_blocklanczos_expand: bad arguments");
    KK_CHECK(c_r >= k + bs_r && c_rnext >= k + bs_r && (c_rnext + bs_r <= c_r || c_r + bs_r <= c_rnext), KK_ERR_INVALID,
             "kk_blocklanczos_expand: residual blocks must lie beyond column k+bs_r and not overlap");
    if (take_tc) b->tc_valid = false;   // consumed by this call (every check has passed)
    KK_TRY(norm_flush(b));              // anything else that is pending on the slab is settled first
    kk_ctx c = b->ctx;
    // cached Gram matrix of the incoming residual block (read before gram_touch, which drops it): only this very block,
    // untouched since the step that produced it
    const bool use_gw = c->gw_valid && c->resid_gram && (c->block_fuse & 4) && c->gw_basis == b->uid && c->gw_col == c_r && c->gw_p == bs_r;
    c->gw_valid = false;
    gram_touch(b, k);
    if (c->block_mode == 1 && c->block_async && bs_r >= 2 && bs_r <= 16 && k + bs_r <= KK_MAX_M) {
        bool fine = false;
        if (kk_bu_stride(bs_r) > bs_r)   // pad columns of the row-major panels: zero once, the kernels write only the first bs_r
            KK_HIP(hipMemsetAsync(c->blk + AB_P, 0, (size_t)((c->block_fuse & 4) ? 3 : 1) * (k + bs_r) * kk_bu_stride(bs_r) * sizeof(double),
                                  c->stream));   // one-pass mode keeps three panels (P, G2, Pc) back to back
        int qr_state = 0;
        KK_TRY(blocklanczos_expand_async(op, b, k, bs_r, c_r, c_rnext, qr_tol, B, ldb, M, ldm, norm_R, &fine, use_gw, take_tc, &qr_state));
        if (fine) {
            *bs_next = bs_r;
            if (is_drift) *is_drift = 0;
            return KK_OK;
        }   // else: a pivot came close to the rank / DGKS thresholds -- the faithful route decides (inputs untouched)
        if (take_tc) {
            // ... except that after a normalised commit the input block exists only as T (or Q = T R2^-1) in columns k..:
            // W = T R1 = Q (R2 R1) goes back into its residual area first (B holds R2 R1 when the second round was applied)
            std::vector<double> X((size_t)bs_r * bs_r);
            for (int j = 0; j < bs_r; ++j)
                for (int i = 0; i < bs_r; ++i) X[i + (size_t)bs_r * j] = qr_state ? c->h_blk[AB_B + i + 16 * j] : b->tc_R1[i + (size_t)bs_r * j];
            KK_TRY(block_update_run(c, b->col(k), b->ld, bs_r, b->col(c_r), b->ld, bs_r, X.data(), bs_r, 1.0, 0.0, nullptr));
            take_tc = false;
        }
    } else if (take_tc) {   // (cannot happen: take_tc implies the asynchronous route)
        b->tc_valid = true;
        KK_TRY(blk_commit_flush(b));
    }
    std::vector<int> good(bs_r);
    int ng = 0, drift = 0;
    // B, good_idx, is_drift = block_qr!(R, qr_tol); out of place: the input block stays intact as Rcopy   :209-211
    KK_TRY(block_qr_run(b, c_r, bs_r, k, qr_tol, B, ldb, good.data(), &ng, &drift));
    const int drift_first = drift;
    if (drift) {  // :212-216
        KK_TRY(block_reorth_run(b, 0, k, k, ng));
        std::vector<double> R2((size_t)ng * ng);
        std::vector<int> good2(ng);
        int ng2 = 0, d2 = 0;
        KK_TRY(block_qr_run(b, k, ng, k, qr_tol, R2.data(), ng, good2.data(), &ng2, &d2));
        ng = ng2;
        KK_TRY(block_inner_run(c, b->col(k), b->ld, ng, b->col(c_r), b->ld, bs_r, b->ld, B, ldb));  // B = block_inner(R[good], Rcopy)
    }
    KK_CHECK(ng >= 1, KK_ERR_ZERO_NORM, "kk_blocklanczos_expand: residual block has numerical rank 0 (invariant subspace)");
    const int kn = k + ng;
    // block_lanczosrecurrence :242-263 : AX = A X ; M = block_inner(X, AX)
    double* AX = b->col(c_rnext);
    KK_TRY(kk_launch_spmm(c, op->A, b->col(k), b->ld, AX, b->ld, ng));
    KK_TRY(block_inner_run(c, b->col(k), b->ld, ng, AX, b->ld, ng, b->ld, M, ldm));
    // AX[j] -= sum_i X[i] M[i,j] + sum_i Xprev[i] conj(B[j,i]): one update over the contiguous [Xprev | X]
    {
        const int mm = bs_r + ng;
        std::vector<double> S((size_t)mm * ng);
        for (int j = 0; j < ng; ++j) {
            for (int i = 0; i < bs_r; ++i) S[i + (size_t)mm * j] = B[j + (size_t)ldb * i];
            for (int i = 0; i < ng; ++i) S[bs_r + i + (size_t)mm * j] = M[i + (size_t)ldm * j];
        }
        KK_TRY(block_update_run(c, b->col(k - bs_r), b->ld, mm, AX, b->ld, ng, S.data(), mm, -1.0, 1.0, nullptr));
    }
    // block_reorthogonalize!(AX, V) with fused Frobenius norm of the result
    std::vector<double> nr(ng);
    if (c->block_mode == 0) {
        KK_TRY(block_reorth_run(b, 0, kn, c_rnext, ng));
        double f = 0;
        for (int j = 0; j < ng; ++j) {
            KK_TRY(kk_launch_nrm2(c, AX + (int64_t)j * b->ld, b->ld, SCP(c, SC_NRM2)));
            KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 1, 0));
            KK_TRY(stream_sync(c));
            f += pin(c, WS_SCAL + SC_NRM2)[0];
        }
        *norm_R = std::sqrt(f);
    } else {
        std::vector<double> P((size_t)kn * ng);
        KK_TRY(block_inner_run(c, b->col(0), b->ld, kn, AX, b->ld, ng, b->ld, P.data(), kn));
        KK_TRY(block_update_run(c, b->col(0), b->ld, kn, AX, b->ld, ng, P.data(), kn, -1.0, 1.0, nr.data()));
        double f = 0;
        for (int j = 0; j < ng; ++j) f += nr[j] * nr[j];
        *norm_R = std::sqrt(f);
    }
    *bs_next = ng;
    if (is_drift) *is_drift = drift_first;
    return KK_OK;
}

