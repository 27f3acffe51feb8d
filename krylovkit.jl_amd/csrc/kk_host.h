// Host-side internals shared by the C-ABI translation units of libkrylov_hip.so (kk_context / kk_sparse / kk_orth /
// kk_krylov / kk_block / kk_solvers / kk_dev .hip).  Nothing here is exported: the library is built with
// -fvisibility=hidden and only the KK_API entry points of include/krylov_hip.h are visible.
#pragma once
#include "kk_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>

#define KK_API extern "C" __attribute__((visibility("default")))

static const double KK_EPS = std::numeric_limits<double>::epsilon();
#define WSP(c, off) ((c)->ws + (off))
#define SCP(c, slot) ((c)->ws + WS_SCAL + (slot))

// ---- argument checks (status + message, never an exception across the C boundary).  A check of a basis argument also
// settles a pending normalised residual WHEN THE CHECKED COLUMNS INCLUDE IT (norm_flush_range below; a pending block commit is
// settled by every check): only the expand! that consumes it -- and scale!!(r, 1 / beta) of a restart, kk_vec_scal /
// kk_vec_copy_scal -- look at the flag themselves.  An entry point that touches a column declares it through one of these
// macros; the fused L3 steps and kk_basis_info (raw pointer) call norm_flush for the whole slab.
// Every PUBLIC entry point other than the expand! steps also voids what was enqueued AHEAD for the context's running factorization
// (ctx_public_touch: the speculative apply, a whole step): its kernels may write the shared device scalars the step in flight
// still has to read -- kk_vec_nrm2 on a basis column of the owner slab overwrites |w| and 1 / |w| -- and the owner's
// next expand! then simply redoes the apply and the sweep from the slab.  CHECK_COL_RO: entry points that launch no kernel
// writing the scalar workspace (downloads, gathers) leave the run-ahead alone.
#define CHECK_COL_BOUNDS(b, c) KK_CHECK((b) && (c) >= 0 && (c) < (b)->cap, KK_ERR_INVALID, "%s: column %d out of range", __func__, (c))
#define CHECK_COL_RO(b, c) do { CHECK_COL_BOUNDS(b, c); KK_TRY(norm_flush_range(b, (c), 1)); } while (0)
#define CHECK_COL(b, c) do { CHECK_COL_RO(b, c); ctx_public_touch(b); } while (0)
#define CHECK_SAME(bx, by) KK_CHECK((bx)->ctx == (by)->ctx && (bx)->n == (by)->n && (bx)->ld == (by)->ld, KK_ERR_DIM, "%s: vector length mismatch (%lld vs %lld)", __func__, (long long)(bx)->n, (long long)(by)->n)
#define CHECK_RANGE(b, c0, m) do { KK_CHECK((b) && (c0) >= 0 && (m) >= 0 && (c0) + (m) <= (b)->cap && (m) <= KK_MAX_M, KK_ERR_INVALID, "%s: column range [%d,%d) invalid (capacity %d, max %d per call)", __func__, (c0), (c0) + (m), (b) ? (b)->cap : 0, KK_MAX_M); KK_TRY(norm_flush_range(b, (c0), (m))); ctx_public_touch(b); } while (0)
// (CHECK_RANGE: one kernel panel, m <= KK_MAX_M; CHECK_BLOCK: any number of columns -- the entry point goes panel by panel)
#define CHECK_BLOCK(b, c0, p) do { KK_CHECK((b) && (c0) >= 0 && (p) >= 0 && (c0) + (p) <= (b)->cap, KK_ERR_INVALID, "%s: block [%d,%d) outside capacity %d", __func__, (c0), (c0) + (p), (b) ? (b)->cap : 0); KK_TRY(norm_flush_range(b, (c0), (p))); ctx_public_touch(b); } while (0)

// ---- scalar read-backs (kk_context.hip): results of the finalize kernels travel through the pinned mirror of the
// scalar workspace; `slot` selects one of its 4 copies
int ws_fetch_async(kk_ctx c, int64_t off, int64_t count, int slot);
int stream_sync(kk_ctx c);
static inline const double* pin(kk_ctx c, int64_t off, int slot = 0) { return c->h_pin + (int64_t)slot * WS_TOTAL + off; }
// any mutation of a slab column: invalidates the cached Gram rows from that column on and a speculative next-step SpMV
static inline void gram_touch(kk_basis b, int col) {
    if (col < b->gram_rows) b->gram_rows = col;
    b->spec_valid = false;
    b->la_valid = false;
    kk_ctx c = b->ctx;   // cached Gram matrix of a residual block: gone as soon as a column at or below its end may have changed
    if (c->gw_valid && c->gw_basis == b->uid && col < c->gw_col + c->gw_p) c->gw_valid = false;
}
// A fused expand! whose sweep ran through the persistent kernel leaves the residual column NORMALISED (w / |w| written at the
// kernel's commit, SURVEY a7: the scale!!(r, 1/beta) of factorizations/lanczos.jl:257 / arnoldi.jl:209 costs no pass of its
// own) and notes (column, beta) on the slab.  The next expand! of the same factorization takes the column as its new basis
// vector; anything else that looks at the slab first gets r = beta * column back (residual(F), shrink!, restarts).
int blk_commit_flush(kk_basis b);   // kk_block.hip: W = T R1 into the residual area of a pending block commit
// Work enqueued AHEAD for one factorization (the speculative apply, a whole step) leaves state in the context's shared scalar
// workspace that the owner's NEXT call relies on: alpha0 of the apply, |w| / 1 / |w| of the step in flight.  Any entry point
// that is handed ANOTHER slab of the same context may launch kernels that write those scalars (a second factorization stepping
// in turns, an un-fused FunctionOperator run, a solver): the owner's run-ahead is dropped -- its next call redoes the apply and the
// step from the slab, which is intact (round 5: two interleaved Lanczos runs on one context returned the other run's beta).
// (the owner is only ever COMPARED, never dereferenced -- its slab may have been freed: a generation counter of the context carries the
//  verdict, the owner recorded the generation it speculated in and finds it moved on)
static inline void ctx_foreign_touch(kk_basis b) {
    if (!b) return;
    kk_ctx c = b->ctx;
    if (c->spec_owner != b) ++c->foreign_gen;
}
static inline void ctx_public_touch(kk_basis b) { ++b->ctx->foreign_gen; }
static inline int norm_flush(kk_basis b) {
    if (!b) return KK_OK;
    ctx_foreign_touch(b);
    if (b->tc_valid) KK_TRY(blk_commit_flush(b));
    if (b->norm_col < 0) return KK_OK;
    const int col = b->norm_col;
    b->norm_col = -1;
    gram_touch(b, col);   // (also drops a speculative apply formed from the normalised bits)
    return kk_launch_scal(b->ctx, b->col(col), b->ld, b->norm_beta, nullptr);
}

// ... only when the column range [c0, c0 + m) contains the normalised column: the thick restart transforms the basis columns
// [0, krylovdim) and THEN asks for scale!!(r, 1 / beta) (eigsolve/lanczos.jl:109-111) -- the transform must leave the commit alone
static inline int norm_flush_range(kk_basis b, int c0, int m) {
    if (!b) return KK_OK;
    ctx_foreign_touch(b);
    if (b->tc_valid) KK_TRY(blk_commit_flush(b));
    if (b->norm_col >= c0 && b->norm_col < c0 + m) return norm_flush(b);
    return KK_OK;
}
// the column is about to be OVERWRITTEN as a whole (upload, zero, fill, copy into it): a pending normalisation of it is moot
static inline void norm_discard(kk_basis b, int col) {
    if (b && col >= 0 && b->norm_col == col) { b->norm_col = -1; gram_touch(b, col); }
}

// Cross-rank (xsync) context: the route of a sweep -- persistent kernel or low-synchronisation pair, and the panel width -- must come
// out THE SAME on every rank, or one rank spins in a launch its peer never made.  The inputs that differ between ranks
// are the shard length (Partition.even hands out different n_local) and the CU count (agreed by kk_comm_init): the longest shard
// of a slab is agreed ONCE per slab and communicator -- one all-reduce (max) at the first sweep, which every rank reaches in the
// same call (SPMD) -- and every decision of the entry point at hand is taken with it (kk_dec_ld).
int kk_comm_allreduce_max_host(kk_ctx c, double* v);   // kk_comm.hip: *v = max over the ranks (blocking)
static inline int route_agree(kk_basis b) {
    kk_ctx c = b->ctx;
    if (!kk_xs_on(c)) { c->dec_ld_local = -1; return KK_OK; }
    if (b->ld_agreed_comm != c->comm->uid) {
        double v = (double)b->ld;
        KK_TRY(kk_comm_allreduce_max_host(c, &v));
        b->ld_agreed = (int64_t)v;
        b->ld_agreed_comm = c->comm->uid;
    }
    c->dec_ld_local = b->ld; c->dec_ld = b->ld_agreed;
    return KK_OK;
}

// ---- sparse operators (kk_sparse.hip)
int get_matrix(kk_op op, int transpose, const kk_sparse_dev** M);
int check_apply(kk_op op, int transpose, kk_basis bx, kk_basis by);

// row-sharded rectangular map (kk_csr_create_sharded_rect): y = A x with the all-gather of the short vector,
// y = A' x with the reduce-scatter of the partial result
int rect_apply(kk_op op, int transpose, const double* x, double* y);

// ---- communicator internals (kk_comm.hip)
int kk_comm_allgather_i64(kk_ctx c, const int64_t* d_send, int64_t* d_recv, int64_t count);
int kk_comm_exchange_i64(kk_ctx c, const int64_t* d_send, const int64_t* send_counts, int64_t* d_recv,
                         const int64_t* recv_counts);
// collective entry points: *worst = the most severe (most negative) status any rank brought along (world 1: `local`)
int kk_comm_agree_status(kk_ctx c, int local, int* worst);
int kk_comm_allgather_f64(kk_ctx c, const double* d_stage, double* d_full, int64_t shard);
int kk_comm_reducescatter_f64(kk_ctx c, const double* d_full, double* d_stage, int64_t shard);

// ---- orthogonalisation (kk_orth.hip)
int orth_run(kk_basis b, int c0, int m, double* w, kk_orth_t alg, double eta, double* x, double* nrm, int* npasses,
             bool want_norm);
int orth_vec_run(kk_ctx c, const double* q, double* w, int64_t ld, kk_orth_t alg, double eta, double* s_out,
                 double* nrm_out, bool want_norm);
// one strict MGS sweep over V[0:m) with the fused axpy+dot kernel; coefficients land in the scalar workspace at ws_s
int pass_mgs_strict(kk_ctx c, const double* V, int64_t ld, int m, double* w, int64_t ws_s, bool want_norm, int slot,
                    const double* carry_q, const double* carry_s, bool leave_carry);
int pass_mgs_strict_sweeps(kk_ctx c, const double* V, int64_t ld, int m, int nsweeps, double* w, const int64_t* ws_s,
                           bool want_norm, int slot, const double* carry_q, const double* carry_s);
int persist_check(kk_ctx c, bool* timed_out);
int persist_check_at(kk_ctx c, int slot, double token, bool* timed_out);   // the same for a launch identified by (pinned slot, token)
int gram_ensure(kk_basis b, int upto /* exclusive */);
int gram_device(kk_basis b);   // device mirror of the host Gram rows (created on first use)
int lowsync_project_dev(kk_basis b, int m, const double* w, const double* pre_vec, const double* pre_a,
                        const double* a0_dev, int64_t ws_coef, int64_t ws_s, bool* rode, int rows_in_stream = 0);
void lowsync_commit_row(kk_basis b, int m, const double* g_host);

// ---- Krylov steps (kk_krylov.hip)
int check_square_op(kk_op op, kk_basis b);
int fetch_mark(kk_ctx c);    // record the event that marks the end of the scalar read-backs of a step ...
int fetch_wait(kk_ctx c);    // ... and wait for it (not for work enqueued after it)
int final_sync(kk_ctx c);    // fetch_mark + speculative next-step apply (if requested) + fetch_wait

// ---- block operations (kk_block.hip)
// W[:, j] = beta W[:, j] + alpha V S[:, j], j < q (S on the host, m x q column-major with leading dimension lds; m <= KK_MAX_M)
int block_update_run(kk_ctx c, const double* V, int64_t ld, int m, double* W, int64_t ldw, int q, const double* S, int lds,
                     double alpha, double beta, double* norms);
int block_inner_run(kk_ctx c, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
                    double* M, int ldm);
