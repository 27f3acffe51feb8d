/*
 * krylov_hip.h -- C ABI of libkrylov_hip.so: the MI355X (gfx950) Krylov `expand!` hot path
 * behind KrylovKit.jl's eigsolve / linsolve / svdsolve.
 *
 * KrylovKit has no FFI registry; its extension seam is Julia multiple dispatch on the vector
 * type T of OrthonormalBasis{T} and on the operator.  Each entry point below
 * names the reference method (file:line under /root/reference) that a `HipVec`-specialised
 * Julia method would forward to with `ccall((:kk_xxx, "libkrylov_hip"), Cint, (...), ...)`;
 * the bindings are spelled out in INTEGRATION.md / julia/KrylovKitHIP.jl.
 *
 * Conventions (modelled on the reference's only real FFI, the LAPACK ccalls in
 * src/dense/linalg.jl:428-454): caller-owned host buffers passed by pointer, every function
 * returns an int status (0 = OK, <0 = error; text via kk_last_error()), dimension errors are
 * reported before any work is queued (the shim turns KK_ERR_DIM into DimensionMismatch).
 *
 * Data model.  All scalars are IEEE binary64 (the north-star dtype); complex is out of scope.
 *   - A *basis* is one contiguous HBM slab of `capacity` columns, column-major, leading
 *     dimension ld >= n (ld is a multiple of 512 rows and ld/512 is odd so that equal row
 *     offsets in different columns do not alias onto the same HBM channel; rows n..ld-1 of every
 *     column are kept zero by every kernel).  A "vector" is (basis, column index).  This replaces
 *     OrthonormalBasis{T} = Vector{T} of separately allocated vectors (src/orthonormal.jl:26-54):
 *     push!/pop!/resize! become host-side integer bookkeeping.
 *   - Calls that return a host scalar are synchronous on return.  All other calls are
 *     stream-ordered on the context's HIP stream (kk_ctx_set_stream lets the caller share
 *     torch's / RCCL's stream so collectives interleave without host syncs).
 *   - One context = one GPU = one calling thread at a time (the reference's threaded kernels
 *     are fork-join inside one call, src/orthonormal.jl:95-105; no concurrent entry).
 *   - Column indices, counts: int (0-based).  Row counts: int64_t.
 */
#ifndef KRYLOV_HIP_H
#define KRYLOV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KK_VERSION 302 /* 0.3.2: same 92 entry points; sweeping single-vector stencil apply (spmv_dia_sw), cross-rank route chosen from the hand-shake's own timings ("xsync" 0 / 1 / 2, "xsync_hop_us", "comm_allreduce_us"), route decisions of a cross-rank context rank-invariant, every entry point other than expand! voids the run-ahead, BlockLanczos initialize ends in the normalised commit.  0.3.1: same 92 entry points; persistent MGS kernels on row-sharded contexts (xsync), run-ahead of the projection route, commit consumed by scale!!(r, 1/beta) (0.3.0: mgs_mode auto is the default, no limit on the basis size, constant-coefficient stencils) */

/* status codes */
#define KK_OK 0
#define KK_ERR_INVALID (-1)     /* bad handle / argument */
#define KK_ERR_DIM (-2)         /* DimensionMismatch (src/orthonormal.jl:93,140,296) */
#define KK_ERR_HIP (-3)         /* HIP runtime failure, see kk_last_error() */
#define KK_ERR_NOMEM (-4)
#define KK_ERR_ZERO_NORM (-5)   /* "initial vector should not have norm zero" (factorizations/lanczos.jl:184) */
#define KK_ERR_UNSUPPORTED (-6)
#define KK_ERR_NO_DEVICE (-7)   /* no gfx950 device visible: the product never falls back to CPU */

typedef struct kk_ctx_s* kk_ctx;
typedef struct kk_basis_s* kk_basis;
typedef struct kk_op_s* kk_op;

/* Orthogonalizer hierarchy, src/algorithms.jl:17-80.  eta is only read for the *IR variants
 * (default 1/sqrt(2), algorithms.jl:66,80). */
typedef enum {
    KK_CGS = 0,   /* ClassicalGramSchmidt   */
    KK_MGS = 1,   /* ModifiedGramSchmidt    */
    KK_CGS2 = 2,  /* ClassicalGramSchmidt2  */
    KK_MGS2 = 3,  /* ModifiedGramSchmidt2 (KrylovDefaults.orth, algorithms.jl:556) */
    KK_CGSIR = 4, /* ClassicalGramSchmidtIR */
    KK_MGSIR = 5  /* ModifiedGramSchmidtIR  */
} kk_orth_t;

/* How the MGS family is executed on the device (kk_ctx_set_option("mgs_mode", v)):
 *   0 = strict:   sequential as in src/orthonormal.jl:417-421, one basis vector after the other: a persistent kernel
 *                 (one block per CU, an ordinary launch; "persist_coop" 1 = cooperative launch) that keeps w in registers
 *                 for the whole sweep and parks the current basis vector on chip (LDS + spare registers; the remaining rows
 *                 are re-read from the L2) between its two uses, so that HBM sees one read per basis vector (8 N bytes /
 *                 vector at N = 10^7; options "mgs_persist", default 1, "persist_threads" 512 / 1024, "persist_lds"
 *                 0 / 1 / 2); when w does not fit the register file of the chip ("persist_capacity_rows"), or the context
 *                 is row-sharded, one fused axpy+dot kernel per vector (32 N bytes).
 *   1 = lowsync:  algebraically identical MGS coefficients from ONE projection pass plus a
 *                 triangular solve (on the device) with the strictly-lower Gram matrix of the
 *                 basis, maintained incrementally (16 N bytes / vector).
 *   2 = auto:     (default) by vector length on an unsharded context: the persistent PANEL kernel ("mgs_panel", default 1:
 *                 w in registers, "panel_width" = 2-3 basis vectors per grid reduction with the exact in-panel triangular
 *                 correction, every basis vector read once) from "panel_min_rows" (2.5e5) rows up to its capacity
 *                 ("panel_capacity_rows", 4.19e6 on 256 CUs), the strict persistent kernel from "persist_min_rows" (3.6e6)
 *                 rows up to "persist_capacity_rows" (10.48e6), lowsync otherwise (shorter and longer vectors; row-sharded
 *                 contexts whose ranks could not map each other's sync areas, see "xsync").  The length thresholds scale with
 *                 "num_cus" / 256.  "lookahead" (default 1): a fused Lanczos / Arnoldi expand! whose sweep runs through a
 *                 persistent kernel -- and, since 0.3.1, a Lanczos expand! on the projection route (CGS2, lowsync MGS2) --
 *                 enqueues the NEXT step's apply, scale, sweep / projection step and read-back before the host waits for the
 *                 current one (bit-identical results; dropped if anything touches the slab, or another slab of the context is
 *                 handed to any entry point, in between).
 *                 Since 0.3.2 EVERY entry point other than the expand! steps themselves (and the plain downloads / gathers, which write
 *                 no device scalar) voids what was enqueued ahead, also on the slab that owns it: a norm or inner product of a basis
 *                 column overwrites the shared device scalars the step in flight reads.  The next expand! redoes apply and sweep.
 * Row-sharded contexts (kk_comm_init): "xsync" (default 1 = where it pays, 2 = always, 0 = never; must be the same on every rank):
 * kk_comm_init times 64 in-kernel reductions over the ranks and 20 small RCCL all-reduces ("xsync_hop_us", "comm_allreduce_us": the
 * slowest rank's figures, identical on all ranks) and a sweep takes the in-kernel route iff
 *     reductions x xsync_hop_us  <=  comm_allreduce_us + vector_steps x t_sync x (rows / threshold_rows - 1)
 * (t_sync = the measured gain per vector-step and multiple of the threshold: 6 us for the register-resident kernel, 0.4 us for the panel kernel; threshold_rows = the single-chip
 * thresholds "persist_min_rows" / "panel_min_rows" scaled by the CU share): what the cross-rank round trips cost against the one RCCL
 * all-reduce per step the low-synchronisation route needs on top and the single-chip gain of the persistent kernel.  Route, panel
 * width and register tile are decided from the LONGEST shard of the slab (one all-reduce per slab and communicator at its first
 * sweep) and from a CU count all ranks share (the minimum; restored at kk_comm_destroy): every rank arrives at the same launch.
 * "xsync": the persistent kernels of the MGS
 * family sum their grid-wide inner products over the RANKS inside the launch -- block 0 of a rank stores the rank's partial as
 * a tagged 16-byte granule into every peer's sync area (4 KB of fine-grained device memory per rank, exchanged by
 * hipIpcGetMemHandle / hipIpcOpenMemHandle inside kk_comm_init and checked there by a hand-shake kernel), every block adds the
 * W partials in rank order (bit-identical on all ranks) -- so that a sharded sweep runs in the reference's sequential order
 * with every basis vector read once, RCCL serving the ghost exchange and alpha0 only; kk_ctx_get_option "xsync_active" tells
 * whether the communicator offers it (<= 8 ranks, every peer mapped), "xsync_launches" counts such launches.  "num_cus"
 * (default: what the device reports, "device_cus"): CUs this context may count on -- blocks of a persistent launch, one per CU,
 * all resident at once; jobs that share a GPU must set it (one launch hands its blocks to the 8 XCDs round-robin: W launches of
 * n blocks need W * ceil(n / 8) CUs per XCD); RANKS of one communicator that share a GPU are recognised by kk_comm_init (PCI bus
 * id; "ranks_on_this_gpu") and get 8 * (CUs per XCD / ranks - 2) unless the option was set explicitly.  "persist_timeout_ms" (default 0 = 50 x the time the sweep's bytes
 * take at 2 TB/s + 1 ms, within [20 ms, 3 s]; >= 1 s on a sharded context): how long a persistent launch waits for a block that
 * is not resident (or a peer that does not answer) before it gives up WITHOUT committing; the sweep is then repeated on the
 * launch-per-vector route, the persistent route backs off, and three timeouts in a row move the launches of a single-rank
 * context to hipLaunchCooperativeKernel ("persist_coop" reads 1 from then on). */
/* Other kk_ctx_set_option keys: "blocks_per_cu" (grid of the streaming kernels, default 4), "block_mode" (0 strict
 * block QR / re-orthogonalisation, 1 MFMA panels + CholQR2, default), "fuse_passes", "speculate" (next-step SpMV
 * enqueued before the host reads alpha/beta), "keep_mb" (MB of trailing basis columns a project pass leaves
 * cache-allocated for the unproject pass that follows; default 160 of the 256 MB Infinity Cache, 0 = none),
 * "block_async" (whole BlockLanczos step enqueued without a host round trip, default 1), "block_fuse" (bit mask of its
 * pass structure: 1 = second CholQR2 round fused, 4 = one-pass projection with Gram correction; default 5; 0 / 1 = the
 * reference's three-term-then-reorthogonalise order), "qr_skip_tol" (the second CholQR2 back-substitution of that step is
 * skipped when the block is orthonormal to this level after the first; default 2e-14, 0 = never), "resid_gram" (Gram matrix
 * of the residual block handed from one step to the next, default 1), "spmv_dia" / "spmm_dia" (diagonal kernels for operators
 * detected as grid stencils, default 1; 0 = the general ELL gather kernels), "spmv_dia_const" (a stencil whose diagonals hold
 * one value each is applied from its 5 / 9 coefficients, no indices and no values read; default 1, bit-identical to 0),
 * "fold_scale" (default 1: a Lanczos / Arnoldi expand! whose sweep ran through the persistent kernel stores the residual
 * already NORMALISED -- the kernel holds |w| before it writes w back -- so that the next expand! of the same factorization
 * needs no scale pass (factorizations/lanczos.jl:257, arnoldi.jl:209); the slab remembers (column, beta) and any other
 * entry point that is handed THAT COLUMN first multiplies it back, so residual(F) and shrink! see r as before (to 1 ulp) --
 * except kk_vec_scal(r, 1 / beta) / kk_vec_copy_scal(dst, r, 1 / beta), the scale!!(r, 1 / beta) of a thick restart
 * (eigsolve/lanczos.jl:111), which CONSUME the commit: no pass in place, one plain copy out of place, the reference's bits
 * ("norm_commits_consumed" counts them); kk_orthonormalize uses the same commit; 0 = every expand! runs its own scale pass;
 * same alpha / beta bits either way), "block_commit" (default 1: the one-pass BlockLanczos step writes its residual block W as T = W R1^-1
 * -- R1 = the first CholQR2 factor, from the Gram matrix the projection panel predicts -- straight into the next basis
 * slot (columns k+bs .. k+2bs-1, which must lie below both residual areas) and does not write the residual area; the next
 * kk_blocklanczos_expand of the same factorization starts at the second CholQR2 round (blocklanczos.jl:209-216), any other
 * entry point that is handed the slab first forms W = T R1 in the residual area, so residual(F), shrink! and restarts see
 * the block as before; "block_commits" counts the commits consumed; 0 = the residual block is always written).
 * A grid-barrier timeout of the persistent kernel (GPU shared with another job) is recovered inside the
 * library on the launch-per-vector route in the same strict order; the persistent route is retried a few sweeps later
 * (kk_ctx_get_option: "persist_timeouts", "persist_skip"); "persist_capacity_rows" = rows of a work vector the register
 * file of the chip holds (longer vectors run the low-synchronisation form in auto mode).
 * Tuning knobs without semantic effect:
 * "nt_store_rows" (results of sparse applies on at least this many rows are written with non-temporal stores, default 4e6),
 * "gram_bpc", "gram2_chunk", "gram2_pipe", "gram2_bpc", "spmm_bpc", "spmm_cols", "spmm_rpl", "spmm_dia_lines", "bu_prefetch", "gram_nt", "persist_nt",
 * "persist_lds", "persist_min_rows", "spmv_dia_pairs" (row pairs per lane of the diagonal SpMV: 0 = by size, 1 / 2 / 4),
 * "spmm_dia_al" (default 2: the sweeping multi-column apply of a value-free 5-point stencil with an even line length runs in its
 * aligned 16-byte form, 2 or 4 columns per wave; 0 = the 8-byte form; bit-identical), "spmm_dia_al_lines" (default 4: grid lines per
 * wave sweep of that form),
 * "panel_apply" (default 1: the run-ahead of an Arnoldi step on the panel kernel hands the sparse apply of a value-free 5-point stencil with an even
 * line length to the sweep launch itself -- one launch per step, w never stored and re-loaded; bit-identical to 0; "panel_apply_launches" counts),
 * "persist_apply" (default 0: the same for the Lanczos step on the register-resident strict kernel -- exact, measured 11 % slower, kept as the record),
 * "fused_step" (default 1: a Lanczos expand! with CGS2 / low-sync MGS2, or an Arnoldi expand! with CGS / CGS2 / low-sync MGS / MGS2, on a SHORT
 * vector -- single-rank context, operator in the ELL format, factorization starting at column 0, at most "fused_step_max_rows" rows
 * (default 131072) and "fused_step_m_limit" basis vectors (default 0 = by vector length: 96 - 64 n / 1e5, at least 16; -1 = no limit but 128)
 * -- runs as ONE kernel launch: apply, the grid reductions, the small solve, the update(s), the norm and the normalised commit, with the
 * scalars delivered through a pinned host slot the call polls (no copy, no event) and the next step's launch enqueued before the host looks:
 * 15 instead of 31 us per expand! at 1e3-1e4 rows, 27 instead of 34 at 1e5.  Equal to the ordinary route to rounding (another summation
 * order of the inner products).  A launch that cannot complete (a block that never becomes resident on a shared GPU) is noticed by the
 * host, the step repeated on the ordinary route and the option switched off for the context ("fstep_failures"; "fstep_launches" counts);
 * 0 = the projection pair of rounds 1-5), "fstep_blocks" (blocks per launch at most, default 128), test hook "fstep_fault",
 * "spmv_dia_sw" (default 1: the single-vector apply of a value-free 5-point stencil with an even line length whose lines start at
 * phase 0 runs as a SWEEP -- k_spmv_dia_sw: a wave walks the grid lines of its 128-wide strips through a four-line register window,
 * one 16-byte x load, one v_prev load and one store per line and lane, the fused epilogues on the line in registers; 1 / 2 = strips per
 * wave, 0 = k_spmv_dia; y bit-identical, inner products to rounding; "spmv_dia_sw_launches" counts), "spmv_dia_sw_lines" (0 = by
 * operator size, else grid lines per wave sweep),
 * "spmv_dia_aligned" (default 1: 5-point stencils with an even line length load their far neighbours as aligned 16-byte pairs
 * and take the +-1 neighbours from the lanes next door; bit-identical to 0), "panel_lag" (default 0; 1 = panel sweeps outside the
 * strict order through the cross-panel lag-1 kernel -- exact algebra, measured slower at every length, kept as the record),
 * "bu_mfma" (default 0; 1..6 = block update through the MFMA kernel in one of six tile shapes -- bit-identical to the default
 * kernel, not faster).  Test hooks: "persist_fault" (the next n persistent launches behave like a grid-barrier timeout),
 * "persist_fault_late" (cross-rank contexts: in the next n k_mgs_persist launches this rank gives up at the last reduction,
 * its partial already published). */

/* Environment variables read by the library (all optional): KK_MGS_MODE, KK_BLOCK_MODE, KK_BLOCKS_PER_CU, KK_MGS_PERSIST,
 * KK_PERSIST_THREADS, KK_PERSIST_NT, KK_NUM_CUS (defaults of the options of the same name, read at kk_ctx_create); KK_XSYNC = 0 (no
 * cross-rank sync areas: kk_comm_init leaves the RCCL routes in charge; same on every rank); KK_XSYNC_DEBUG = 1 (post-mortem of a
 * persistent launch that did not commit, on stderr); KK_SPMV_FORMAT = ell |
 * sell | csr, KK_SPMV_TILE_COLS and KK_SELLW_ROUNDS = 1 | 2 | 4 | 8 (force a device format / the column-tile width / the
 * sorting window of the tiled format in units of 256 rows, default 4, at operator creation); KK_NO_DIA (no
 * grid-stencil diagonals); KK_BASISTRANSFORM_LDS (LDS-tile basistransform instead of the MFMA kernel); KK_RCCL_LIB (path of
 * librccl for kk_comm_*); KK_LOOPBACK_GHOST_FROM = r / KK_LOOPBACK_GHOST_BELOW = r2 (test aids, world size 1 only: columns >= r / < r2 of a
 * kk_csr_create_sharded operator go through the ghost-exchange machinery although this rank owns them). */

/* ---------------------------------------------------------------- library / context */
int kk_version(void);
const char* kk_last_error(void);
int kk_device_count(int* count);
int kk_ctx_create(int device, kk_ctx* out);
int kk_ctx_destroy(kk_ctx ctx);
int kk_ctx_set_stream(kk_ctx ctx, void* hip_stream); /* NULL -> the context's own stream */
int kk_ctx_get_stream(kk_ctx ctx, void** hip_stream);
int kk_ctx_sync(kk_ctx ctx);
int kk_ctx_set_option(kk_ctx ctx, const char* key, double value);
int kk_ctx_get_option(kk_ctx ctx, const char* key, double* value);
/* elapsed GPU milliseconds between two HIP events recorded on the context stream (bench.py) */
int kk_ctx_timer_start(kk_ctx ctx);
int kk_ctx_timer_stop(kk_ctx ctx, double* ms);
/* per-kernel-class HIP-event timing (roofline leg of bench.py): when enabled every launch of
 * the named class is bracketed by events; totals are read back with kk_ctx_prof_get. */
int kk_ctx_prof_enable(kk_ctx ctx, int on); /* 0 off, 1 all kernel classes, 2 only k_project/k_unproject/k_unproj_proj */
int kk_ctx_prof_reset(kk_ctx ctx);
int kk_ctx_prof_get(kk_ctx ctx, const char* kernel_class, double* total_ms, int64_t* launches);

/* ---------------------------------------------------------------- row-sharded operation (hooks)
 * With these hooks EVERY entry point of this library works on row-sharded vectors (one process /
 * context per GPU, each holding a contiguous row block of every vector): all inner-product-type
 * results pass through a handful of finalize kernels, and right after each of them the library
 * calls `allreduce(user, device_ptr, count)` so the caller can sum the `count` doubles at
 * `device_ptr` across ranks IN PLACE on the context's stream (RCCL); sparse applies call the
 * operator's halo hook first so the caller can fill the ghost buffer.  Host-side control flow then
 * sees identical scalars on every rank.  Hooks return 0 on success. */
typedef int (*kk_allreduce_fn)(void* user, void* device_ptr, int64_t count);
typedef int (*kk_halo_fn)(void* user, const void* x_device);
int kk_ctx_set_allreduce(kk_ctx ctx, kk_allreduce_fn fn, void* user);
/* Let the caller own the two device scratch areas the reductions land in (e.g. torch tensors that
 * RCCL can address): ws (ws_count doubles) and blk (blk_count doubles); sizes from kk_ctx_workspace_size. */
int kk_ctx_workspace_size(kk_ctx ctx, int64_t* ws_count, int64_t* blk_count);
int kk_ctx_set_workspace(kk_ctx ctx, void* ws_device, void* blk_device);
int kk_op_set_halo_hook(kk_op op, kk_halo_fn fn, void* user);

/* ---------------------------------------------------------------- row-sharded operation, native (RCCL inside the library)
 * north_star: "the basis is optionally sharded row-wise across the 8 GPUs of one node with RCCL allreduce over xGMI".
 * The reference has no communication layer, so these entry points replace nothing in it; they are
 * what a Julia host needs IN ADDITION to the single-GPU calls: one process per GPU, each creates its context, rank 0
 * calls kk_comm_get_unique_id and hands the 128 bytes to the others (any channel: MPI.jl, a file, a socket), everyone
 * calls kk_comm_init.  From then on EVERY entry point of this header works on row-sharded vectors -- a basis of
 * `n` rows is this rank's block of the global vectors -- and issues its own collectives on the context stream:
 *   - all inner-product-type results (every finalize site) are summed with ncclAllReduce(f64, sum);
 *     kk_lanczos_expand with the projection-based orthogonalisers (CGS2, low-sync MGS2) needs exactly TWO per step:
 *     [alpha0 | V'w | V'v] (2m+1 doubles) and |w|^2;
 *   - operators made by kk_csr_create_sharded exchange their ghost entries with grouped ncclSend / ncclRecv before an apply
 *     (a grid stencil partitioned along grid lines runs the diagonal kernels on the rows that need no ghost entry and
 *     the gather kernels on its first and last line);
 *   - rectangular maps made by kk_csr_create_sharded_rect (GKL / svdsolve) all-gather the short vector for A x and
 *     reduce-scatter the partial result of A' x.
 * Host-side control flow sees identical scalars on every rank.  librccl is loaded (dlopen) at the first kk_comm_* call.
 * flags of kk_comm_init: KK_COMM_FORCE_COLLECTIVES issues the collectives even at world = 1 (plumbing tests on a
 * one-GPU box; by default a communicator of one rank costs nothing). */
#define KK_COMM_ID_BYTES 128
#define KK_COMM_FORCE_COLLECTIVES 1
int kk_comm_get_unique_id(void* id128);
int kk_comm_init(kk_ctx ctx, const void* id128, int rank, int world, int flags);
int kk_comm_destroy(kk_ctx ctx);
int kk_comm_info(kk_ctx ctx, int* rank, int* world, int* rccl_version);
/* collectives issued so far: all-reduces, grouped ghost exchanges, all-gather / reduce-scatter calls */
int kk_comm_stats(kk_ctx ctx, int64_t* n_allreduce, int64_t* n_p2p_groups, int64_t* n_gather);
/* caller-visible all-reduce of `count` doubles in device memory on the context stream; op: 0 sum, 1 max, 2 min */
int kk_comm_allreduce(kk_ctx ctx, void* dev_ptr, int64_t count, int op);
int kk_comm_barrier(kk_ctx ctx); /* every rank has reached this call and drained its stream */
/* This rank's rows [row_offsets[rank], row_offsets[rank+1]) of a SQUARE global operator, CSR with GLOBAL int64 column
 * indices; row_offsets (world+1 entries, row_offsets[0] = 0) is the row partition = the ownership of the vector
 * entries.  The ghost-exchange plan is negotiated inside (collective call: every rank of the communicator must make it).
 * Applies take / produce local blocks (nrows_local entries).  Without a communicator: world = 1, plain operator. */
int kk_csr_create_sharded(kk_ctx ctx, int64_t nrows_local, const int64_t* row_offsets, int64_t nnz, const int64_t* rowptr,
                          const int64_t* colind_global, const double* val, int index_base, int flags, kk_op* out);
/* This rank's rows of a RECTANGULAR global map (apply_normal / apply_adjoint of svdsolve, apply.jl:14-15).  The short
 * vectors (length ncols_global) are sharded evenly: rank r owns entries [r*s, min((r+1)*s, ncols_global)),
 * s = ceil(ncols_global / world); *ncols_local receives this rank's count (the `n` of its V-basis). */
int kk_csr_create_sharded_rect(kk_ctx ctx, int64_t nrows_local, int64_t ncols_global, int64_t nnz, const int64_t* rowptr,
                               const int64_t* colind_global, const double* val, int index_base, kk_op* out,
                               int64_t* ncols_local);
/* gather out[i] = x[idx[i]] from a raw device vector (packing halo send buffers inside a halo hook) */
int kk_gather_ptr(kk_ctx ctx, const void* x_device, const int64_t* device_idx, int64_t count, void* device_out);

/* ---------------------------------------------------------------- basis slab
 * replaces OrthonormalBasis{T} (src/orthonormal.jl:26-54) and the residual / work vectors of
 * the factorizations (factorizations/lanczos.jl:31-37, arnoldi.jl:31-36, gkl.jl:31-38). */
int kk_basis_create(kk_ctx ctx, int64_t n, int capacity, kk_basis* out);
int kk_basis_free(kk_basis b);
int kk_basis_info(kk_basis b, int64_t* n, int64_t* ld, int* capacity, void** device_ptr);
int kk_basis_upload(kk_basis b, int col, const double* host);   /* x0 in  (eigsolve/eigsolve.jl:195-201) */
int kk_basis_download(kk_basis b, int col, double* host);       /* Ritz vectors out (eigsolve/lanczos.jl:131-133) */
/* same with device pointers (n contiguous doubles, e.g. a torch tensor's data_ptr) */
int kk_basis_upload_device(kk_basis b, int col, const void* dptr);
int kk_basis_download_device(kk_basis b, int col, void* dptr);

/* ---------------------------------------------------------------- L1 vector verbs
 * VectorInterface.jl verbs as the reference calls them: the un-fused
 * fallback so that ANY KrylovKit algorithm runs on a HipVec. */
int kk_vec_dot(kk_basis bx, int cx, kk_basis by, int cy, double* out);             /* inner(x,y) */
int kk_vec_nrm2(kk_basis bx, int cx, double* out);                                 /* norm(x) */
int kk_vec_axpby(kk_basis by, int cy, kk_basis bx, int cx, double a, double b);    /* add!!(y,x,a,b): y = b*y + a*x */
int kk_vec_scal(kk_basis bx, int cx, double a);                                    /* scale!!(x,a) */
int kk_vec_copy_scal(kk_basis by, int cy, kk_basis bx, int cx, double a);          /* scale!!(y,x,a) / scale(x,a) */
int kk_vec_zero(kk_basis bx, int cx);                                              /* zerovector!! */
int kk_vec_fill_random(kk_basis bx, int cx, uint64_t seed);                        /* rand! for x0 (uniform [0,1), counter-based, independent of launch shape) */

/* ---------------------------------------------------------------- operators (src/apply.jl:1-19)
 * kk_csc_create takes Julia's SparseMatrixCSC{Float64,Int64} arrays as they are
 * (index_base = 1); kk_csr_create takes CSR with int64 row pointers and int32 columns.
 * flags: bit0 = matrix is symmetric (A' * x uses A). */
#define KK_OP_SYMMETRIC 1
int kk_csr_create(kk_ctx ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* rowptr,
                  const int32_t* colind, const double* val, int index_base, int flags, kk_op* out);
int kk_csc_create(kk_ctx ctx, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* colptr,
                  const int64_t* rowval, const double* nzval, int index_base, int flags, kk_op* out);
int kk_op_free(kk_op op);
int kk_op_info(kk_op op, int64_t* nrows, int64_t* ncols, int64_t* nnz, int* format /*0=ELL,1=CSR,2=SELL-64-sigma,3=column-tiled SELL,4=ELL + grid-stencil diagonals,5=the same with constant coefficients (applied without reading indices or values)*/,
               int64_t* device_bytes);
/* Row-sharded operators (one process per GPU): column indices >= n_local_cols address a
 * caller-owned device buffer of n_ghost doubles that the caller fills before each apply
 * (halo rows / all-gathered x); ncols of the operator must equal n_local_cols + n_ghost. */
int kk_op_set_ghost(kk_op op, int64_t n_local_cols, int64_t n_ghost, void* device_ghost);
/* y = A*x (transpose=0: apply / apply_normal, apply.jl:1,14) or A'*x (apply_adjoint, apply.jl:15) */
int kk_spmv(kk_op op, int transpose, kk_basis bx, int cx, kk_basis by, int cy);
/* affine form apply(op, x, a0, a1) = a0*x + a1*A*x (apply.jl:4-11) */
int kk_spmv_affine(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, double a0, double a1);
/* short-recurrence solvers: q = a0*p + a1*A*p with the fused inner(p, q), and the fused CG
 * update x += alpha p ; r -= alpha q ; *rnorm = |r|  (linsolve/cg.jl:35-36,61-66) */
int kk_spmv_affine_dot(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, double a0, double a1, double* dot);
int kk_cg_update(kk_basis bx, int cx, kk_basis bp, int cp, kk_basis br, int cr, kk_basis bq, int cq, double alpha,
                 double* rnorm);
/* one CG iteration body with ONE host sync: [p = r + beta p unless first]; q = a0 p + a1 A p; alpha = rho/<p,q>
 * (formed on the device); x += alpha p; r -= alpha q; returns <p,q> and |r|  (linsolve/cg.jl:60-66) */
int kk_cg_iterate(kk_op op, kk_basis b, int cx, int cr, int cp, int cq, double a0, double a1, double beta, int first,
                  double rho, double* pq, double* rnorm);
/* BiCGStab (linsolve/bicgstab.jl:118-199) in two calls per iteration.  rho, sigma, alpha, omega stay in device
 * scalars between the calls; the host reads the two norms the reference compares with tol.
 * kk_bicgstab_half, cols = {x, r, r_shadow, p, v, s, t, p_prev, v_prev} (9 columns of `b`):
 *   [p = r + beta (p_prev - omega v_prev)] ; v = a0 p + a1 A p ; alpha = rho/<r_shadow, v> ; s = r - alpha v ;
 *   returns |s| and alpha; t = a0 s + a1 A s (with <t,s>, <t,t>) is enqueued before the host waits for |s|.
 *   mode 0: rho is the device value left by kk_bicgstab_full; mode 1 (first iteration): p == r already, rho =
 *   <r_shadow, r> passed by the host; mode 2: rho passed by the host (r was replaced by the explicit residual);
 *   mode 3: collect the half that the previous kk_bicgstab_full enqueued on `ahead_cols` (only waits).
 * kk_bicgstab_full, cols = first 7 of the above: omega = <t,s>/<t,t> ; x += alpha p + omega s ; r = s - omega t ;
 *   returns |r|, rho = <r_shadow, r>, omega.  redo_t != 0: the host replaced s by the explicit residual (:143-146).
 *   ahead_cols != NULL: the next BiCG half is enqueued on those 9 columns (p, v double-buffered against
 *   p_prev, v_prev) before the host waits, so the GPU works through the round trip. */
int kk_bicgstab_half(kk_op op, kk_basis b, const int* cols, double a0, double a1, int mode, double rho,
                     double* snorm, double* alpha);
int kk_bicgstab_full(kk_op op, kk_basis b, const int* cols, double a0, double a1, int redo_t, const int* ahead_cols,
                     double* rnorm, double* rho, double* omega);
/* LSMR (lssolve/lsmr.jl:61-128) fused vector updates.
 * kk_lsmr_step_u:  Ah = Av - c Ah ; u = Av - alpha u ; *beta = |u|        (:64-68; one pass, one host sync)
 * kk_lsmr_update:  hbar = h - c1 hbar ; x += c2 hbar ; h = v - c3 h (cv < 0: skip the h update)   (:121-128)
 *   called once for (h, hbar, x, v) and once, with cv = -1 and c2 negated, for (Ah, Ahbar, r). Stream-ordered. */
int kk_lsmr_step_u(kk_basis b, int c_av, int c_ah, int c_u, double c, double alpha, double* beta);
int kk_lsmr_update(kk_basis b, int ch, int chbar, int cx, kk_basis bv, int cv, double c1, double c2, double c3);
/* gather x[idx[i]] -> out[i] on device (packing halo/ghost send buffers) */
int kk_gather(kk_basis bx, int cx, const int64_t* device_idx, int64_t count, void* device_out);

/* ---------------------------------------------------------------- L2 basis operations
 * (src/orthonormal.jl).  The basis vectors addressed are columns c0 .. c0+m-1 of `b`. */
/* project!!  y[j] = beta*y[j] + alpha*inner(b[c0+j], x)   (orthonormal.jl:88-118) */
int kk_project(kk_basis b, int c0, int m, kk_basis bx, int cx, double alpha, double beta, double* y);
/* unproject!! y = beta*y + alpha*sum_j b[c0+j]*x[j]        (orthonormal.jl:132-196); also
 * Base.:*(b, x) (orthonormal.jl:57-60) and the GMRES x-update (linsolve/gmres.jl:105-108). */
int kk_unproject(kk_basis by, int cy, kk_basis b, int c0, int m, const double* x, double alpha, double beta);
/* rank1update! b[c0+j] = beta*b[c0+j] + alpha*y*conj(x[j]) (orthonormal.jl:210-275) */
int kk_rank1update(kk_basis b, int c0, int m, kk_basis by, int cy, const double* x, double alpha, double beta);
/* basistransform! b[c0+j] <- sum_i b[c0+i]*U[i,j], i<m, j<n; U column-major, ldu >= m
 * (orthonormal.jl:291-354); columns c0+n .. c0+m-1 keep their old contents. */
int kk_basistransform(kk_basis b, int c0, int m, int n, const double* U, int ldu);
/* rmul!(b, G::Givens): (q1,q2) <- (c*q1 - s*q2, s*q1 + c*q2)   (dense/givens.jl:12-36) */
int kk_givens_rmul(kk_basis b, int i1, int i2, double c, double s);
/* rmul!(b, H::Householder) over columns c0..c0+m-1 (dense/reflector.jl:143-154):
 * w = sum_j b[c0+j]*v[j]; b[c0+j] -= beta*w*v[j].  One fused kernel (row-local). */
int kk_householder_rmul(kk_basis b, int c0, int m, const double* v, double beta);
/* orthogonalize!!(w, b, x, alg) (orthonormal.jl:378-452): w <- w - sum_j x[j] b[c0+j].
 * x (m doubles, host) receives the coefficients, *nrm (optional) the norm of the result
 * (it is a by-product of the last pass), *npasses the number of passes over the basis. */
int kk_orthogonalize(kk_basis b, int c0, int m, kk_basis bw, int cw, kk_orth_t alg, double eta,
                     double* x, double* nrm, int* npasses);
/* orthonormalize!! (orthonormal.jl:522-527): as above, then w <- w/nrm. */
int kk_orthonormalize(kk_basis b, int c0, int m, kk_basis bw, int cw, kk_orth_t alg, double eta,
                      double* x, double* nrm, int* npasses);
/* orthogonalize!!(v, q, alg) vector-vs-vector variants (orthonormal.jl:455-489) */
int kk_orthogonalize_vec(kk_basis bq, int cq, kk_basis bw, int cw, kk_orth_t alg, double eta,
                         double* s, double* nrm);
/* Gram-matrix bookkeeping for mgs_mode=1: tell the library that columns c0.. of `b` changed
 * other than through the fused expands (basistransform, upload, ...). */
int kk_basis_invalidate_gram(kk_basis b);

/* ---------------------------------------------------------------- L3 fused expand! steps
 * One call = one `expand!` = one operator application (the reference's `numops` unit,
 * eigsolve/lanczos.jl:79).  One host synchronisation at the end (alpha, beta, h are needed on
 * the host every iteration: eigsolve/lanczos.jl:34-45, linsolve/gmres.jl:55); the *IR variants
 * add one per extra pass because the loop condition lives on the host side of the reference too.
 */

/* Lanczos expand! (factorizations/lanczos.jl:250-272) + lanczosrecurrence (:295-376).
 * On entry columns c0..c0+k-1 hold V and column c0+k holds the residual r with |r| = beta_old.
 * On return column c0+k holds v_{k+1} = r/beta_old and column c0+k+1 the new residual.
 * For keepvecs=false (only CGS/MGS, lanczos.jl:140-142) pass k = min(k,1): only V[end-1] is read. */
int kk_lanczos_expand(kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, double eta,
                      double beta_old, double* alpha, double* beta, int* npasses);
/* Lanczos initialize (factorizations/lanczos.jl:180-222): column c0 holds x0 on entry; on
 * return column c0 = v1, column c0+1 = r. */
int kk_lanczos_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta,
                          double* alpha, double* beta);

/* Arnoldi expand! (factorizations/arnoldi.jl:199-219) + arnoldirecurrence!! (:239-245).
 * Columns as for Lanczos.  h receives the k+1 new Hessenberg entries H[1:k+1, k+1]
 * (the packed column, dense/packedhessenberg.jl:32-48); *beta = |r|. */
int kk_arnoldi_expand(kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, double eta,
                      double beta_old, double* h, double* beta, int* npasses);
/* Arnoldi initialize (arnoldi.jl:135-175); identical arithmetic to the Lanczos one. */
int kk_arnoldi_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta,
                          double* alpha, double* beta);

/* GKL expand! (factorizations/gkl.jl:246-269) + gklrecurrence (:294-404).
 * U-basis bu: columns 0..k-1 = U, column k = r (|r| = beta_old).  V-basis bv: columns 0..k-1 = V.
 * On return bu column k = u_{k+1}, bu column k+1 = new r, bv column k = v_{k+1}. */
int kk_gkl_expand(kk_op op, kk_basis bu, kk_basis bv, int k, kk_orth_t orth, double eta,
                  double beta_old, double* alpha, double* beta, int* npasses_v, int* npasses_u);
/* GKL initialize (gkl.jl:183-215): bu column 0 holds u0 on entry. */
int kk_gkl_initialize(kk_op op, kk_basis bu, kk_basis bv, double* alpha, double* beta);

/* ---------------------------------------------------------------- BlockLanczos (config 5)
 * src/factorizations/blocklanczos.jl.  A Block of p vectors = p consecutive columns of a slab.
 * kk_ctx_set_option("block_mode", v): 0 = strict (every inner/add!! pair executed sequentially as
 * the reference does), 1 = panel (default): block_inner / block_reorthogonalize! run as
 * tall-skinny panels on v_mfma_f64_16x16x4_f64 + a multi-right-hand-side update kernel. */
/* block_inner (blocklanczos.jl:43-52): M[i + ldm*j] = inner(X[i], Y[j]); X = columns cx..cx+p-1 of bx */
int kk_block_inner(kk_basis bx, int cx, int p, kk_basis by, int cy, int q, double* M, int ldm);
/* apply(f, ::Block) (blocklanczos.jl:39): Y[j] = A X[j], j < nb (one SpMM: the matrix is read once) */
int kk_block_apply(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, int nb);
/* W[j] = beta*W[j] + alpha*sum_i b[c0+i]*S[i + lds*j], i < m, j < q: the AX[j] -= X[i]*M[i,j] +
 * Xprev[i]*conj(B[j,i]) update (blocklanczos.jl:253-260), the panel update of
 * block_reorthogonalize!, basistransform! of a residual block (eigsolve/blocklanczos.jl:96).
 * norms (optional, q doubles) receives |W[j]| of the result. W must not overlap the basis range. */
int kk_block_update(kk_basis bw, int cw, int q, kk_basis b, int c0, int m, const double* S, int lds, double alpha,
                    double beta, double* norms);
/* block_qr! (blocklanczos.jl:312-353): MGS QR with rank detection of the p columns c_in..c_in+p-1.
 * The orthonormal vectors are written COMPACTED to columns c_out..c_out+ngood-1 (c_out == c_in:
 * in place; otherwise the input block is left untouched and serves as `Rcopy`, blocklanczos.jl:209).
 * R (ngood x p, column-major, ldr >= p) = R[good_idx, :]; good_idx[p] (0-based); *is_drift as in :334-336. */
int kk_block_qr(kk_basis b, int c_in, int p, int c_out, double tol, double* R, int ldr, int* good_idx, int* ngood,
                int* is_drift);
/* block_reorthogonalize! (blocklanczos.jl:277-284): W = columns cw..cw+q-1 against basis columns c0..c0+m-1 */
int kk_block_reorthogonalize(kk_basis b, int c0, int m, int cw, int q);
/* initialize(::BlockLanczosIterator) (blocklanczos.jl:159-198).  On entry columns c_x0..c_x0+bs0-1 hold
 * the start block.  On return columns 0..bs-1 = V (orthonormalised start block, bs = rank), columns
 * c_r..c_r+bs-1 = residual block; M1 (bs x bs, ldm) = block_inner(X1, A X1); *norm_R = Frobenius norm. */
int kk_blocklanczos_initialize(kk_op op, kk_basis b, int c_x0, int bs0, int c_r, double qr_tol, int* bs,
                               double* M1, int ldm, double* norm_R);
/* expand!(::BlockLanczosIterator, state) (blocklanczos.jl:200-240) + block_lanczosrecurrence (:242-263).
 * Columns 0..k-1 = V, the residual block (bs_r vectors) sits in columns c_r.. ; the new basis
 * vectors go to columns k..k+bs_next-1 and the new residual block to columns c_rnext.. (must not
 * overlap [0, k+bs_r)).  B (bs_next x bs_r, ldb), M (bs_next x bs_next, ldm) are the new blocks of
 * the block-tridiagonal matrix (blocklanczos.jl:220-221,229). */
int kk_blocklanczos_expand(kk_op op, kk_basis b, int k, int bs_r, int c_r, int c_rnext, double qr_tol, int* bs_next,
                           double* B, int ldb, double* M, int ldm, double* norm_R, int* is_drift);

/* ---------------------------------------------------------------- split-phase pieces for
 * row-sharded (one process per GPU) runs.  Every inner-product-type result is a LOCAL partial
 * written to a CALLER-OWNED device buffer (e.g. a torch tensor), so the caller can all-reduce
 * it in place with RCCL on the same stream between the two halves of a pass.
 * Nothing here synchronises the host. */
/* w = A v - beta_old v_prev (col_prev < 0: no subtraction);
 * dev_dot[0] = local <v, A v> (dot_mode 1, lanczos.jl:298) or local <v, w> (dot_mode 2, :308); 0: none */
int kk_apply_fused_dev(kk_op op, kk_basis b, int col_v, int col_prev, int col_w, double beta_old, int dot_mode,
                       void* dev_dot);
/* same with device-resident scalars: w = A (xs*v) - bp*v_prev with xs = *dev_xscale (e.g. 1/beta of the
 * previous iteration, all-reduced on the device) and bp = *dev_bprev; NULL pointers mean xs = 1, bp = beta_old.
 * Lets the next iteration's SpMV be enqueued before the host has seen beta. */
int kk_apply_fused_dev2(kk_op op, kk_basis b, int col_v, int col_prev, int col_w, const void* dev_xscale,
                        const void* dev_bprev, double beta_old, int dot_mode, void* dev_dot);
/* dev_out[j] = local <b[c0+j], x>, j < m; with a second right-hand side (col_rhs2 >= 0, same
 * basis as x) dev_out[m+j] = local <b[c0+j], rhs2>: the Gram row of the newest vector rides along. */
int kk_project_dev(kk_basis b, int c0, int m, kk_basis bx, int cx, int col_rhs2, void* dev_out);
/* y = beta*y + alpha*sum_j coef[j] b[c0+j] (coef on the host, passed in the kernarg segment);
 * dev_nrm (optional, 3 doubles) = local |y|^2, its sqrt, 1/sqrt */
int kk_unproject_dev(kk_basis by, int cy, kk_basis b, int c0, int m, const double* coef, double alpha, double beta,
                     void* dev_nrm);
/* as kk_unproject_dev with the m coefficients read from device memory (no host round trip) */
int kk_unproject_devcoef(kk_basis by, int cy, kk_basis b, int c0, int m, const void* dev_coef, double alpha,
                         double beta, void* dev_nrm);
int kk_dot_dev(kk_basis bx, int cx, kk_basis by, int cy, void* dev_out);      /* dev_out[0] = local <x,y> */
int kk_nrm2_dev(kk_basis bx, int cx, void* dev_out3);                          /* local |x|^2, sqrt, 1/sqrt */
/* y += sign * dev_a[0] * x   (device scalar, e.g. an all-reduced coefficient) */
int kk_axpy_dev(kk_basis by, int cy, kk_basis bx, int cx, const void* dev_a, double sign);
/* x *= 1/sqrt(dev_nrm2[0])   (normalise by an all-reduced squared norm, no host round trip) */
int kk_scal_rsqrt_dev(kk_basis bx, int cx, const void* dev_nrm2);

/* row-sharded Lanczos step, coefficient algebra between its two all-reduces in one launch (no host sync):
 * buf = [alpha0 | V'w (m) | V'v (m)] summed over the ranks; rhs = V'w - alpha0 V'v; lowsync != 0 stores V'v[0:m-1] as the
 * Gram row of the newest vector in L (cap x cap, row-major) and solves (I + L) s = rhs; coef_out = s with alpha0 added to
 * its last entry; res = {alpha0, s[m-1]}.  kk_norm_scalars_dev: sc = {1/sqrt(n2), sqrt(n2)}, *res2 = n2. */
int kk_lanczos_coef_dev(kk_ctx ctx, const void* buf_dev, void* L_dev, int cap, int m, int lowsync, void* coef_out_dev,
                        void* res_dev);
int kk_norm_scalars_dev(kk_ctx ctx, const void* nrm2_dev, void* sc_dev, void* res2_dev);

#ifdef __cplusplus
}
#endif
#endif /* KRYLOV_HIP_H */
