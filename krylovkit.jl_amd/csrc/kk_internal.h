// Internal declarations shared by the .hip translation units of libkrylov_hip.so.
// gfx950 (CDNA4) only: wave64, 256 CUs in 8 XCDs, HBM3E-bound streaming kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include "../../include/krylov_hip.h"

#define KK_MAX_M 256          // max basis vectors touched by one project/unproject call
#define KK_MAX_BLOCKS 4096    // max thread blocks of a reducing kernel (partials row stride)
#define KK_BLK_SCRATCH 131072 // doubles of device/pinned scratch for block matrices (gram panels, S)
#define KK_STAGE_SLOTS 8      // ring of coefficient-panel staging slots inside the block scratch (first half)
#define KK_STAGE_DOUBLES 4096  // doubles per slot (KK_MAX_M rows x 16)
// synchronisation area of the persistent MGS kernels: 2 granule sets x one 128-byte line per block + error flag
#define KK_SYNC_LINE 128
#define KK_SYNC_MAX_BLOCKS 1024
#define KK_SYNC_ERR_OFFSET (2 * KK_SYNC_MAX_BLOCKS * KK_SYNC_LINE)
#define KK_SYNC_BYTES (KK_SYNC_ERR_OFFSET + 64)
// one-launch Lanczos step of short vectors (kk_kernels_fstep.hip): at most KK_FS_MAX_M basis vectors, at most KK_FS_MAX_BLOCKS blocks of 256 threads
// x up to 8 row pairs (64 blocks, the default: 262 144 rows); granule area = (2 m + 1 values + the norm) x blocks x 16 bytes, error flag behind it
#define KK_FS_MAX_M 128
#define KK_FS_MAX_BLOCKS 128
#define KK_FS_SET_BYTES ((2 * KK_FS_MAX_M + 2) * KK_FS_MAX_BLOCKS * 16)   // one granule set; two sets: the two passes of an Arnoldi CGS2 / MGS2 step alternate
#define KK_FS_SYNC_BYTES (2 * KK_FS_SET_BYTES)
#define KK_MAX_DEVICES 64     // per-device bookkeeping of function attributes
#define KK_TPB 256            // threads per block of every streaming kernel (4 waves)
#define KK_SUB 512            // rows covered by one block sub-step: 256 threads x 2 rows (16 B/lane)
// register tile of the two basis-streaming kernels: RG sub-steps of 512 rows per row group (2*RG rows per
// lane) x CB basis columns per load batch  ->  RG*CB 16-byte loads in flight per lane.
// Defaults from tools/tile_sweep.sh on MI355X (10M rows, m=2..100): project 8x2 (16x1 with the Gram-row ride-along),
// unproject 16x2 -- 7 % faster than 8x4 once the leftover chunks of a block run through the same code at 8/4/2/1
// chunks (before that, 16-row groups lost 45 % in the masked tail path).
#ifndef KK_RG_P
#define KK_RG_P 8
#endif
#ifndef KK_CB_P
#define KK_CB_P 2
#endif
#ifndef KK_RG_P2             // project with a second right-hand side (Gram row ride-along of low-sync MGS)
#define KK_RG_P2 16
#endif
#ifndef KK_CB_P2
#define KK_CB_P2 1
#endif
#ifndef KK_RG_U
#define KK_RG_U 16
#endif
#ifndef KK_CB_U
#define KK_CB_U 2
#endif

// scalar workspace layout (doubles)
#define WS_S 0                // coefficients of the current pass            [KK_MAX_M]
#define WS_G (WS_S + KK_MAX_M)      // second right-hand side (Gram row)     [KK_MAX_M]
#define WS_X (WS_G + KK_MAX_M)      // accumulated coefficients              [KK_MAX_M]
#define WS_Y (WS_X + KK_MAX_M)      // first-pass MGS coefficients (device solve)  [KK_MAX_M]
#define WS_Z (WS_Y + KK_MAX_M)      // second-pass coefficients                  [KK_MAX_M]
#define WS_SCAL (WS_Z + KK_MAX_M)   // named scalars                         [64]
#define WS_USER (WS_SCAL + 64)      // row-sharded fused steps: [alpha0 | V'w (m) | V'v (m)] of ONE all-reduce   [KK_WS_USER]
#define KK_WS_USER 4096
#define WS_SHBUF WS_USER
#define WS_TOTAL (WS_USER + KK_WS_USER)
// a finalize with_sqrt writes three consecutive slots: sum, sqrt(sum), 1/sqrt(sum)
enum { SC_ALPHA0 = 0, SC_NRM2 = 1, SC_NRM = 2, SC_INVNRM = 3, SC_DOT = 4, SC_TMP0 = 5, SC_TMP1 = 6, SC_TMP2 = 7,
       SC_NRM2B = 8, SC_NRMB = 9, SC_INVNRMB = 10, SC_DOTB = 11,
       SC_SPECA = 12 /* alpha of a speculative next-step SpMV: written by nothing else */,
       SC_PERSIST_OK = 13 /* completion token of the last persistent MGS launch (travels with the sweep's scalars) */,
       SC_XS = 14 /* factor a speculative apply still has to put on the residual of that launch: 1 (stored normalised) or 1/|w| */,
       SC_BICG = 16 /* rho, rho_old, sigma, alpha, omega, <t,s>, <t,t> (+2 for the sqrt triple) */,
       SC_BICG_SN = 25 /* |s|^2, |s|, 1/|s| */, SC_BICG_RN = 28 /* |r|^2, |r|, 1/|r| */ };

void kk_set_error(const char* fmt, ...);
int kk_hip_fail(hipError_t e, const char* what, const char* file, int line);
#define KK_HIP(call)                                                        \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return kk_hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)
#define KK_CHECK(cond, code, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            kk_set_error(__VA_ARGS__);            \
            return (code);                        \
        }                                         \
    } while (0)
#define KK_TRY(call)                \
    do {                            \
        int _s = (call);            \
        if (_s != KK_OK) return _s; \
    } while (0)

// ---- cross-rank in-kernel reduction ("xsync", kk_comm.hip / kk_xsync.h): the persistent MGS kernels of a row-sharded run sum
// their grid-wide inner products over the ranks THEMSELVES -- block 0 of every rank stores its rank's partial as a tagged
// 16-byte granule into every peer's sync area (fine-grained device memory, mapped through hipIpcOpenMemHandle: stores travel
// over xGMI), every block polls its OWN rank's area and adds the W partials in rank order (same bits on all ranks).
#define KK_XS_MAX_RANKS 8        // one node
#define KK_XS_MAX_VALS 8         // values per reduction (panel kernel: P (P + 1) / 2 <= 6)
#define KK_XS_SET_BYTES (KK_XS_MAX_VALS * KK_XS_MAX_RANKS * 16)   // granule (value v, rank r) of a set at (v * 8 + r) * 16
#define KK_XS_ERR_OFFSET (2 * KK_XS_SET_BYTES)                    // id of the launch a peer gave up in (0 = none)
#define KK_XS_DBG_OFFSET 3072                                     // post-mortem of the first block that gave up in a cross-rank reduction: [launch, tag, nval, why, block, red | 64 observed tags]
#define KK_XS_BYTES 4096
struct kk_xs_dev {               // by value in the kernarg segment of the persistent kernels; world == 0: single-rank launch
    const unsigned long long* table = nullptr;   // device: base address of every rank's area as mapped into THIS process
    char* mine = nullptr;        // this rank's area
    unsigned tag0 = 0;           // tag of the launch's first reduction (tags count reductions over the life of the communicator: set = tag & 1)
    unsigned launch = 0;         // id of the launch (same on all ranks)
    int rank = 0, world = 0;
};

struct kk_prof_entry {
    double ms = 0;
    int64_t launches = 0;
};

// RCCL communicator of a row-sharded run (kk_comm.hip); the library dlopens librccl at kk_comm_init
struct kk_comm_s {
    void* nccl = nullptr;   // ncclComm_t
    uint64_t uid = 0;       // unique over the life of the process (what a slab's agreed length refers to)
    int rank = 0, world = 1;
    bool active = false;    // collectives are issued: world > 1, or forced at world 1 (plumbing tests on a 1-GPU box)
    int64_t n_allreduce = 0, n_p2p = 0, n_gather = 0;   // statistics (kk_comm_stats)
    // cross-rank in-kernel reduction (set up by kk_comm_init when every rank could map every peer's area)
    bool xs_active = false;
    char* xs_mine = nullptr;                    // fine-grained device memory, KK_XS_BYTES
    void* xs_peer[KK_XS_MAX_RANKS] = {};        // rank r's area in this process (xs_peer[rank] == xs_mine)
    bool xs_opened[KK_XS_MAX_RANKS] = {};       // ... mapped by hipIpcOpenMemHandle (to be closed)
    unsigned long long* xs_table = nullptr;     // device copy of xs_peer
    unsigned xs_red = 0;                        // reductions issued so far (identical on all ranks: SPMD call sequence)
    unsigned xs_launch = 0;                     // launches issued so far
    int64_t n_xs_launches = 0;                  // statistics
    int xs_share = 1;                           // ranks of this communicator on this rank's GPU (> 1: num_cus was cut to the rank's share)
    int cus_before = 0;                         // "num_cus" of the context before the communicator set it to the value ALL ranks launch with (0: untouched); restored by xs_release
    bool xs_clear_word = false;                 // a lost launch left its id in this rank's abort word: cleared before the next persistent launch
    double xs_hop_us = 0;                       // one in-kernel cross-rank reduction, measured by the hand-shake (slowest rank)
    double ar_us = 0;                           // one RCCL all-reduce of 8 doubles on the context stream, measured by the hand-shake (slowest rank)
};

// epilogue / fusion description of an SpMV launch
struct kk_spmv_fuse {
    double a1 = 1.0;                 // y = a1*(A x)*xscale + a0*x - bprev*vprev
    double a0 = 0.0;
    const double* xscale_dev = nullptr;  // optional device scalar multiplying x (e.g. 1/alpha)
    const double* vprev = nullptr;   // optional vector subtracted with weight bprev
    double bprev = 0.0;
    const double* bprev_dev = nullptr;   // if set: weight = *bprev_dev (device scalar)
    int dot_mode = 0;                // 0 none, 1 = <x, A x> before subtracting vprev (CGS order,
                                     // lanczos.jl:298), 2 = <x, y> after (MGS order, lanczos.jl:308),
                                     // 3 = <dot_vec, y> (BiCGStab <r_shadow, A p>)
    const double* dot_vec = nullptr; // third vector of dot_mode 3
    double* dot_out = nullptr;       // device scalar receiving the dot (required when dot_mode != 0)
    double* nrm_out = nullptr;       // optional: device triple receiving |y|^2, sqrt, 1/sqrt
};

// constant-coefficient grid stencil as the kernels take it: coefficient of diagonal q, position of row 0 inside its grid line, line length
struct dia_cst { double c[9]; int64_t phase, D; };
// A sparse apply that the NEXT persistent sweep launch performs itself (round 6: k_mgs_panel<.., APPLY> forms w = (A x) * xs in its registers instead of
// loading a w that a separate launch has just written): set by the run-ahead of an Arnoldi step, consumed by pass_mgs_strict_sweeps
struct kk_sweep_apply {
    bool on = false;
    const struct kk_sparse_dev* M = nullptr;
    const double* x = nullptr;
    kk_spmv_fuse f;   // the apply as the separate launch would have run it (scale, - beta v_prev, alpha dot): what the sweep kernel does instead -- or, on a route that cannot, kk_launch_spmv
};

struct kk_ctx_s {
    int device = 0;
    kk_comm_s* comm = nullptr;   // set by kk_comm_init: every reduction of the library is summed over the ranks
    // route decisions of a cross-rank (xsync) context use the LONGEST shard of the slab at hand, agreed once per slab
    // (route_agree, kk_host.h): local leading dimension the agreement below belongs to, and the agreed one
    int64_t dec_ld_local = -1, dec_ld = -1;
    bool ar_suspend = false;     // fused sharded steps collect several local partials and all-reduce them at once
    int num_cus = 256;           // blocks of a persistent launch / partition unit of the streaming kernels (option "num_cus": fewer than the device has when the GPU is shared)
    int dev_cus = 256;           // what the device reports
    int dev_xcds = 8;            // ... and its XCDs (a launch deals its blocks to them round-robin)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    double* ws = nullptr;        // device scalar workspace [WS_TOTAL] (ws_own unless the caller supplied one)
    double* ws_own = nullptr;
    double* blk_own = nullptr;
    kk_allreduce_fn allreduce = nullptr;   // row-sharded operation: sum partial results across ranks in place
    void* allreduce_user = nullptr;
    double* partials = nullptr;  // device partial sums [(2*KK_MAX_M + 8) * KK_MAX_BLOCKS]
    double* h_pin = nullptr;     // pinned host staging [4][WS_TOTAL]
    double* h_U = nullptr;       // pinned staging for basistransform's U [KK_MAX_M^2]
    double* blk = nullptr;       // device scratch for small block matrices [KK_BLK_SCRATCH]
    double* h_blk = nullptr;     // pinned twin of blk
    int block_mode = 1;          // 0 strict, 1 panel (MFMA gram + multi-rhs update)
    int spmm_bpc = 4;            // resident blocks per CU of the multi-column sparse apply (L2 window, see kk_launch_spmm); 0 = fill the chip
    int spmm_rpl = 2;            // SpMM on ELL: rows per lane (1 or 2)
    int spmv_dia = 1;            // single-column apply of a detected grid stencil: diagonal kernel (0: ELL gather kernel)
    int spmv_dia_pairs = 0;      // row pairs per lane of k_spmv_dia: 0 = by size, or 1 / 2 / 4 (4: value-free form only)
    int spmv_dia_aligned = 1;    // ... 5-point stencils with an even line length: far neighbours as aligned 16-byte pairs, +-1 neighbours by lane shift (0: the 8-byte loads of rounds 2-4)
    int spmv_dia_const = 1;      // ... value-free kernel when the stencil has constant coefficients (0: always stream the diagonals)
    int spmv_dia_sw = 1;         // ... value-free 5-point stencils with an even line length: the SWEEPING kernel (k_spmv_dia_sw), strips per wave (1 / 2); 0 = k_spmv_dia
    int spmv_dia_sw_lines = 0;   // ... its grid lines per wave sweep (0: by operator size)
    int64_t spmv_dia_sw_launches = 0;   // launches of the sweeping single-vector apply (diagnostics: "spmv_dia_sw_launches")
    int spmm_dia = 1;            // multi-column apply of a detected grid stencil: sweeping diagonal kernel (0: ELL gather kernel)
    int spmm_dia_lines = 16;     // ... grid lines per wave sweep
    int spmm_dia_al = 2;         // ... aligned 16-byte form (k_spmm_dia_al): columns per wave (2 / 4), 0 = the 8-byte form
    int spmm_dia_al_lines = 4;   // ... and its grid lines per wave sweep (short sweeps win: tools/spmm_dia_ab.py, profiles/r05_spmm_dia_ab.jsonl)
    int spmm_cols = 16;          // SpMM on ELL: right-hand sides per launch (16, 8 or 4)
    int bu_mfma = 0;             // block update W = beta W + alpha V S through the MFMA kernel (k_block_update_mfma: transposed product, 16-byte operand loads)
    int bu_prefetch = 1;         // block update kernel: 1 = coefficient panel in LDS (default), 0 = scalar-load kernel of round 1, 8/16/24 = deep-prefetch experiments
    int gram_nt = 0;             // Gram panel: non-temporal loads for the X stream
    // Gram matrix of the residual block the last asynchronous one-pass block step left behind (device, AB_GW), valid while
    // nothing touched that block: the next step's first CholQR2 round starts from it instead of reading the block
    bool gw_valid = false;
    uint64_t gw_basis = 0;   // uid of the slab
    int gw_col = -1, gw_p = 0;
    int64_t nt_store_rows = 4000000;   // SpMV results of at least this many rows are written with non-temporal stores
    int resid_gram = 1;          // use it (0: always run the Gram pass)
    int block_commit = 1;        // one-pass block step: residual update writes T = W R1^-1 into the next basis slot (normalised commit)
    uint64_t tc_owner = 0;       // uid of the slab whose pending commit (R1, G2 = T'T) sits in the block scratch; 0 = none
    int64_t block_commits = 0;   // commits consumed by a following expand! (diagnostics)
    double qr_skip_tol = 2e-14;  // async block step: skip the second CholQR2 back-substitution when |Q1'Q1 - I|_max <= this (0: never)
    double last_qr_dev = 0;      // |Q1'Q1 - I|_max of the last asynchronous block step (diagnostics)
    int gram2_chunk = 80;        // two-panel Gram kernel (one-pass block step): basis columns per launch (64, 80 or 128)
    int gram2_pipe = 1;          // two-panel Gram kernel: software-pipelined form (k_block_gram2p), NG <= 5
    int gram2_bpc = 0;           // its blocks per CU (0 = what the occupancy query reports)
    int gram_bpc = 8;            // Gram panel: blocks per CU of a one-group (p <= 16) launch; NG groups -> gram_bpc / NG, >= 2
    int block_fuse = 5;          // async block step, bit mask: 1 = CholQR2 round 2 fused (update + Gram in one pass), 2 = three-term
                                 // update folded into the re-orthogonalisation panel, 4 = one-pass projection against the whole basis
    int block_async = 1;         // panel mode: whole block step enqueued without host round trips (device-side CholQR2 algebra)
    int blocks_per_cu = 4;       // 4 resident 256-thread blocks per CU (measured best on the 10M-row sweep)
    int mgs_mode = 2;            // MGS family: 0 strict (reference order), 1 low-synchronisation, 2 auto = strict through the persistent
                                 // kernel where that is the faster one (eligible and >= persist_min_rows rows), low-sync otherwise
    int64_t persist_min_rows = 3600000;   // auto mode: below this the per-vector grid reduction outweighs the saved basis traffic (and the panel kernel, which holds up to 4.19 M rows, is the faster persistent route)
    int keep_mb = 160;           // MB of trailing basis columns a project pass leaves cache-allocated for the unproject
                                 // pass that follows (the Infinity Cache holds 256 MB); 0 = all loads non-temporal
    int mgs_panel = 1;           // MGS sweeps of vectors of <= 16 grid-rows (4.19 M rows) through the persistent PANEL kernel (kk_kernels_panel.hip)
    int panel_lag = 0;           // panel sweeps outside the strict order through the cross-panel lag-1 kernel (k_mgs_panel_lag) where the vector fits three
                                 // register-resident panels of two vectors.  OFF by default: measured SLOWER than k_mgs_panel at every length it can hold
                                 // (profiles/r05_panel_lag_ab.jsonl: 4289 vs 5552 it/s on a 1 M-row GMRES(60) cycle) -- the reduction chain of its wave 0
                                 // (publish -> sweep -> publish) is as long as the reduction it was meant to hide; kept, tested, as the record of the experiment
    int panel_width = 0;         // basis vectors per grid reduction of that kernel: 0 = by vector length (3 / 2 / 1), else min(value, by length); mgs_mode 0 forces 1
    int64_t panel_min_rows = 250000;   // auto mode: below this the projection pair (two launch-bound passes) is the faster route.  Round 4: 1.4 M rows (one grid
                                       // reduction per panel cost ~6 us); since the values of a reduction are swept by as many waves at once (round 5) the panel
                                       // kernel wins from ~0.2 M rows: Arnoldi MGS2 cycle of 60, 128 k rows 11.7 vs 13.1 k it/s (projection pair ahead), 250 k 12.3 vs
                                       // 11.6, 500 k 11.1 vs 8.0, 1 M 8.3 vs 5.5 (profiles/r05_panel_sweep_par.jsonl)
    int persist_apply = 0;       // Lanczos steps on the register-resident strict kernel, value-free 5-point stencil: the kernel forms w = A v - beta v_prev and the alpha dot itself.
                                 // OFF by default: measured SLOWER on the headline sweep -- 1140-1158 vs 1281-1283 it/s, the launch grows from 0.718 to 0.858 ms
                                 // (profiles/r06_persist_apply_ab.txt): with the grid-strided row ownership of this kernel the ten batches of apply loads are ten
                                 // exposed memory round trips at the head of every launch, and the instantiation spills 17 registers (2 without); the same idea
                                 // on the panel kernel, whose blocks own contiguous rows, is a gain (panel_apply).  Kept, tested, as the record of the experiment
    int64_t persist_apply_launches = 0;   // diagnostics
    int panel_apply = 1;         // Arnoldi steps enqueued ahead on the panel kernel, value-free 5-point stencil (even line length, phase 0): the kernel forms w = A v itself
    kk_sweep_apply sweep_apply;  // ... the pending request (see kk_sweep_apply)
    bool sweep_apply_fused = false;   // ... the last pass_mgs_strict_sweeps honoured one inside its launch (the work vector was never written by an apply of its own)
    int64_t panel_apply_launches = 0;   // diagnostics
    int xsync = 1;               // row-sharded context: persistent kernels with the in-kernel cross-rank reduction where the communicator offers it (0: RCCL all-reduce per inner-product batch, low-sync route)
    int mgs_persist = 1;         // strict MGS sweeps through the persistent register-resident kernel when the vector fits
    int persist_coop = 0;        // launch the persistent kernels through hipLaunchCooperativeKernel (1) or as ordinary launches (0): see kk_launch_resident
    int persist_threads = 512;   // threads per block (= per CU) of that kernel: 1024 (<= 40 doubles of w per thread) or 512 (<= 80)
    int persist_nt = 1;          // second read of a basis vector (served by the Infinity Cache) with non-temporal loads
    int persist_sync = 0;        // granule layout of that kernel's grid reduction: 0 = one 128-byte line per block (lowest latency on an idle chip), 1 = packed (16 bytes per block: 8 x less sweep traffic)
    int persist_lds = 2;         // park grid-rows of the current basis vector on chip between its two uses: 1 = as many as fit the LDS,
                                 // 2 = those plus KK_PERSIST_NR more in spare registers (512-thread blocks), 0 = none (second read from memory)
    void* d_sync = nullptr;      // device: hand-off granules + error flag of the in-kernel grid reduction (KK_SYNC_BYTES)
    int* h_sync = nullptr;       // pinned: read-back of the error flag
    bool persist_pending = false;  // a persistent launch has not been checked for a barrier timeout yet
    int persist_slot = 0;          // pinned slot its scalars (and completion token) were fetched into
    double persist_token = 0;      // token COUNTER: the last launch enqueued (a launch that committed wrote its token to SC_PERSIST_OK)
    double persist_check_token = 0;  // token of the launch `persist_pending` refers to (a run-ahead launch behind it has its own: la_token)
    unsigned persist_epoch = 0;    // epochs handed out so far: tags of the grid reductions are unique over the life of the context
    bool persist_norm_req = false; // the caller of the sweep wants w / |w| stored (expand!'s scale of the next step, orthonormalize!!)
    bool persist_norm_done = false;  // ... and the last sweep went through a launch that was asked to do so
    int persist_timeouts = 0;      // grid-barrier timeouts recovered so far
    int persist_timeouts_row = 0;  // ... in a row (no clean launch in between): 3 switch the launches to the cooperative API (residency guaranteed by the runtime)
    double persist_timeout_ms = 0; // spin budget of a persistent launch in ms; 0 = by the size of the sweep (kk_persist_timeout_ticks)
    int persist_skip = 0;          // strict sweeps still to run on the launch-per-vector route before the persistent one is retried
    int persist_backoff = 4;       // ... how many after the next timeout (doubles with every timeout in a row)
    int persist_fault = 0;         // test hook (option "persist_fault"): the next N persistent launches time out artificially
    int persist_fault_late = 0;    // test hook (option "persist_fault_late"): in the next N k_mgs_persist launches of a cross-rank context this rank publishes
                                   // its partial of the LAST reduction and then declares the launch lost (a peer that arrives after this rank's patience ran out)
    int64_t spmm_dia_al_launches = 0;    // launches of the aligned sweeping SpMM (diagnostics: "spmm_dia_al_launches")
    int64_t norm_commits_consumed = 0;   // normalised residual columns taken over by scale!!(r, 1 / beta) of a restart without a pass (diagnostics)
    // ---- whole Lanczos step of a short vector in one launch (kk_kernels_fstep.hip; option "fused_step")
    int fused_step = 1;              // CGS2 / low-sync MGS2 Lanczos steps of vectors of at most fused_step_max_rows rows (single rank, ELL-format operator, <= 128 basis vectors)
    int64_t fused_step_max_rows = 131072;   // ... above this the projection pair / the panel kernel are the faster routes (tools/fstep_probe.py, profiles/r06_fstep_probe.jsonl:
                                            // 15 vs 30 us per step at 1 k rows, 17 vs 30 at 10 k, 26 vs 34 at 102 k; 34 vs 38 (MGS2) but 38 vs 32 (CGS2) at 200 k)
    int fstep_blocks = 128;          // blocks of a launch at most (<= KK_FS_MAX_BLOCKS; option "fstep_blocks")
    int fstep_threads = 256;         // threads per block: 256 (1024 measured slower at every length: kept as the record)
    int fused_step_m_limit = 0;      // basis vectors up to which a step takes the one launch: 0 = by vector length (96 - 64 n / 1e5, at least 16), > 0 = fixed, -1 = no limit but KK_FS_MAX_M
    void* d_fsync = nullptr;         // granule area + error flag (KK_FS_SYNC_BYTES + 64)
    unsigned fs_epoch = 0;           // tags of its grid reductions: unique over the life of the context (two per launch)
    double fs_token = 0;             // token counter: a launch that committed stores its token into the pinned host slot
    int fstep_fault = 0;             // test hook (option "fstep_fault"): the next N launches give up at once
    int64_t fstep_launches = 0, fstep_failures = 0;   // diagnostics; a failure (launch that did not commit) switches the route off for the context
    int fold_scale = 1;          // persistent kernel stores r / |r| at its commit when an expand! ends with it (no scale pass in the next step)
    int fuse_passes = 1;         // fuse unproject(pass i) with project(pass i+1)
    int speculate = 1;           // enqueue the next expand's SpMV before syncing the host
    kk_basis spec_owner = nullptr;   // basis whose speculative result currently sits in SC_SPECA / its next column
    uint64_t foreign_gen = 0;        // bumped by every entry point that is handed a slab other than spec_owner's (ctx_foreign_touch)
    struct { bool active = false; kk_op op = nullptr; kk_basis b = nullptr; int c0 = 0, k_next = 0, la_nsweeps = 0; } spec_req;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipEvent_t ev_fetch2 = nullptr; // read-backs of a run-ahead BiCGStab half
    int stage_slot = 0;             // next free coefficient staging slot (stage_coef)
    bool bicg_ahead = false;        // a BiCG half was enqueued by kk_bicgstab_full and not collected yet
    hipEvent_t ev_fetch = nullptr;  // marks the end of the scalar read-backs of an expand (host waits on this, not on the stream)
    hipEvent_t ev_la[2] = {nullptr, nullptr};   // ... of a step whose sweep was enqueued one call ahead (pinned slots 2 and 3)
    int lookahead = 1;              // Lanczos / Arnoldi expand! with a persistent MGS sweep: the NEXT step's apply AND sweep are enqueued before the host waits for this step's scalars
    int prof = 0;                // 0 off, 1 every kernel class, 2 only the basis-streaming classes (project/unproject)
    std::map<std::string, kk_prof_entry> prof_tab;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
    std::vector<hipEvent_t> event_pool;
};

struct kk_basis_s {
    kk_ctx ctx = nullptr;
    uint64_t uid = 0;   // unique over the life of the process (a freed slab's address may be reused)
    int64_t n = 0, ld = 0;
    int cap = 0;
    int64_t ld_agreed = 0;      // max of ld over the ranks of communicator `ld_agreed_comm` (0: not agreed yet)
    uint64_t ld_agreed_comm = 0;
    double* d = nullptr;
    // Gram bookkeeping for mgs_mode=1: gram[i*cap + j] = <b_i, b_j> for j < i < gram_rows
    std::vector<double> gram;
    int gram_c0 = 0;    // first column the Gram rows refer to
    int gram_rows = 0;  // rows [0, gram_rows) of the strictly-lower Gram matrix are valid
    double* d_gram = nullptr;  // device mirror of `gram` (same layout), valid for the same rows
    double* d_gdiag = nullptr; // device only: |b_i|^2 - 1 where a block step measured it (0 elsewhere), rows as d_gram
    // speculative next-step SpMV (hides the host round trip between two expand! calls)
    bool spec_valid = false;
    uint64_t spec_gen = 0;       // ctx->foreign_gen when the speculative apply (and what was enqueued behind it) went out
    const void* spec_op = nullptr;
    int spec_c0 = 0, spec_k = 0, spec_dot_mode = 0;
    const double* spec_dot_ptr = nullptr;   // device slot that received the speculative apply's inner product
    double spec_beta = 0;
    // the whole next step (apply + persistent sweep + read-back) enqueued one call ahead: valid only together with spec_valid
    bool la_valid = false;
    int la_k = 0, la_slot = 0, la_nsweeps = 0;
    double la_token = 0;
    int la_kind = 0;          // 0: persistent sweep (la_token), 1: projection-based Lanczos step (CGS2 / low-sync MGS2: la_orth, la_rode), 2: one-launch step (k_lanczos_fstep: la_token = its token)
    int la_orth = 0;
    bool la_rode = false;
    bool la_inside = false;   // the launch enqueued ahead formed A v itself (k_mgs_panel<.., APPLY>): if it is lost, the apply has to be repeated too
    // residual column left NORMALISED by a fused expand! (persistent kernel, w / |w| written at commit): logically the column
    // still holds r = norm_beta * stored; the next expand! of the same factorization takes it as its new basis vector without
    // the scale pass, every other access multiplies it back first (norm_flush)
    int norm_col = -1;
    double norm_beta = 0;
    // pending normalised commit of a residual BLOCK (one-pass BlockLanczos step): columns [tc_k, tc_k + tc_p) hold
    // T = W R1^-1, the residual area at tc_cr still holds A X; W = T R1 is formed on demand (blk_commit_flush, kk_host.h)
    bool tc_valid = false;
    int tc_k = -1, tc_cr = -1, tc_p = 0;
    int tc_skip = 0, tc_streak = 0;   // steps that shall NOT commit / commits settled in a row without one being consumed: a caller that
                                      // looks at the residual block after every step turns each commit into an extra pass
    double tc_R1[256];   // column-major, leading dimension tc_p
    inline double* col(int c) const { return d + (int64_t)c * ld; }
};

struct kk_sparse_dev {  // one direction (A or A') on the device
    int format = -1;    // 0 = ELL (column-major, padded), 1 = CSR, 2 = SELL-64-sigma (sliced ELL, rows sorted by length inside
                        // sigma-row windows), 3 = column-tiled SELL (`tiles`, each of format 2 over all rows)
    int64_t nrows = 0, ncols = 0, nnz = 0;
    // ELL
    int width = 0;
    int64_t ell_ld = 0;
    int32_t* ell_col = nullptr;
    double* ell_val = nullptr;
    // 2-D stencil diagonals next to the ELL arrays (square operators whose column offsets are a subset of
    // {-D, 0, +D} + {-1, 0, +1}: 5-point / 9-point grids): dense diagonals dia_val[slot * dia_ld + row], zero where the
    // matrix has no entry; slots ordered by offset.  Feeds the sweeping multi-column apply (k_spmm_dia).
    int64_t dia_D = 0;             // far offset (grid line length); 0 = no stencil structure detected
    int dia_pts = 0;               // 5 or 9 stored diagonals
    int64_t dia_ld = 0;
    double* dia_val = nullptr;
    // constant-coefficient stencil (detected at upload): every stored entry of diagonal q equals dia_c[q], and an entry is
    // absent exactly where the neighbour falls off its grid line -- the single-column apply then needs neither indices NOR
    // values: 24 N bytes per apply instead of 64 N (k_spmv_dia<.., CONST>).  Row i sits at position (i + dia_phase) % dia_D of its line.
    bool dia_const = false;
    double dia_c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t dia_phase = 0;
    // row-sharded stencil: rows [int_lo, int_hi) (both even) reference local columns only -> diagonal kernels; the boundary
    // strips [0, int_lo) and [int_hi, nrows), whose rows read the ghost buffer, keep the gather kernels
    int64_t int_lo = 0, int_hi = 0;
    // CSR
    int32_t* rowptr = nullptr;
    int32_t* colind = nullptr;
    double* val = nullptr;
    int lanes_per_row = 4;
    // SELL-C-sigma (C = 64 rows per chunk = one wavefront)
    int64_t sell_nchunks = 0;
    int64_t sell_sigma = 0;        // sorting window (rows); == KK_TPB selects the window kernel k_spmv_sellw
    int64_t* sell_off = nullptr;   // [nchunks + 1] element offset of each chunk
    int32_t* sell_perm = nullptr;  // [nchunks * 64] original row of each slot (-1 = padding slot)
    int32_t* sell_col = nullptr;
    double* sell_val = nullptr;
    // column tiles (format 3)
    int ntiles = 0;
    int64_t tile_cols = 0;
    kk_sparse_dev* tiles = nullptr;
    // ghost columns (row-sharded operators)
    int64_t n_local = -1, n_ghost = 0;
    double* ghost = nullptr;
    kk_halo_fn halo = nullptr;   // called with the x vector before every apply (fills `ghost`)
    void* halo_user = nullptr;
    struct kk_halo_plan* plan = nullptr;   // native ghost exchange (kk_csr_create_sharded): grouped ncclSend / ncclRecv
    int64_t bytes = 0;
};
// ghost exchange plan of a row-sharded operator: before every apply the entries x[send_idx] go to their peers and the
// peers' entries land in the ghost buffer (both library-owned), all on the context stream
struct kk_halo_plan {
    std::vector<int64_t> send_counts, recv_counts;   // per peer rank
    int64_t total_send = 0, total_recv = 0;
    int64_t* d_send_idx = nullptr;
    double* d_sendbuf = nullptr;
    double* d_ghost = nullptr;
    double* d_sendbuf_blk = nullptr;   // [16][total_send], [16][n_ghost]: block applies (allocated on first use)
    double* d_ghost_blk = nullptr;
};
// all-gather / reduce-scatter plan of a row-sharded rectangular map (GKL, config 4): the short vectors (length ncols
// of A) are sharded evenly with stride `shard`; A x gathers them into `vfull`, A' u reduce-scatters `zfull`
struct kk_gather_plan {
    int64_t ncols_global = 0, shard = 0, n_local = 0;
    double* vfull = nullptr;   // world * shard
    double* zfull = nullptr;   // world * shard
    double* stage = nullptr;   // shard
};
int kk_halo_exchange(kk_ctx ctx, const kk_sparse_dev& M, const double* x);
// the same for nb vectors X[:, j] at once (one grouped exchange): *G = ghost block, column j at *G + j * *ldg
int kk_halo_exchange_block(kk_ctx ctx, const kk_sparse_dev& M, const double* X, int64_t ldx, int nb, const double** G, int64_t* ldg);

struct kk_host_csr {
    int64_t nrows = 0, ncols = 0;
    std::vector<int64_t> rowptr;
    std::vector<int32_t> col;
    std::vector<double> val;
};

struct kk_op_s {
    kk_ctx ctx = nullptr;
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int flags = 0;
    kk_sparse_dev A, At;
    kk_host_csr hA;  // kept until A' has been built (or never needed)
    bool have_At = false;
    int64_t n_local_cols = -1, n_ghost = 0;
    kk_gather_plan* gather = nullptr;   // row-sharded rectangular map
};

// ---- launch helpers (kk_context.hip)
struct kk_part {  // static even row partition of [0, ld) over nblk blocks
    int nblk;
    int64_t rpb;  // rows per block, multiple of KK_SUB
};
kk_part kk_partition(kk_ctx ctx, int64_t ld);
void kk_prof_begin(kk_ctx ctx, const char* cls);
void kk_prof_end(kk_ctx ctx);
struct kk_prof_scope {
    kk_ctx c;
    bool on;
    kk_prof_scope(kk_ctx ctx, const char* cls) : c(ctx) {
        // the basis-streaming classes: k_project / k_unproject / k_unproj_proj ("k_p", "k_u") and k_mgs_persist
        on = c->prof == 1 || (c->prof == 2 && (cls[2] == 'p' || cls[2] == 'u' || (cls[2] == 'm' && cls[5] == '_' && cls[6] == 'p')));
        if (on) kk_prof_begin(c, cls);
    }
    ~kk_prof_scope() {
        if (on) kk_prof_end(c);
    }
};

// ---- kernel launchers (kk_kernels_*.hip)
// coefficient vector passed by value in the kernarg segment (no H2D copy, scalar loads)
struct kk_coef {
    double v[KK_MAX_M];
};

int kk_launch_spmv(kk_ctx ctx, const kk_sparse_dev& M, const double* x, double* y, int64_t ld_y_rows,
                   const kk_spmv_fuse& f);
int kk_launch_dot(kk_ctx ctx, const double* x, const double* y, int64_t ld, double* out);
int kk_launch_nrm2(kk_ctx ctx, const double* x, int64_t ld, double* out3);  // out3 = {|x|^2, |x|, 1/|x|}
int kk_launch_axpby(kk_ctx ctx, double* y, const double* x, int64_t ld, double a, double b,
                    const double* a_dev, double a_dev_sign, int a_dev_mode);
int kk_launch_scal(kk_ctx ctx, double* x, int64_t ld, double a, const double* a_dev, int rsqrt_mode = 0);
int kk_launch_copy_scal(kk_ctx ctx, double* y, const double* x, int64_t ld, double a);
int kk_launch_fill_random(kk_ctx ctx, double* x, int64_t n, uint64_t seed);
int kk_launch_gather(kk_ctx ctx, const double* x, const int64_t* idx, int64_t count, double* out);
// s = V' * (w - pre_a * pre_vec);  optional second rhs g = V' * rhs2
int kk_launch_project(kk_ctx ctx, const double* V, int64_t ld, int m, const double* w,
                      const double* pre_vec, const double* pre_a_dev, const double* rhs2,
                      double* out_s, double* out_g);
// w_out = beta*w_in + alpha * sum_j coef[j] V_j ; coef from kernarg (coef_host) or device (coef_dev)
// extra: coefficient add_idx gets += *add_dev ; optional |w_out|^2 -> ws[nrm_off] (+sqrt, 1/sqrt)
int kk_launch_unproject(kk_ctx ctx, const double* V, int64_t ld, int m, const double* w_in, double* w_out,
                        const kk_coef* coef_host, const double* coef_dev, double alpha, double beta,
                        int add_idx, const double* add_dev, double* nrm_out3);
int kk_launch_mgs_step(kk_ctx ctx, double* w, int64_t ld, const double* q_prev, const double* s_prev_dev,
                       const double* q_next, double* dot_out, double* nrm_out3);
int kk_launch_basistransform(kk_ctx ctx, double* V, int64_t ld, int m, int n, const double* U_dev);
int kk_launch_givens(kk_ctx ctx, double* q1, double* q2, int64_t ld, double c, double s);
int kk_launch_householder(kk_ctx ctx, double* V, int64_t ld, int m, const kk_coef* v, double beta);
int kk_launch_householder_dev(kk_ctx ctx, double* V, int64_t ld, int m, const double* v_dev, double beta);
int kk_launch_rank1(kk_ctx ctx, double* V, int64_t ld, int m, const double* y, const kk_coef* x, double alpha,
                    double beta);

// ---- block (multi-vector) launchers
int kk_launch_block_gram(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
                         double* C_dev, int ldc);
int kk_launch_block_gram_rs(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
                            double* C_dev, int rs, int cs);
int kk_launch_blk_chol1(kk_ctx ctx, const double* G, int p, double abs_min, double* R1, double* S1, int st, double* flag);
int kk_launch_blk_chol2(kk_ctx ctx, const double* G2, int p, const double* R1, double* B, int ldb, double* S2, double* S3, int st,
                        double* flag);
int kk_launch_blk_fill_m(kk_ctx ctx, const double* M, int ldm, int p, double* S3, int st);
int kk_launch_blk_combine(kk_ctx ctx, double* P, const double* S3, int kn, int nz, int st);
int kk_launch_block_gram2(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, const double* Y2,
                          int64_t ldy2, int q2, int64_t ld, double* C_dev, int rs, double* C2_dev, int rs2, double* C3_dev = nullptr);
int kk_launch_blk_resid_gram(kk_ctx ctx, const double* P, const double* Pc, int st, int kn, int p, const double* GYY,
                             const double* nrm2, double* GW);
int kk_launch_blk_commit_prep(kk_ctx ctx, const double* GWE, const double* GYY, int p, double abs_min, double keep, double* R1, double* S1,
                              int st, double* cflag);
int kk_launch_block_update_commit(kk_ctx ctx, const double* V, int64_t ld, int m, const double* Win, double* Wout, int64_t ldw, double* Tout,
                                  int64_t ldt, int nb, const double* S_dev, double* norms2_dev, const double* cflag_dev,
                                  const double* S1_dev, double* G2_dev);
int kk_launch_blk_gram_rows(kk_ctx ctx, const double* G2, int st, int k, int p, double* gram, int cap, double* gdiag);
int kk_launch_blk_panel_correct(kk_ctx ctx, const double* P, int st, int kn, int p, const double* gram, int cap, double* Pc,
                                const double* gdiag);
int kk_launch_blk_panel_m(kk_ctx ctx, const double* P, int st, int k, int p, double* M, int ldm);
int kk_launch_blk_onepass_check(kk_ctx ctx, const double* P, int st, int kn, int p, const double* nrm2, double eta, double* flag);
int kk_launch_block_gram_tile(kk_ctx ctx, const double* X, int64_t ldx, int p, const double* Yin, int64_t ldy, const double* Z,
                              int64_t ldz, int nz, const double* S_dev, int st, double alpha, double beta, double* Yout,
                              int64_t ldyo, int q, int64_t ld, double* C_dev, int rs, int cs);
// row stride (= kernel width NB) of the coefficient panel handed to kk_launch_block_update for nb right-hand sides
static inline int kk_bu_stride(int nb) { return nb <= 4 ? 4 : (nb <= 8 ? 8 : 16); }
int kk_launch_block_update(kk_ctx ctx, const double* V, int64_t ld, int m, const double* Win, double* Wout, int64_t ldw_in,
                           int64_t ldw_out, int nb, const double* S_dev, double alpha, double beta, double* norms2_dev,
                           const double* skip_dev = nullptr);
int kk_launch_spmm(kk_ctx ctx, const kk_sparse_dev& M, const double* X, int64_t ldx, double* Y, int64_t ldy, int nb);
int kk_launch_unproj_proj(kk_ctx ctx, const double* V, int64_t ld, int m, const double* w_in, double* w_out,
                          const kk_coef* coef_host, const double* coef_dev, double* out_s, double* nrm_out3);
// row-sharded operation: sum `count` doubles at dev_ptr over the ranks, in place, on the context stream -- through the
// library's own RCCL communicator (kk_comm_init) or the caller's hook (kk_ctx_set_allreduce); no-op otherwise
int kk_comm_allreduce_sum(kk_ctx ctx, double* dev_ptr, int64_t count);
static inline bool kk_sharded(kk_ctx ctx) {
    return !ctx->ar_suspend && ((ctx->comm && ctx->comm->active) || ctx->allreduce);
}
// the persistent kernels of this (row-sharded) context reduce over the ranks inside the launch
static inline bool kk_xs_on(kk_ctx ctx) {
    return ctx->xsync && !ctx->allreduce && ctx->comm && ctx->comm->active && ctx->comm->xs_active;
}
// the vector length the ROUTE of a sweep is decided with: on a cross-rank context the longest shard of the slab (same on every rank)
static inline int64_t kk_dec_ld(kk_ctx ctx, int64_t ld) {
    return (kk_xs_on(ctx) && ctx->dec_ld_local == ld && ctx->dec_ld >= ld) ? ctx->dec_ld : ld;
}
// Does the in-kernel sum over the ranks PAY for a sweep of `nvec` vector-steps with `nred` grid reductions?
// Against the low-synchronisation route (one more RCCL all-reduce per step: 2.05 instead of 1.06) a persistent launch pays one
// cross-rank round trip per grid reduction and gains what it gains on ONE chip: nothing at the single-chip threshold `ld_min`,
// `t_sync_us` per vector-step for every further multiple of it (the saved basis traffic grows with the rows, the exposed
// reduction does not; t_sync_us = the measured gain per vector-step and multiple: 6 us for the register-resident kernel -- 786 vs
// 1359 us per iteration at 2.78 x its threshold --, 0.4 us for the panel kernel -- 8 B/row x 250 k rows of saved traffic; 0.58 us per
// vector-step ahead at 2 x, 1.0 at 4 x its threshold).  Both prices were measured by this communicator's hand-shake (kk_comm_init; the slowest rank's figures,
// identical on all ranks):   take the in-kernel route  iff  nred x hop  <=  all-reduce + nvec x t_sync x (ld / ld_min - 1).
// Option "xsync" = 2 skips the rule (tests; A/B runs), 0 switches the in-kernel route off.
static inline bool kk_xs_pays(kk_ctx ctx, int64_t ld, int nvec, int nred, double ld_min, double t_sync_us) {
    if (!kk_xs_on(ctx)) return true;   // (not a cross-rank launch: nothing to weigh)
    if (ctx->xsync >= 2 || ld_min <= 0.0) return true;
    const kk_comm_s* k = ctx->comm;
    const double gain = k->ar_us + (double)nvec * t_sync_us * std::max(0.0, (double)ld / ld_min - 1.0);
    return (double)nred * k->xs_hop_us <= gain;
}
// fills the kernel argument of one persistent launch with `nred` grid reductions (all zero on a single-rank context)
kk_xs_dev kk_xs_launch_args(kk_ctx ctx, unsigned nred);
int kk_launch_xs_selftest(kk_ctx ctx, const kk_xs_dev& xs, int* out_dev);
int kk_launch_xs_timing(kk_ctx ctx, const kk_xs_dev& xs, int nred, long long* out_dev);   // out[0] = ok, out[1] = 100 MHz ticks of reductions 1 .. nred - 1
void kk_xs_postmortem(kk_ctx ctx, const char* where);   // KK_XSYNC_DEBUG=1: what the first block that gave up saw (stderr)   // 4 reductions; *out_dev = 1 on success
struct kk_ar_suspend {   // RAII: local partials only inside the scope
    kk_ctx c; bool prev;
    explicit kk_ar_suspend(kk_ctx ctx) : c(ctx), prev(ctx->ar_suspend) { c->ar_suspend = true; }
    ~kk_ar_suspend() { c->ar_suspend = prev; }
};
static inline int kk_allreduce(kk_ctx ctx, double* dev_ptr, int64_t count) {
    if (count <= 0 || ctx->ar_suspend) return KK_OK;
    if (ctx->comm && ctx->comm->active) return kk_comm_allreduce_sum(ctx, dev_ptr, count);
    if (!ctx->allreduce) return KK_OK;
    int st = ctx->allreduce(ctx->allreduce_user, dev_ptr, count);
    if (st != 0) {
        kk_set_error("all-reduce hook failed with status %d", st);
        return KK_ERR_INVALID;
    }
    return KK_OK;
}
int kk_launch_cg_update(kk_ctx ctx, double* x, const double* p, double* r, const double* q, int64_t ld, double alpha,
                        const double* pq_dev, double* nrm_out3);
// (I + L) s = p on the device (one block, exact forward substitution); optional ride-along Gram row
// and the Lanczos alpha0 folded into the last coefficient.  See kk_kernels_stream.hip.
int kk_launch_bicg_p(kk_ctx ctx, double* p_out, const double* p, const double* r, const double* v, int64_t ld,
                     const double* sc);
int kk_launch_set_scalar(kk_ctx ctx, double* dst, double v);
int kk_launch_bicg_s(kk_ctx ctx, double* s, const double* r, const double* v, int64_t ld, double* sc, double* nrm_out3);
int kk_launch_bicg_xr(kk_ctx ctx, double* x, const double* p, const double* s, const double* t, double* r,
                      const double* rs, int64_t ld, double* sc, double* nrm_out3, double* rho_out);
int kk_launch_lsmr_u(kk_ctx ctx, const double* av, double* ah, double* u, int64_t ld, double c, double alpha,
                     double* nrm_out3);
int kk_launch_lsmr_hx(kk_ctx ctx, double* h, double* hbar, double* x, const double* v, int64_t ld, double c1, double c2,
                      double c3);
bool kk_mgs_persist_eligible(kk_ctx ctx, int64_t ld, int m, int nsweeps);
int kk_launch_resident(kk_ctx ctx, const void* fn, int threads, void** args, size_t dyn_lds, const char* what);
long long kk_persist_timeout_ticks(kk_ctx ctx, int64_t ld, int nvec, bool cross_rank);
bool kk_mgs_panel_eligible(kk_ctx ctx, int64_t ld);
int64_t kk_mgs_panel_capacity(kk_ctx ctx);
int kk_mgs_panel_width(kk_ctx ctx, int64_t ld, bool strict);
int kk_launch_mgs_panel(kk_ctx ctx, const double* V, int64_t ld, int m, int nsweeps, double* w, const double* carry_q,
                        const double* carry_s, double* out_s, int out_stride, double* nrm_out3, bool normalize_w, bool strict,
                        const kk_sweep_apply* apply = nullptr);
bool kk_sweep_apply_ok(kk_ctx ctx, const kk_sparse_dev& M, int64_t ld);   // can the panel kernel apply this operator itself?
// does an MGS-family sweep over vectors of leading dimension ld run in the low-synchronisation (projection-based) form?
// (option "mgs_mode").  auto: the persistent kernels wherever they are the faster route -- the panel kernel for vectors of
// panel_min_rows .. 4.19 M rows, the register-resident strict kernel from persist_min_rows rows up to its capacity -- and the
// projection pair otherwise (tiny vectors: launch-bound; vectors beyond the register file; row-sharded contexts).
static inline bool kk_mgs_lowsync(kk_ctx ctx, int64_t ld_local, int m) {
    if (ctx->mgs_mode != 2) return ctx->mgs_mode == 1;
    // (the thresholds were measured on the whole chip: what decides is the length of a CU's share of the vector, so a context
    // that owns fewer CUs -- option "num_cus" -- scales them down with it.  Cross-rank context: the longest shard decides, and
    // the in-kernel reduction has to pay for its round trips -- kk_xs_pays)
    const int64_t ld = kk_dec_ld(ctx, ld_local);
    const double share = (double)ctx->num_cus / (double)(ctx->dev_cus > 0 ? ctx->dev_cus : 256);   // (the thresholds were measured on the whole chip)
    if ((double)ld >= share * (double)ctx->panel_min_rows && kk_mgs_panel_eligible(ctx, ld_local)) {
        const int P = kk_mgs_panel_width(ctx, ld_local, false);
        // (gain per vector-step and multiple of the threshold = the basis traffic the panel kernel saves AT the threshold: 8 bytes per row x
        //  250 k rows at ~5 TB/s = 0.4 us -- profiles/r05_panel_sweep_par.jsonl: 0.58 us per vector-step ahead at 500 k rows, 1.0 at 1 M)
        if (kk_xs_pays(ctx, ld, 2 * m, (2 * m + P - 1) / P + 1, share * (double)ctx->panel_min_rows, 0.4)) return false;
    }
    if ((double)ld >= share * (double)ctx->persist_min_rows && kk_mgs_persist_eligible(ctx, ld_local, m, 2))
        // (gain per vector-step and multiple of the threshold, calibrated on the measured step times: at 10 M rows = 2.78 x the threshold the strict
        //  persistent sweep takes 786 us per iteration, the low-sync pair 1359 us -- 11 us per vector-step over 1.78 multiples = 6 us)
        return !kk_xs_pays(ctx, ld, 2 * m, 2 * m + 1, share * (double)ctx->persist_min_rows, 6.0);
    return true;
}
int kk_launch_mgs_persist(kk_ctx ctx, const double* V, int64_t ld, int m, int nsweeps, double* w, const double* carry_q,
                          const double* carry_s, double* out_s, int out_stride, double* nrm_out3, bool normalize_w, const kk_sweep_apply* apply = nullptr);
bool kk_sweep_apply_ok_persist(kk_ctx ctx, const kk_sparse_dev& M, int64_t ld, int m);   // can k_mgs_persist form w = A v - beta v_prev (+ the alpha dot) itself?
int64_t kk_mgs_persist_capacity(kk_ctx ctx);
// the kernel's own test for the normalised commit, on the host's copy of |w|
static inline bool kk_persist_norm_applies(double nrm) { return nrm > 0.0 && 1.0 / nrm <= 1.79769313486231570815e308; }
int kk_launch_lanczos_coef(kk_ctx ctx, const double* buf, double* L, int cap, int m, int lowsync, double* coef_out, double* res);
// one fused Lanczos step on columns [0, m) of V (v = column m - 1, normalised; v_prev = column m - 2), result column m; kk_kernels_fstep.hip
// (arnoldi: w = A v without the three-term part, npass = 1 / 2 orthogonalisation passes, the column of H at host_out + 8 + KK_FS_MAX_M)
int kk_launch_lanczos_fstep(kk_ctx ctx, const kk_sparse_dev& M, double* V, int64_t ld, int m, bool lowsync, bool cgs_order, const double* bprev_dev,
                            double bprev, double* L, int cap, double* host_out, double token, bool normalize, bool arnoldi = false, int npass = 1);
int64_t kk_fstep_capacity_rows(kk_ctx ctx);
int kk_launch_norm_scalars(kk_ctx ctx, const double* nrm2, double* sc, double* res2);
int kk_launch_lowsync_solve(kk_ctx ctx, const double* p, const double* g_ride, double* L, int cap, int m, int newest,
                            const double* a0_dev, double* coef_out, double* s_out);
