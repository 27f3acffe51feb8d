// libkrylov_hip.so, C ABI part 1 (include/krylov_hip.h): error reporting, contexts and their options, event
// profiling, basis slabs and the L1 vector verbs.  Host-side C++ only orchestrates kernel launches on one HIP stream;
// there is NO CPU compute fallback anywhere in this library.
#include "kk_host.h"
#include <atomic>

static thread_local std::string g_last_error;

void kk_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
int kk_hip_fail(hipError_t e, const char* what, const char* file, int line) {
    kk_set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return KK_ERR_HIP;
}

// ------------------------------------------------------------------------------------------
// library / context
// ------------------------------------------------------------------------------------------
KK_API int kk_version(void) { return KK_VERSION; }
KK_API const char* kk_last_error(void) { return g_last_error.c_str(); }

KK_API int kk_device_count(int* count) {
    KK_CHECK(count, KK_ERR_INVALID, "kk_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return KK_OK;
}

// device-side resources of a context; on failure the caller destroys the partially built context
static int ctx_allocate(kk_ctx c) {
    KK_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    KK_HIP(hipMalloc(&c->ws_own, WS_TOTAL * sizeof(double)));
    KK_HIP(hipMemset(c->ws_own, 0, WS_TOTAL * sizeof(double)));
    c->ws = c->ws_own;
    KK_HIP(hipMalloc(&c->partials, (size_t)(2 * KK_MAX_M + 8) * KK_MAX_BLOCKS * sizeof(double)));
    KK_HIP(hipHostMalloc(&c->h_pin, 4 * WS_TOTAL * sizeof(double), hipHostMallocDefault));
    memset(c->h_pin, 0, 4 * WS_TOTAL * sizeof(double));   // (the one-launch step's completion tokens are compared against these slots)
    KK_HIP(hipHostMalloc(&c->h_U, (size_t)KK_MAX_M * KK_MAX_M * sizeof(double), hipHostMallocDefault));
    KK_HIP(hipMalloc(&c->blk_own, (size_t)KK_BLK_SCRATCH * sizeof(double)));
    c->blk = c->blk_own;
    KK_HIP(hipHostMalloc(&c->h_blk, (size_t)KK_BLK_SCRATCH * sizeof(double), hipHostMallocDefault));
    KK_HIP(hipEventCreate(&c->t0));
    KK_HIP(hipEventCreate(&c->t1));
    KK_HIP(hipEventCreateWithFlags(&c->ev_fetch, hipEventDisableTiming));
    KK_HIP(hipEventCreateWithFlags(&c->ev_fetch2, hipEventDisableTiming));
    KK_HIP(hipEventCreateWithFlags(&c->ev_la[0], hipEventDisableTiming));
    KK_HIP(hipEventCreateWithFlags(&c->ev_la[1], hipEventDisableTiming));
    KK_HIP(hipMalloc(&c->d_sync, KK_SYNC_BYTES));
    KK_HIP(hipMemset(c->d_sync, 0, KK_SYNC_BYTES));
    KK_HIP(hipMalloc(&c->d_fsync, KK_FS_SYNC_BYTES + 64));
    KK_HIP(hipMemset(c->d_fsync, 0, KK_FS_SYNC_BYTES + 64));
    KK_HIP(hipHostMalloc((void**)&c->h_sync, 64, hipHostMallocDefault));
    c->h_sync[0] = 0;
    int coop = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, c->device) != hipSuccess || !coop) c->mgs_persist = 0;
    return KK_OK;
}

KK_API int kk_ctx_create(int device, kk_ctx* out) {
    KK_CHECK(out, KK_ERR_INVALID, "kk_ctx_create: null out");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        kk_set_error("kk_ctx_create: no HIP device visible; libkrylov_hip has no CPU fallback");
        return KK_ERR_NO_DEVICE;
    }
    KK_CHECK(device >= 0 && device < n, KK_ERR_INVALID, "kk_ctx_create: device %d out of range [0,%d)", device, n);
    KK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    KK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {   // the library carries gfx950 code objects only
        kk_set_error("kk_ctx_create: device %d is %s, not gfx950 (MI355X); libkrylov_hip has no other code path", device,
                     prop.gcnArchName);
        return KK_ERR_NO_DEVICE;
    }
    kk_ctx c = new kk_ctx_s();
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->dev_cus = c->num_cus;
    {
        int xcds = 0;
        if (hipDeviceGetAttribute(&xcds, hipDeviceAttributeNumberOfXccs, device) != hipSuccess) (void)hipGetLastError();
        c->dev_xcds = (xcds >= 1 && c->dev_cus % xcds == 0) ? xcds : 8;
    }
    const int st = ctx_allocate(c);
    if (st != KK_OK) {   // release whatever was created before the failing call
        const std::string msg = kk_last_error();
        kk_ctx_destroy(c);
        kk_set_error("%s", msg.c_str());
        return st;
    }
    if (getenv("KK_BLOCK_MODE")) c->block_mode = atoi(getenv("KK_BLOCK_MODE"));
    const char* env = getenv("KK_BLOCKS_PER_CU");
    if (env && atoi(env) > 0) c->blocks_per_cu = atoi(env);
    env = getenv("KK_MGS_MODE");
    if (env) c->mgs_mode = atoi(env);
    env = getenv("KK_MGS_PERSIST");
    if (env) c->mgs_persist = c->mgs_persist && atoi(env) != 0;
    env = getenv("KK_PERSIST_THREADS");
    if (env && (atoi(env) == 512 || atoi(env) == 1024)) c->persist_threads = atoi(env);
    env = getenv("KK_PERSIST_NT");
    if (env) c->persist_nt = atoi(env) != 0;
    env = getenv("KK_NUM_CUS");   // a process confined to part of the chip (HSA_CU_MASK, a partition mode) says how many CUs it owns
    if (env && atoi(env) >= 1 && atoi(env) <= c->dev_cus) c->num_cus = atoi(env);
    *out = c;
    return KK_OK;
}

KK_API int kk_ctx_destroy(kk_ctx c) {
    if (!c) return KK_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)kk_comm_destroy(c);
    for (auto& p : c->prof_pending) { c->event_pool.push_back(p.second.first); c->event_pool.push_back(p.second.second); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->ev_fetch) (void)hipEventDestroy(c->ev_fetch);
    if (c->ev_fetch2) (void)hipEventDestroy(c->ev_fetch2);
    for (int i = 0; i < 2; ++i) if (c->ev_la[i]) (void)hipEventDestroy(c->ev_la[i]);
    (void)hipFree(c->ws_own);
    (void)hipFree(c->partials);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_U) (void)hipHostFree(c->h_U);
    (void)hipFree(c->blk_own);
    (void)hipFree(c->d_sync);
    (void)hipFree(c->d_fsync);
    if (c->h_sync) (void)hipHostFree(c->h_sync);
    if (c->h_blk) (void)hipHostFree(c->h_blk);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return KK_OK;
}

KK_API int kk_ctx_set_stream(kk_ctx c, void* s) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return KK_OK;
}
KK_API int kk_ctx_get_stream(kk_ctx c, void** s) {
    KK_CHECK(c && s, KK_ERR_INVALID, "null arg");
    *s = (void*)c->stream;
    return KK_OK;
}
KK_API int kk_ctx_sync(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    return KK_OK;
}
KK_API int kk_ctx_set_option(kk_ctx c, const char* key, double value) {
    KK_CHECK(c && key, KK_ERR_INVALID, "null arg");
    if (!strcmp(key, "blocks_per_cu")) {
        KK_CHECK(value >= 1 && value * c->num_cus <= KK_MAX_BLOCKS, KK_ERR_INVALID, "blocks_per_cu out of range");
        c->blocks_per_cu = (int)value;
    } else if (!strcmp(key, "mgs_mode")) {
        KK_CHECK(value == 0 || value == 1 || value == 2, KK_ERR_INVALID, "mgs_mode must be 0 (strict), 1 (lowsync) or 2 (auto)");
        c->mgs_mode = (int)value;
    } else if (!strcmp(key, "speculate")) {
        c->speculate = value != 0;
    } else if (!strcmp(key, "keep_mb")) {
        KK_CHECK(value >= 0 && value <= 4096, KK_ERR_INVALID, "keep_mb out of range");
        c->keep_mb = (int)value;
    } else if (!strcmp(key, "fuse_passes")) {
        c->fuse_passes = value != 0;
    } else if (!strcmp(key, "mgs_persist")) {
        c->mgs_persist = value != 0;
        c->persist_skip = 0;
    } else if (!strcmp(key, "persist_coop")) {
        c->persist_coop = value != 0;
    } else if (!strcmp(key, "persist_timeout_ms")) {
        KK_CHECK(value >= 0 && value <= 60000, KK_ERR_INVALID, "persist_timeout_ms must be in 0 (by sweep size) .. 60000");
        c->persist_timeout_ms = value;
    } else if (!strcmp(key, "num_cus")) {
        // CUs this context may count on (a GPU shared between ranks / jobs: HSA_CU_MASK, CU-masked streams): the persistent
        // kernels launch one block per CU and need all of them resident at once
        KK_CHECK(value >= 1 && value <= c->dev_cus, KK_ERR_INVALID, "num_cus must be in 1..%d", c->dev_cus);
        KK_HIP(hipStreamSynchronize(c->stream));
        c->num_cus = (int)value;
        if (c->comm) c->comm->cus_before = 0;   // (the caller's figure from now on -- to be set alike on every rank of a cross-rank communicator)
    } else if (!strcmp(key, "xsync")) {
        c->xsync = value >= 2 ? 2 : (value != 0 ? 1 : 0);   // 0 off, 1 where it pays (kk_xs_pays: the hand-shake's own timings decide), 2 always; must be set to the same value on every rank
    } else if (!strcmp(key, "lookahead")) {
        c->lookahead = value != 0;
    } else if (!strcmp(key, "mgs_panel")) {
        c->mgs_panel = value != 0;
    } else if (!strcmp(key, "panel_lag")) {
        c->panel_lag = value != 0;
    } else if (!strcmp(key, "panel_width")) {
        KK_CHECK(value == 0 || value == 1 || value == 2 || value == 3, KK_ERR_INVALID, "panel_width must be 0 (by vector length), 1, 2 or 3");
        c->panel_width = (int)value;
    } else if (!strcmp(key, "panel_min_rows")) {
        KK_CHECK(value >= 0, KK_ERR_INVALID, "panel_min_rows must be >= 0");
        c->panel_min_rows = (int64_t)value;
    } else if (!strcmp(key, "persist_fault")) {
        c->persist_fault = (int)value;   // test hook: the next `value` persistent launches behave like a grid-barrier timeout
    } else if (!strcmp(key, "persist_fault_late")) {
        c->persist_fault_late = (int)value;   // test hook (cross-rank contexts): ... give up at the last reduction, partial already published
    } else if (!strcmp(key, "persist_threads")) {
        KK_CHECK(value == 512 || value == 1024, KK_ERR_INVALID, "persist_threads must be 512 or 1024");
        c->persist_threads = (int)value;
    } else if (!strcmp(key, "persist_nt")) {
        c->persist_nt = value != 0;
    } else if (!strcmp(key, "persist_sync")) {
        c->persist_sync = value != 0;
    } else if (!strcmp(key, "persist_min_rows")) {
        KK_CHECK(value >= 0, KK_ERR_INVALID, "persist_min_rows must be >= 0");
        c->persist_min_rows = (int64_t)value;
    } else if (!strcmp(key, "persist_lds")) {
        KK_CHECK(value == 0 || value == 1 || value == 2, KK_ERR_INVALID, "persist_lds must be 0 (off), 1 (LDS) or 2 (LDS + spare registers)");
        c->persist_lds = (int)value;
    } else if (!strcmp(key, "fold_scale")) {
        c->fold_scale = value != 0;   // 0: the persistent kernel stores w itself and every expand! runs its own scale pass (round-3 behaviour)
    } else if (!strcmp(key, "gram_nt")) {
        c->gram_nt = value != 0;
    } else if (!strcmp(key, "spmv_dia")) {
        c->spmv_dia = value != 0;
    } else if (!strcmp(key, "spmv_dia_const")) {
        c->spmv_dia_const = value != 0;
    } else if (!strcmp(key, "spmv_dia_aligned")) {
        c->spmv_dia_aligned = value != 0;
    } else if (!strcmp(key, "persist_apply")) {
        c->persist_apply = value != 0;
    } else if (!strcmp(key, "panel_apply")) {
        c->panel_apply = value != 0;
    } else if (!strcmp(key, "fused_step")) {
        c->fused_step = value != 0;
    } else if (!strcmp(key, "fused_step_max_rows")) {
        KK_CHECK(value >= 0, KK_ERR_INVALID, "fused_step_max_rows must be >= 0");
        c->fused_step_max_rows = (int64_t)value;
    } else if (!strcmp(key, "fused_step_m_limit")) {
        KK_CHECK(value >= -1 && value <= KK_FS_MAX_M, KK_ERR_INVALID, "fused_step_m_limit must be -1 (none), 0 (by vector length) or 1..%d", KK_FS_MAX_M);
        c->fused_step_m_limit = (int)value;
    } else if (!strcmp(key, "fstep_blocks")) {
        KK_CHECK(value >= 8 && value <= KK_FS_MAX_BLOCKS, KK_ERR_INVALID, "fstep_blocks must be in 8..%d", KK_FS_MAX_BLOCKS);
        c->fstep_blocks = (int)value;
    } else if (!strcmp(key, "fstep_threads")) {
        KK_CHECK(value == 256 || value == 512 || value == 1024, KK_ERR_INVALID, "fstep_threads must be 256, 512 or 1024");
        c->fstep_threads = (int)value;
    } else if (!strcmp(key, "fstep_fault")) {
        c->fstep_fault = (int)value;
    } else if (!strcmp(key, "spmv_dia_sw")) {
        KK_CHECK(value == 0 || value == 1 || value == 2, KK_ERR_INVALID, "spmv_dia_sw must be 0 (k_spmv_dia), 1 or 2 (strips per wave of k_spmv_dia_sw)");
        c->spmv_dia_sw = (int)value;
    } else if (!strcmp(key, "spmv_dia_sw_lines")) {
        KK_CHECK(value == 0 || (value >= 2 && value <= 4096), KK_ERR_INVALID, "spmv_dia_sw_lines must be 0 (by size) or in 2..4096");
        c->spmv_dia_sw_lines = (int)value;
    } else if (!strcmp(key, "spmv_dia_pairs")) {
        KK_CHECK(value == 0 || value == 1 || value == 2 || value == 4, KK_ERR_INVALID, "spmv_dia_pairs must be 0 (by size), 1, 2 or 4");
        c->spmv_dia_pairs = (int)value;
    } else if (!strcmp(key, "spmm_dia")) {
        c->spmm_dia = value != 0;
    } else if (!strcmp(key, "spmm_dia_lines")) {
        KK_CHECK(value >= 2 && value <= 4096, KK_ERR_INVALID, "spmm_dia_lines must be in 2..4096");
        c->spmm_dia_lines = (int)value;
    } else if (!strcmp(key, "spmm_dia_al_lines")) {
        KK_CHECK(value >= 1 && value <= 4096, KK_ERR_INVALID, "spmm_dia_al_lines must be in 1..4096");
        c->spmm_dia_al_lines = (int)value;
    } else if (!strcmp(key, "spmm_dia_al")) {
        KK_CHECK(value == 0 || value == 2 || value == 4, KK_ERR_INVALID, "spmm_dia_al must be 0, 2 or 4");
        c->spmm_dia_al = (int)value;
    } else if (!strcmp(key, "spmm_cols")) {
        KK_CHECK(value == 4 || value == 8 || value == 16, KK_ERR_INVALID, "spmm_cols must be 4, 8 or 16");
        c->spmm_cols = (int)value;
    } else if (!strcmp(key, "spmm_rpl")) {
        KK_CHECK(value == 1 || value == 2, KK_ERR_INVALID, "spmm_rpl must be 1 or 2");
        c->spmm_rpl = (int)value;
    } else if (!strcmp(key, "nt_store_rows")) {
        KK_CHECK(value >= 0, KK_ERR_INVALID, "nt_store_rows must be >= 0");
        c->nt_store_rows = (int64_t)value;
    } else if (!strcmp(key, "block_commit")) {
        c->block_commit = value != 0;
    } else if (!strcmp(key, "resid_gram")) {
        c->resid_gram = value != 0;
        c->gw_valid = false;
    } else if (!strcmp(key, "qr_skip_tol")) {
        KK_CHECK(value >= 0 && value <= 1e-8, KK_ERR_INVALID, "qr_skip_tol must be in [0, 1e-8]");
        c->qr_skip_tol = value;
    } else if (!strcmp(key, "gram2_chunk")) {
        KK_CHECK(value == 64 || value == 80 || value == 128, KK_ERR_INVALID, "gram2_chunk must be 64, 80 or 128");
        c->gram2_chunk = (int)value;
    } else if (!strcmp(key, "gram2_pipe")) {
        c->gram2_pipe = value != 0;
    } else if (!strcmp(key, "gram2_bpc")) {
        KK_CHECK(value >= 0 && value <= 8, KK_ERR_INVALID, "gram2_bpc must be in 0..8");
        c->gram2_bpc = (int)value;
    } else if (!strcmp(key, "gram_bpc")) {
        KK_CHECK(value >= 2 && value <= 16, KK_ERR_INVALID, "gram_bpc must be in 2..16");
        c->gram_bpc = (int)value;
    } else if (!strcmp(key, "block_fuse")) {
        KK_CHECK(value >= 0 && value <= 7, KK_ERR_INVALID, "block_fuse is a bit mask: 1 = CholQR2 round 2, 2 = three-term + panel, 4 = one-pass projection");
        c->block_fuse = (int)value;
    } else if (!strcmp(key, "block_async")) {
        c->block_async = value != 0;
    } else if (!strcmp(key, "bu_mfma")) {
        c->bu_mfma = (int)value;
    } else if (!strcmp(key, "bu_prefetch")) {
        KK_CHECK(value == 0 || value == 1 || value == 8 || value == 16 || value == 24, KK_ERR_INVALID, "bu_prefetch must be 0, 1, 8, 16 or 24");
        c->bu_prefetch = (int)value;
    } else if (!strcmp(key, "spmm_bpc")) {
        KK_CHECK(value >= 0 && value <= 16, KK_ERR_INVALID, "spmm_bpc out of range");
        c->spmm_bpc = (int)value;
    } else if (!strcmp(key, "block_mode")) {
        KK_CHECK(value == 0 || value == 1, KK_ERR_INVALID, "block_mode must be 0 (strict) or 1 (panel)");
        c->block_mode = (int)value;
    } else {
        kk_set_error("unknown option '%s'", key);
        return KK_ERR_INVALID;
    }
    return KK_OK;
}
KK_API int kk_ctx_get_option(kk_ctx c, const char* key, double* value) {
    KK_CHECK(c && key && value, KK_ERR_INVALID, "null arg");
    if (!strcmp(key, "blocks_per_cu")) *value = c->blocks_per_cu;
    else if (!strcmp(key, "mgs_mode")) *value = c->mgs_mode;
    else if (!strcmp(key, "num_cus")) *value = c->num_cus;
    else if (!strcmp(key, "device_cus")) *value = c->dev_cus;
    else if (!strcmp(key, "bu_mfma")) *value = c->bu_mfma;
    else if (!strcmp(key, "panel_lag")) *value = c->panel_lag;
    else if (!strcmp(key, "norm_commits_consumed")) *value = (double)c->norm_commits_consumed;
    else if (!strcmp(key, "spmm_dia_al_launches")) *value = (double)c->spmm_dia_al_launches;
    else if (!strcmp(key, "persist_timeout_ms")) *value = c->persist_timeout_ms;
    else if (!strcmp(key, "xsync")) *value = c->xsync;
    else if (!strcmp(key, "xsync_active")) *value = kk_xs_on(c) ? 1 : 0;
    else if (!strcmp(key, "xsync_launches")) *value = c->comm ? (double)c->comm->n_xs_launches : 0.0;
    else if (!strcmp(key, "ranks_on_this_gpu")) *value = c->comm ? c->comm->xs_share : 1;
    else if (!strcmp(key, "xsync_hop_us")) *value = c->comm ? c->comm->xs_hop_us : 0.0;           // one in-kernel cross-rank reduction, measured by kk_comm_init
    else if (!strcmp(key, "comm_allreduce_us")) *value = c->comm ? c->comm->ar_us : 0.0;         // one small RCCL all-reduce on the stream, measured by kk_comm_init
    else if (!strcmp(key, "device_xcds")) *value = c->dev_xcds;
    else if (!strcmp(key, "block_mode")) *value = c->block_mode;
    else if (!strcmp(key, "block_async")) *value = c->block_async;
    else if (!strcmp(key, "block_fuse")) *value = c->block_fuse;
    else if (!strcmp(key, "spmm_bpc")) *value = c->spmm_bpc;
    else if (!strcmp(key, "bu_prefetch")) *value = c->bu_prefetch;
    else if (!strcmp(key, "keep_mb")) *value = c->keep_mb;
    else if (!strcmp(key, "fuse_passes")) *value = c->fuse_passes;
    else if (!strcmp(key, "mgs_persist")) *value = c->mgs_persist;
    else if (!strcmp(key, "persist_threads")) *value = c->persist_threads;
    else if (!strcmp(key, "persist_timeouts")) *value = c->persist_timeouts;
    else if (!strcmp(key, "persist_skip")) *value = c->persist_skip;
    else if (!strcmp(key, "persist_capacity_rows")) *value = (double)kk_mgs_persist_capacity(c);
    else if (!strcmp(key, "fold_scale")) *value = c->fold_scale;
    else if (!strcmp(key, "mgs_panel")) *value = c->mgs_panel;
    else if (!strcmp(key, "lookahead")) *value = c->lookahead;
    else if (!strcmp(key, "persist_coop")) *value = c->persist_coop;
    else if (!strcmp(key, "panel_width")) *value = c->panel_width;
    else if (!strcmp(key, "panel_min_rows")) *value = (double)c->panel_min_rows;
    else if (!strcmp(key, "panel_capacity_rows")) *value = (double)kk_mgs_panel_capacity(c);
    else if (!strcmp(key, "persist_nt")) *value = c->persist_nt;
    else if (!strcmp(key, "persist_lds")) *value = c->persist_lds;
    else if (!strcmp(key, "persist_sync")) *value = c->persist_sync;
    else if (!strcmp(key, "persist_min_rows")) *value = (double)c->persist_min_rows;
    else if (!strcmp(key, "speculate")) *value = c->speculate;
    else if (!strcmp(key, "spmv_dia")) *value = c->spmv_dia;
    else if (!strcmp(key, "spmv_dia_const")) *value = c->spmv_dia_const;
    else if (!strcmp(key, "spmv_dia_aligned")) *value = c->spmv_dia_aligned;
    else if (!strcmp(key, "persist_apply")) *value = c->persist_apply;
    else if (!strcmp(key, "persist_apply_launches")) *value = (double)c->persist_apply_launches;
    else if (!strcmp(key, "panel_apply")) *value = c->panel_apply;
    else if (!strcmp(key, "panel_apply_launches")) *value = (double)c->panel_apply_launches;
    else if (!strcmp(key, "fused_step")) *value = c->fused_step;
    else if (!strcmp(key, "fused_step_max_rows")) *value = (double)c->fused_step_max_rows;
    else if (!strcmp(key, "fused_step_m_limit")) *value = c->fused_step_m_limit;
    else if (!strcmp(key, "fstep_blocks")) *value = c->fstep_blocks;
    else if (!strcmp(key, "fstep_threads")) *value = c->fstep_threads;
    else if (!strcmp(key, "fstep_launches")) *value = (double)c->fstep_launches;
    else if (!strcmp(key, "fstep_failures")) *value = (double)c->fstep_failures;
    else if (!strcmp(key, "spmv_dia_sw")) *value = c->spmv_dia_sw;
    else if (!strcmp(key, "spmv_dia_sw_lines")) *value = c->spmv_dia_sw_lines;
    else if (!strcmp(key, "spmv_dia_sw_launches")) *value = (double)c->spmv_dia_sw_launches;
    else if (!strcmp(key, "spmm_dia")) *value = c->spmm_dia;
    else if (!strcmp(key, "spmm_dia_lines")) *value = c->spmm_dia_lines;
    else if (!strcmp(key, "spmm_dia_al")) *value = c->spmm_dia_al;
    else if (!strcmp(key, "spmm_dia_al_lines")) *value = c->spmm_dia_al_lines;
    else if (!strcmp(key, "spmm_cols")) *value = c->spmm_cols;
    else if (!strcmp(key, "spmm_rpl")) *value = c->spmm_rpl;
    else if (!strcmp(key, "gram_bpc")) *value = c->gram_bpc;
    else if (!strcmp(key, "gram2_pipe")) *value = c->gram2_pipe;
    else if (!strcmp(key, "gram2_bpc")) *value = c->gram2_bpc;
    else if (!strcmp(key, "gram2_chunk")) *value = c->gram2_chunk;
    else if (!strcmp(key, "qr_skip_tol")) *value = c->qr_skip_tol;
    else if (!strcmp(key, "resid_gram")) *value = c->resid_gram;
    else if (!strcmp(key, "block_commit")) *value = c->block_commit;
    else if (!strcmp(key, "nt_store_rows")) *value = (double)c->nt_store_rows;
    else if (!strcmp(key, "block_commits")) *value = (double)c->block_commits;
    else if (!strcmp(key, "last_qr_dev")) *value = c->last_qr_dev;
    else if (!strcmp(key, "gram_nt")) *value = c->gram_nt;
    else {
        kk_set_error("unknown option '%s'", key);
        return KK_ERR_INVALID;
    }
    return KK_OK;
}
KK_API int kk_ctx_set_allreduce(kk_ctx c, kk_allreduce_fn fn, void* user) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    c->allreduce = fn;
    c->allreduce_user = user;
    return KK_OK;
}
KK_API int kk_ctx_workspace_size(kk_ctx c, int64_t* ws_count, int64_t* blk_count) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (ws_count) *ws_count = WS_TOTAL;
    if (blk_count) *blk_count = KK_BLK_SCRATCH;
    return KK_OK;
}
KK_API int kk_ctx_set_workspace(kk_ctx c, void* ws_device, void* blk_device) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    c->ws = ws_device ? (double*)ws_device : c->ws_own;
    c->blk = blk_device ? (double*)blk_device : c->blk_own;
    KK_HIP(hipMemsetAsync(c->ws, 0, WS_TOTAL * sizeof(double), c->stream));
    return KK_OK;
}
KK_API int kk_ctx_timer_start(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipEventRecord(c->t0, c->stream));
    return KK_OK;
}
KK_API int kk_ctx_timer_stop(kk_ctx c, double* ms) {
    KK_CHECK(c && ms, KK_ERR_INVALID, "null arg");
    KK_HIP(hipEventRecord(c->t1, c->stream));
    KK_HIP(hipEventSynchronize(c->t1));
    float f = 0;
    KK_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
    *ms = f;
    return KK_OK;
}

// ---- per-kernel-class event profiling
static hipEvent_t prof_event(kk_ctx c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void prof_resolve(kk_ctx c) {
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->prof_pending) {
        float f = 0;
        if (hipEventElapsedTime(&f, p.second.first, p.second.second) == hipSuccess) {
            auto& e = c->prof_tab[p.first];
            e.ms += f;
            e.launches += 1;
        }
        c->event_pool.push_back(p.second.first);
        c->event_pool.push_back(p.second.second);
    }
    c->prof_pending.clear();
}
void kk_prof_begin(kk_ctx c, const char* cls) {
    hipEvent_t a = prof_event(c), b = prof_event(c);
    (void)hipEventRecord(a, c->stream);
    c->prof_pending.push_back({cls, {a, b}});
}
void kk_prof_end(kk_ctx c) {
    if (c->prof_pending.empty()) return;
    (void)hipEventRecord(c->prof_pending.back().second.second, c->stream);
    if (c->prof_pending.size() > 8192) prof_resolve(c);
}
KK_API int kk_ctx_prof_enable(kk_ctx c, int on) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (!on && c->prof) prof_resolve(c);
    c->prof = (on == 2) ? 2 : (on != 0);
    return KK_OK;
}
KK_API int kk_ctx_prof_reset(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    prof_resolve(c);
    c->prof_tab.clear();
    return KK_OK;
}
KK_API int kk_ctx_prof_get(kk_ctx c, const char* cls, double* total_ms, int64_t* launches) {
    KK_CHECK(c && cls, KK_ERR_INVALID, "null arg");
    prof_resolve(c);
    auto it = c->prof_tab.find(cls);
    if (total_ms) *total_ms = it == c->prof_tab.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->prof_tab.end() ? 0 : it->second.launches;
    return KK_OK;
}

kk_part kk_partition(kk_ctx c, int64_t ld) {
    const int64_t nsub = ld / KK_SUB;
    int64_t target = (int64_t)c->num_cus * c->blocks_per_cu;
    if (target > KK_MAX_BLOCKS) target = KK_MAX_BLOCKS;
    if (target < 1) target = 1;
    const int64_t spb = std::max<int64_t>(1, (nsub + target - 1) / target);
    kk_part p;
    p.rpb = spb * KK_SUB;
    p.nblk = (int)std::max<int64_t>(1, (nsub + spb - 1) / spb);
    return p;
}

// D2H fetch of `count` workspace doubles starting at `off` into pinned slot `slot` (queued; no sync)
int ws_fetch_async(kk_ctx c, int64_t off, int64_t count, int slot) {
    KK_HIP(hipMemcpyAsync(c->h_pin + (int64_t)slot * WS_TOTAL + off, c->ws + off, count * sizeof(double),
                          hipMemcpyDeviceToHost, c->stream));
    return KK_OK;
}
int stream_sync(kk_ctx c) {
    KK_HIP(hipStreamSynchronize(c->stream));
    return KK_OK;
}

// ------------------------------------------------------------------------------------------
// basis slab
// ------------------------------------------------------------------------------------------
KK_API int kk_basis_create(kk_ctx c, int64_t n, int capacity, kk_basis* out) {
    KK_CHECK(c && out, KK_ERR_INVALID, "kk_basis_create: null arg");
    KK_CHECK(n > 0 && capacity > 0, KK_ERR_INVALID, "kk_basis_create: n=%lld capacity=%d", (long long)n, capacity);
    KK_HIP(hipSetDevice(c->device));
    int64_t ld = (n + KK_SUB - 1) / KK_SUB * KK_SUB;
    if (((ld / KK_SUB) & 1) == 0) ld += KK_SUB;  // odd number of 4 KiB row chunks per column: no channel aliasing
    kk_basis b = new kk_basis_s();
    static std::atomic<uint64_t> next_uid{1};
    b->uid = next_uid++;
    b->ctx = c; b->n = n; b->ld = ld; b->cap = capacity;
    size_t bytes = (size_t)ld * capacity * sizeof(double);
    hipError_t e = hipMalloc(&b->d, bytes);
    if (e != hipSuccess) {
        delete b;
        kk_set_error("kk_basis_create: hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return KK_ERR_NOMEM;
    }
    KK_HIP(hipMemsetAsync(b->d, 0, bytes, c->stream));
    *out = b;
    return KK_OK;
}
KK_API int kk_basis_free(kk_basis b) {
    if (!b) return KK_OK;
    (void)hipDeviceSynchronize();  // not the context's stream: finalizers may run after the context is gone
    (void)hipFree(b->d_gram);
    (void)hipFree(b->d_gdiag);
    (void)hipFree(b->d);
    delete b;
    return KK_OK;
}
KK_API int kk_basis_info(kk_basis b, int64_t* n, int64_t* ld, int* capacity, void** dptr) {
    KK_CHECK(b, KK_ERR_INVALID, "null basis");
    if (n) *n = b->n;
    if (ld) *ld = b->ld;
    if (capacity) *capacity = b->cap;
    if (dptr) {
        KK_TRY(norm_flush(b));   // whoever takes the raw pointer sees the residual itself, not its normalised form
        ctx_public_touch(b);     // ... and may write through it: whatever was enqueued ahead for the slab is void
        gram_touch(b, 0);
        *dptr = b->d;
    }
    return KK_OK;
}
KK_API int kk_basis_invalidate_gram(kk_basis b) {
    KK_CHECK(b, KK_ERR_INVALID, "null basis");
    b->gram_rows = 0;
    return KK_OK;
}
KK_API int kk_basis_upload(kk_basis b, int col, const double* host) {
    CHECK_COL_BOUNDS(b, col);   // (validation first: a rejected call must leave the slab's deferred state as it was -- ADVICE r5)
    KK_CHECK(host, KK_ERR_INVALID, "null host pointer");
    norm_discard(b, col);   // (overwritten as a whole)
    CHECK_COL(b, col);
    gram_touch(b, col);
    KK_HIP(hipMemcpyAsync(b->col(col), host, b->n * sizeof(double), hipMemcpyHostToDevice, b->ctx->stream));
    return stream_sync(b->ctx);
}
KK_API int kk_basis_download(kk_basis b, int col, double* host) {
    CHECK_COL_BOUNDS(b, col);
    KK_CHECK(host, KK_ERR_INVALID, "null host pointer");
    CHECK_COL_RO(b, col);
    KK_HIP(hipMemcpyAsync(host, b->col(col), b->n * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
    return stream_sync(b->ctx);
}
KK_API int kk_basis_upload_device(kk_basis b, int col, const void* dptr) {
    CHECK_COL_BOUNDS(b, col);
    KK_CHECK(dptr, KK_ERR_INVALID, "null device pointer");
    norm_discard(b, col);
    CHECK_COL(b, col);
    gram_touch(b, col);
    KK_HIP(hipMemcpyAsync(b->col(col), dptr, b->n * sizeof(double), hipMemcpyDeviceToDevice, b->ctx->stream));
    return KK_OK;
}
KK_API int kk_basis_download_device(kk_basis b, int col, void* dptr) {
    CHECK_COL_BOUNDS(b, col);
    KK_CHECK(dptr, KK_ERR_INVALID, "null device pointer");
    CHECK_COL_RO(b, col);
    KK_HIP(hipMemcpyAsync(dptr, b->col(col), b->n * sizeof(double), hipMemcpyDeviceToDevice, b->ctx->stream));
    return KK_OK;
}

// ------------------------------------------------------------------------------------------
// L1 verbs
// ------------------------------------------------------------------------------------------

KK_API int kk_vec_dot(kk_basis bx, int cx, kk_basis by, int cy, double* out) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    KK_CHECK(out, KK_ERR_INVALID, "null out");
    kk_ctx c = bx->ctx;
    KK_TRY(kk_launch_dot(c, bx->col(cx), by->col(cy), bx->ld, SCP(c, SC_DOT)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_DOT, 1, 0));
    KK_TRY(stream_sync(c));
    *out = *pin(c, WS_SCAL + SC_DOT);
    return KK_OK;
}
KK_API int kk_vec_nrm2(kk_basis bx, int cx, double* out) {
    CHECK_COL(bx, cx);
    KK_CHECK(out, KK_ERR_INVALID, "null out");
    kk_ctx c = bx->ctx;
    KK_TRY(kk_launch_nrm2(c, bx->col(cx), bx->ld, SCP(c, SC_NRM2)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
    KK_TRY(stream_sync(c));
    *out = pin(c, WS_SCAL + SC_NRM2)[1];
    return KK_OK;
}
KK_API int kk_vec_axpby(kk_basis by, int cy, kk_basis bx, int cx, double a, double b) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    gram_touch(by, cy);
    return kk_launch_axpby(by->ctx, by->col(cy), bx->col(cx), by->ld, a, b, nullptr, 1.0, 0);
}
// A residual column left NORMALISED by a fused expand! (norm_col: the slab holds r * (1 / beta), logically r) meets the one
// operation that asks for exactly those bits -- scale!!(r, 1 / beta) of a thick restart (eigsolve/lanczos.jl:111,
// svdsolve.jl:249, arnoldi restarts): the commit is CONSUMED instead of being undone (norm_flush forms (r / beta) * beta, 1-2 ulp
// off r, and costs a pass) and redone.  `a` must be the very double the host forms as 1 / beta.
static inline bool norm_commit_matches(kk_basis b, int col, double a) {
    return b && b->norm_col == col && col >= 0 && col < b->cap && a == 1.0 / b->norm_beta;
}
KK_API int kk_vec_scal(kk_basis bx, int cx, double a) {
    if (norm_commit_matches(bx, cx, a)) {          // in place: the column already IS scale!!(r, 1 / beta) -- no pass at all
        if (bx->tc_valid) KK_TRY(blk_commit_flush(bx));
        bx->norm_col = -1;
        gram_touch(bx, cx);                        // (drops a step enqueued ahead: it took the column for the next basis vector of the OLD factorization)
        ++bx->ctx->norm_commits_consumed;
        return KK_OK;
    }
    CHECK_COL(bx, cx);
    gram_touch(bx, cx);
    return kk_launch_scal(bx->ctx, bx->col(cx), bx->ld, a, nullptr);
}
KK_API int kk_vec_copy_scal(kk_basis by, int cy, kk_basis bx, int cx, double a) {
    if (norm_commit_matches(bx, cx, a) && by && cy >= 0 && cy < by->cap && !(by == bx && cy == cx) && by->ctx == bx->ctx && by->n == bx->n &&
        by->ld == bx->ld) {
        // y = scale!!(r, 1 / beta) from the stored bits: one copy (x * 1.0 is x), the source stays as it is -- normalised, logically r
        if (by != bx) KK_TRY(norm_flush(by));
        else if (by->tc_valid) KK_TRY(blk_commit_flush(by));
        gram_touch(by, cy);
        ++bx->ctx->norm_commits_consumed;
        return kk_launch_copy_scal(by->ctx, by->col(cy), bx->col(cx), by->ld, 1.0);
    }
    CHECK_COL_BOUNDS(bx, cx); CHECK_COL_BOUNDS(by, cy); CHECK_SAME(bx, by);   // (validation before any side effect)
    // x = 1 * x in place: nothing to do (a start block handed over as columns of the slab it already lives in, blocklanczos.jl:159-166
    // -- the bench's 16 start vectors used to cost 16 read + write passes per sweep here)
    if (by == bx && cy == cx && a == 1.0) { CHECK_COL_RO(bx, cx); return KK_OK; }
    if (!(by == bx && cy == cx)) norm_discard(by, cy);   // the destination is overwritten as a whole: no point in settling it first
    CHECK_COL(bx, cx); CHECK_COL(by, cy);
    gram_touch(by, cy);
    return kk_launch_copy_scal(by->ctx, by->col(cy), bx->col(cx), by->ld, a);
}
KK_API int kk_vec_zero(kk_basis bx, int cx) {
    CHECK_COL_BOUNDS(bx, cx);
    norm_discard(bx, cx);
    CHECK_COL(bx, cx);
    gram_touch(bx, cx);
    KK_HIP(hipMemsetAsync(bx->col(cx), 0, bx->ld * sizeof(double), bx->ctx->stream));
    return KK_OK;
}
KK_API int kk_vec_fill_random(kk_basis bx, int cx, uint64_t seed) {
    CHECK_COL_BOUNDS(bx, cx);
    norm_discard(bx, cx);
    CHECK_COL(bx, cx);
    gram_touch(bx, cx);
    return kk_launch_fill_random(bx->ctx, bx->col(cx), bx->n, seed);
}

