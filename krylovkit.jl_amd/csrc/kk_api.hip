// C ABI of libkrylov_hip.so (see include/krylov_hip.h): contexts, basis slabs, sparse operators,
// the L1/L2 verbs and the fused L3 expand! steps.  Host-side C++ only orchestrates kernel
// launches on one HIP stream; there is NO CPU compute fallback anywhere in this file.
#include "kk_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>

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

static const double KK_EPS = std::numeric_limits<double>::epsilon();
#define WSP(c, off) ((c)->ws + (off))
#define SCP(c, slot) ((c)->ws + WS_SCAL + (slot))

// ------------------------------------------------------------------------------------------
// library / context
// ------------------------------------------------------------------------------------------
extern "C" int kk_version(void) { return KK_VERSION; }
extern "C" const char* kk_last_error(void) { return g_last_error.c_str(); }

extern "C" int kk_device_count(int* count) {
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

extern "C" int kk_ctx_create(int device, kk_ctx* out) {
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
    kk_ctx c = new kk_ctx_s();
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    KK_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    KK_HIP(hipMalloc(&c->ws_own, WS_TOTAL * sizeof(double)));
    KK_HIP(hipMemset(c->ws_own, 0, WS_TOTAL * sizeof(double)));
    c->ws = c->ws_own;
    KK_HIP(hipMalloc(&c->partials, (size_t)(2 * KK_MAX_M + 8) * KK_MAX_BLOCKS * sizeof(double)));
    KK_HIP(hipHostMalloc(&c->h_pin, 4 * WS_TOTAL * sizeof(double), hipHostMallocDefault));
    KK_HIP(hipHostMalloc(&c->h_U, (size_t)KK_MAX_M * KK_MAX_M * sizeof(double), hipHostMallocDefault));
    KK_HIP(hipMalloc(&c->blk_own, (size_t)KK_BLK_SCRATCH * sizeof(double)));
    c->blk = c->blk_own;
    KK_HIP(hipHostMalloc(&c->h_blk, (size_t)KK_BLK_SCRATCH * sizeof(double), hipHostMallocDefault));
    if (getenv("KK_BLOCK_MODE")) c->block_mode = atoi(getenv("KK_BLOCK_MODE"));
    KK_HIP(hipEventCreate(&c->t0));
    KK_HIP(hipEventCreate(&c->t1));
    KK_HIP(hipEventCreateWithFlags(&c->ev_fetch, hipEventDisableTiming));
    KK_HIP(hipEventCreateWithFlags(&c->ev_fetch2, hipEventDisableTiming));
    const char* env = getenv("KK_BLOCKS_PER_CU");
    if (env && atoi(env) > 0) c->blocks_per_cu = atoi(env);
    env = getenv("KK_MGS_MODE");
    if (env) c->mgs_mode = atoi(env);
    *out = c;
    return KK_OK;
}

extern "C" int kk_ctx_destroy(kk_ctx c) {
    if (!c) return KK_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->prof_pending) { c->event_pool.push_back(p.second.first); c->event_pool.push_back(p.second.second); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->t0);
    (void)hipEventDestroy(c->t1);
    (void)hipEventDestroy(c->ev_fetch);
    (void)hipEventDestroy(c->ev_fetch2);
    (void)hipFree(c->ws_own);
    (void)hipFree(c->partials);
    (void)hipHostFree(c->h_pin);
    (void)hipHostFree(c->h_U);
    (void)hipFree(c->blk_own);
    (void)hipHostFree(c->h_blk);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return KK_OK;
}

extern "C" int kk_ctx_set_stream(kk_ctx c, void* s) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return KK_OK;
}
extern "C" int kk_ctx_get_stream(kk_ctx c, void** s) {
    KK_CHECK(c && s, KK_ERR_INVALID, "null arg");
    *s = (void*)c->stream;
    return KK_OK;
}
extern "C" int kk_ctx_sync(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    return KK_OK;
}
extern "C" int kk_ctx_set_option(kk_ctx c, const char* key, double value) {
    KK_CHECK(c && key, KK_ERR_INVALID, "null arg");
    if (!strcmp(key, "blocks_per_cu")) {
        KK_CHECK(value >= 1 && value * c->num_cus <= KK_MAX_BLOCKS, KK_ERR_INVALID, "blocks_per_cu out of range");
        c->blocks_per_cu = (int)value;
    } else if (!strcmp(key, "mgs_mode")) {
        KK_CHECK(value == 0 || value == 1, KK_ERR_INVALID, "mgs_mode must be 0 (strict) or 1 (lowsync)");
        c->mgs_mode = (int)value;
    } else if (!strcmp(key, "speculate")) {
        c->speculate = value != 0;
    } else if (!strcmp(key, "keep_mb")) {
        KK_CHECK(value >= 0 && value <= 4096, KK_ERR_INVALID, "keep_mb out of range");
        c->keep_mb = (int)value;
    } else if (!strcmp(key, "fuse_passes")) {
        c->fuse_passes = value != 0;
    } else if (!strcmp(key, "block_mode")) {
        KK_CHECK(value == 0 || value == 1, KK_ERR_INVALID, "block_mode must be 0 (strict) or 1 (panel)");
        c->block_mode = (int)value;
    } else {
        kk_set_error("unknown option '%s'", key);
        return KK_ERR_INVALID;
    }
    return KK_OK;
}
extern "C" int kk_ctx_get_option(kk_ctx c, const char* key, double* value) {
    KK_CHECK(c && key && value, KK_ERR_INVALID, "null arg");
    if (!strcmp(key, "blocks_per_cu")) *value = c->blocks_per_cu;
    else if (!strcmp(key, "mgs_mode")) *value = c->mgs_mode;
    else if (!strcmp(key, "num_cus")) *value = c->num_cus;
    else if (!strcmp(key, "block_mode")) *value = c->block_mode;
    else if (!strcmp(key, "keep_mb")) *value = c->keep_mb;
    else if (!strcmp(key, "fuse_passes")) *value = c->fuse_passes;
    else if (!strcmp(key, "speculate")) *value = c->speculate;
    else {
        kk_set_error("unknown option '%s'", key);
        return KK_ERR_INVALID;
    }
    return KK_OK;
}
extern "C" int kk_ctx_set_allreduce(kk_ctx c, kk_allreduce_fn fn, void* user) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    c->allreduce = fn;
    c->allreduce_user = user;
    return KK_OK;
}
extern "C" int kk_ctx_workspace_size(kk_ctx c, int64_t* ws_count, int64_t* blk_count) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (ws_count) *ws_count = WS_TOTAL;
    if (blk_count) *blk_count = KK_BLK_SCRATCH;
    return KK_OK;
}
extern "C" int kk_ctx_set_workspace(kk_ctx c, void* ws_device, void* blk_device) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipStreamSynchronize(c->stream));
    c->ws = ws_device ? (double*)ws_device : c->ws_own;
    c->blk = blk_device ? (double*)blk_device : c->blk_own;
    KK_HIP(hipMemsetAsync(c->ws, 0, WS_TOTAL * sizeof(double), c->stream));
    return KK_OK;
}
extern "C" int kk_ctx_timer_start(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    KK_HIP(hipEventRecord(c->t0, c->stream));
    return KK_OK;
}
extern "C" int kk_ctx_timer_stop(kk_ctx c, double* ms) {
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
extern "C" int kk_ctx_prof_enable(kk_ctx c, int on) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (!on && c->prof) prof_resolve(c);
    c->prof = (on == 2) ? 2 : (on != 0);
    return KK_OK;
}
extern "C" int kk_ctx_prof_reset(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    prof_resolve(c);
    c->prof_tab.clear();
    return KK_OK;
}
extern "C" int kk_ctx_prof_get(kk_ctx c, const char* cls, double* total_ms, int64_t* launches) {
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
static int ws_fetch_async(kk_ctx c, int64_t off, int64_t count, int slot) {
    KK_HIP(hipMemcpyAsync(c->h_pin + (int64_t)slot * WS_TOTAL + off, c->ws + off, count * sizeof(double),
                          hipMemcpyDeviceToHost, c->stream));
    return KK_OK;
}
static inline const double* pin(kk_ctx c, int64_t off, int slot = 0) { return c->h_pin + (int64_t)slot * WS_TOTAL + off; }
static int stream_sync(kk_ctx c) {
    KK_HIP(hipStreamSynchronize(c->stream));
    return KK_OK;
}

// ------------------------------------------------------------------------------------------
// basis slab
// ------------------------------------------------------------------------------------------
extern "C" int kk_basis_create(kk_ctx c, int64_t n, int capacity, kk_basis* out) {
    KK_CHECK(c && out, KK_ERR_INVALID, "kk_basis_create: null arg");
    KK_CHECK(n > 0 && capacity > 0, KK_ERR_INVALID, "kk_basis_create: n=%lld capacity=%d", (long long)n, capacity);
    KK_HIP(hipSetDevice(c->device));
    int64_t ld = (n + KK_SUB - 1) / KK_SUB * KK_SUB;
    if (((ld / KK_SUB) & 1) == 0) ld += KK_SUB;  // odd number of 4 KiB row chunks per column: no channel aliasing
    kk_basis b = new kk_basis_s();
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
extern "C" int kk_basis_free(kk_basis b) {
    if (!b) return KK_OK;
    (void)hipDeviceSynchronize();  // not the context's stream: finalizers may run after the context is gone
    (void)hipFree(b->d_gram);
    (void)hipFree(b->d);
    delete b;
    return KK_OK;
}
extern "C" int kk_basis_info(kk_basis b, int64_t* n, int64_t* ld, int* capacity, void** dptr) {
    KK_CHECK(b, KK_ERR_INVALID, "null basis");
    if (n) *n = b->n;
    if (ld) *ld = b->ld;
    if (capacity) *capacity = b->cap;
    if (dptr) *dptr = b->d;
    return KK_OK;
}
#define CHECK_COL(b, c) KK_CHECK((b) && (c) >= 0 && (c) < (b)->cap, KK_ERR_INVALID, "%s: column %d out of range", __func__, (c))
static inline void gram_touch(kk_basis b, int col) {
    if (col < b->gram_rows) b->gram_rows = col;
    b->spec_valid = false;  // any mutation of the slab cancels a speculative next-step SpMV
}
extern "C" int kk_basis_invalidate_gram(kk_basis b) {
    KK_CHECK(b, KK_ERR_INVALID, "null basis");
    b->gram_rows = 0;
    return KK_OK;
}
extern "C" int kk_basis_upload(kk_basis b, int col, const double* host) {
    CHECK_COL(b, col);
    KK_CHECK(host, KK_ERR_INVALID, "null host pointer");
    gram_touch(b, col);
    KK_HIP(hipMemcpyAsync(b->col(col), host, b->n * sizeof(double), hipMemcpyHostToDevice, b->ctx->stream));
    return stream_sync(b->ctx);
}
extern "C" int kk_basis_download(kk_basis b, int col, double* host) {
    CHECK_COL(b, col);
    KK_CHECK(host, KK_ERR_INVALID, "null host pointer");
    KK_HIP(hipMemcpyAsync(host, b->col(col), b->n * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
    return stream_sync(b->ctx);
}
extern "C" int kk_basis_upload_device(kk_basis b, int col, const void* dptr) {
    CHECK_COL(b, col);
    gram_touch(b, col);
    KK_HIP(hipMemcpyAsync(b->col(col), dptr, b->n * sizeof(double), hipMemcpyDeviceToDevice, b->ctx->stream));
    return KK_OK;
}
extern "C" int kk_basis_download_device(kk_basis b, int col, void* dptr) {
    CHECK_COL(b, col);
    KK_HIP(hipMemcpyAsync(dptr, b->col(col), b->n * sizeof(double), hipMemcpyDeviceToDevice, b->ctx->stream));
    return KK_OK;
}

// ------------------------------------------------------------------------------------------
// L1 verbs
// ------------------------------------------------------------------------------------------
#define CHECK_SAME(bx, by) KK_CHECK((bx)->ctx == (by)->ctx && (bx)->n == (by)->n && (bx)->ld == (by)->ld, KK_ERR_DIM, "%s: vector length mismatch (%lld vs %lld)", __func__, (long long)(bx)->n, (long long)(by)->n)

extern "C" int kk_vec_dot(kk_basis bx, int cx, kk_basis by, int cy, double* out) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    KK_CHECK(out, KK_ERR_INVALID, "null out");
    kk_ctx c = bx->ctx;
    KK_TRY(kk_launch_dot(c, bx->col(cx), by->col(cy), bx->ld, SCP(c, SC_DOT)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_DOT, 1, 0));
    KK_TRY(stream_sync(c));
    *out = *pin(c, WS_SCAL + SC_DOT);
    return KK_OK;
}
extern "C" int kk_vec_nrm2(kk_basis bx, int cx, double* out) {
    CHECK_COL(bx, cx);
    KK_CHECK(out, KK_ERR_INVALID, "null out");
    kk_ctx c = bx->ctx;
    KK_TRY(kk_launch_nrm2(c, bx->col(cx), bx->ld, SCP(c, SC_NRM2)));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
    KK_TRY(stream_sync(c));
    *out = pin(c, WS_SCAL + SC_NRM2)[1];
    return KK_OK;
}
extern "C" int kk_vec_axpby(kk_basis by, int cy, kk_basis bx, int cx, double a, double b) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    gram_touch(by, cy);
    return kk_launch_axpby(by->ctx, by->col(cy), bx->col(cx), by->ld, a, b, nullptr, 1.0, 0);
}
extern "C" int kk_vec_scal(kk_basis bx, int cx, double a) {
    CHECK_COL(bx, cx);
    gram_touch(bx, cx);
    return kk_launch_scal(bx->ctx, bx->col(cx), bx->ld, a, nullptr);
}
extern "C" int kk_vec_copy_scal(kk_basis by, int cy, kk_basis bx, int cx, double a) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    gram_touch(by, cy);
    return kk_launch_copy_scal(by->ctx, by->col(cy), bx->col(cx), by->ld, a);
}
extern "C" int kk_vec_zero(kk_basis bx, int cx) {
    CHECK_COL(bx, cx);
    gram_touch(bx, cx);
    KK_HIP(hipMemsetAsync(bx->col(cx), 0, bx->ld * sizeof(double), bx->ctx->stream));
    return KK_OK;
}
extern "C" int kk_vec_fill_random(kk_basis bx, int cx, uint64_t seed) {
    CHECK_COL(bx, cx);
    gram_touch(bx, cx);
    return kk_launch_fill_random(bx->ctx, bx->col(cx), bx->n, seed);
}

// ------------------------------------------------------------------------------------------
// operators
// ------------------------------------------------------------------------------------------
static void free_sparse(kk_sparse_dev& M) {
    (void)hipFree(M.ell_col); (void)hipFree(M.ell_val);
    (void)hipFree(M.rowptr); (void)hipFree(M.colind); (void)hipFree(M.val);
    (void)hipFree(M.sell_off); (void)hipFree(M.sell_perm); (void)hipFree(M.sell_col); (void)hipFree(M.sell_val);
    for (int t = 0; t < M.ntiles; ++t) free_sparse(M.tiles[t]);
    delete[] M.tiles;
    M = kk_sparse_dev();
}

// SELL-64-sigma image of a host CSR matrix on the device (fills the sell_* fields of M, sets format = 2)
static int build_sell(const kk_host_csr& h, kk_sparse_dev& M, int64_t sigma = 64 * 64) {
    const int64_t nrows = h.nrows;
    // SELL-64-sigma: sort rows by length inside windows of sigma rows, slice into chunks of 64
    M.format = 2;
    M.sell_sigma = sigma;
    const int64_t C = 64;
    const int64_t nchunks = (nrows + C - 1) / C;
    std::vector<int32_t> perm((size_t)nchunks * C, -1);
    std::vector<int32_t> order(nrows);
    for (int64_t i = 0; i < nrows; ++i) order[i] = (int32_t)i;
    for (int64_t w0 = 0; w0 < nrows; w0 += sigma) {
        const int64_t w1 = std::min(nrows, w0 + sigma);
        std::stable_sort(order.begin() + w0, order.begin() + w1, [&](int32_t a, int32_t b) {
            return (h.rowptr[a + 1] - h.rowptr[a]) > (h.rowptr[b + 1] - h.rowptr[b]);
        });
    }
    for (int64_t i = 0; i < nrows; ++i) perm[i] = order[i];
    std::vector<int64_t> coff(nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) {
        int64_t wmax = 0;
        for (int64_t l = 0; l < C; ++l) {
            const int32_t r = perm[c * C + l];
            if (r >= 0) wmax = std::max(wmax, h.rowptr[r + 1] - h.rowptr[r]);
        }
        coff[c + 1] = coff[c] + wmax * C;
    }
    const int64_t total = coff[nchunks];
    std::vector<int32_t> sc((size_t)std::max<int64_t>(total, 1), 0);
    std::vector<double> sv((size_t)std::max<int64_t>(total, 1), 0.0);
    for (int64_t c = 0; c < nchunks; ++c)
        for (int64_t l = 0; l < C; ++l) {
            const int32_t r = perm[c * C + l];
            if (r < 0) continue;
            int64_t k = 0;
            for (int64_t p = h.rowptr[r]; p < h.rowptr[r + 1]; ++p, ++k) {
                sc[coff[c] + k * C + l] = h.col[p];
                sv[coff[c] + k * C + l] = h.val[p];
            }
        }
    M.sell_nchunks = nchunks;
    KK_HIP(hipMalloc(&M.sell_off, (nchunks + 1) * sizeof(int64_t)));
    KK_HIP(hipMalloc(&M.sell_perm, perm.size() * sizeof(int32_t)));
    KK_HIP(hipMalloc(&M.sell_col, sc.size() * sizeof(int32_t)));
    KK_HIP(hipMalloc(&M.sell_val, sv.size() * sizeof(double)));
    KK_HIP(hipMemcpy(M.sell_off, coff.data(), (nchunks + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    KK_HIP(hipMemcpy(M.sell_perm, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    KK_HIP(hipMemcpy(M.sell_col, sc.data(), sc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    KK_HIP(hipMemcpy(M.sell_val, sv.data(), sv.size() * sizeof(double), hipMemcpyHostToDevice));
    M.bytes = (nchunks + 1) * 8 + perm.size() * 4 + sc.size() * 12;
    return KK_OK;
}

// Column-tiled SELL for operators whose gathers have no locality (rows touching columns all over a vector that does
// not fit the 4 MB L2 of an XCD, e.g. the random rectangular map of the GKL configuration and its transpose): the
// columns are cut into tiles of `tile_cols`, every tile is its own SELL matrix over ALL rows, and an apply runs the
// tiles one after the other accumulating into y, so that each launch gathers from one L2-resident slice of x.
static int build_tiled(const kk_host_csr& h, int64_t tile_cols, kk_sparse_dev& M) {
    const int64_t nrows = h.nrows, nnz = h.rowptr[nrows];
    const int T = (int)((h.ncols + tile_cols - 1) / tile_cols);
    M.format = 3;
    M.ntiles = T;
    M.tiles = new kk_sparse_dev[T];
    M.tile_cols = tile_cols;
    // pass 1: entries per (tile, row); pass 2: scatter into per-tile CSR
    std::vector<kk_host_csr> ht(T);
    for (int t = 0; t < T; ++t) {
        ht[t].nrows = nrows; ht[t].ncols = h.ncols;
        ht[t].rowptr.assign(nrows + 1, 0);
    }
    for (int64_t i = 0; i < nrows; ++i)
        for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) ht[h.col[p] / tile_cols].rowptr[i + 1]++;
    for (int t = 0; t < T; ++t) {
        for (int64_t i = 0; i < nrows; ++i) ht[t].rowptr[i + 1] += ht[t].rowptr[i];
        ht[t].col.resize(ht[t].rowptr[nrows]);
        ht[t].val.resize(ht[t].rowptr[nrows]);
    }
    {
        std::vector<int64_t> cur(T);
        for (int64_t i = 0; i < nrows; ++i) {
            for (int t = 0; t < T; ++t) cur[t] = ht[t].rowptr[i];
            for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) {
                const int t = (int)(h.col[p] / tile_cols);
                const int64_t q = cur[t]++;
                ht[t].col[q] = h.col[p];
                ht[t].val[q] = h.val[p];
            }
        }
    }
    M.bytes = 0;
    for (int t = 0; t < T; ++t) {
        kk_sparse_dev& S = M.tiles[t];
        S.nrows = nrows; S.ncols = h.ncols; S.nnz = ht[t].rowptr[nrows];
        KK_TRY(build_sell(ht[t], S, KK_TPB));   // sigma = the 256 rows of one thread block: k_spmv_sellw
        M.bytes += S.bytes;
        kk_host_csr().rowptr.swap(ht[t].rowptr);
        std::vector<int32_t>().swap(ht[t].col);
        std::vector<double>().swap(ht[t].val);
    }
    (void)nnz;
    return KK_OK;
}

// mean distance between the smallest and the largest column index of a row (sampled): small for stencils / banded
// operators whose gathers are cache friendly as they are, ~ncols for random sparsity
static double mean_row_span(const kk_host_csr& h) {
    const int64_t step = std::max<int64_t>(1, h.nrows / 65536);
    double sum = 0;
    int64_t cnt = 0;
    for (int64_t i = 0; i < h.nrows; i += step) {
        if (h.rowptr[i + 1] == h.rowptr[i]) continue;
        int32_t lo = h.col[h.rowptr[i]], hi = lo;
        for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) { lo = std::min(lo, h.col[p]); hi = std::max(hi, h.col[p]); }
        sum += (double)(hi - lo);
        ++cnt;
    }
    return cnt ? sum / cnt : 0.0;
}

static int upload_sparse(kk_ctx c, const kk_host_csr& h, kk_sparse_dev& M) {
    const int64_t nrows = h.nrows, nnz = h.rowptr[nrows];
    M.nrows = nrows; M.ncols = h.ncols; M.nnz = nnz;
    KK_CHECK(nnz < (int64_t)1 << 31, KK_ERR_UNSUPPORTED, "nnz >= 2^31 not supported (int32 row pointers on device)");
    KK_CHECK(h.ncols < (int64_t)1 << 31 && nrows < (int64_t)1 << 31, KK_ERR_UNSUPPORTED, "dimension >= 2^31 not supported");
    int64_t maxw = 0;
    for (int64_t i = 0; i < nrows; ++i) maxw = std::max(maxw, h.rowptr[i + 1] - h.rowptr[i]);
    const bool force_csr = getenv("KK_SPMV_FORMAT") && !strcmp(getenv("KK_SPMV_FORMAT"), "csr");
    const bool force_ell = getenv("KK_SPMV_FORMAT") && !strcmp(getenv("KK_SPMV_FORMAT"), "ell");
    const bool force_sell = getenv("KK_SPMV_FORMAT") && !strcmp(getenv("KK_SPMV_FORMAT"), "sell");
    const bool ell = !force_csr && !force_sell && (force_ell || (maxw <= 64 && (double)maxw * nrows <= 1.25 * (double)nnz + 4096));
    // column tiling: default tile = 3 MB of the gathered vector (an XCD's L2 is 4 MB; measured optimum on the
    // config-4 operator, flat between 2 and 4 MB); KK_SPMV_TILE_COLS overrides (0 = never)
    int64_t tile_cols = 393216;
    if (const char* tc = getenv("KK_SPMV_TILE_COLS")) tile_cols = atoll(tc);
    const bool fmt_forced = force_csr || force_ell || force_sell;
    if (!fmt_forced && tile_cols > 0 && 2 * h.ncols > 3 * tile_cols && nnz > 0 && mean_row_span(h) > 1.5 * (double)tile_cols)
        return build_tiled(h, tile_cols, M);
    if (ell) {
        M.format = 0;
        M.width = (int)std::max<int64_t>(maxw, 1);
        M.ell_ld = (nrows + 63) / 64 * 64;
        std::vector<int32_t> ec((size_t)M.ell_ld * M.width, 0);
        std::vector<double> ev((size_t)M.ell_ld * M.width, 0.0);
        for (int64_t i = 0; i < nrows; ++i) {
            int k = 0;
            for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p, ++k) {
                ec[(size_t)k * M.ell_ld + i] = h.col[p];
                ev[(size_t)k * M.ell_ld + i] = h.val[p];
            }
        }
        KK_HIP(hipMalloc(&M.ell_col, ec.size() * sizeof(int32_t)));
        KK_HIP(hipMalloc(&M.ell_val, ev.size() * sizeof(double)));
        KK_HIP(hipMemcpy(M.ell_col, ec.data(), ec.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        KK_HIP(hipMemcpy(M.ell_val, ev.data(), ev.size() * sizeof(double), hipMemcpyHostToDevice));
        M.bytes = ec.size() * 4 + ev.size() * 8;
    } else if (!force_csr) {
        KK_TRY(build_sell(h, M));
    } else {
        M.format = 1;
        std::vector<int32_t> rp(nrows + 1);
        for (int64_t i = 0; i <= nrows; ++i) rp[i] = (int32_t)h.rowptr[i];
        KK_HIP(hipMalloc(&M.rowptr, (nrows + 1) * sizeof(int32_t)));
        KK_HIP(hipMalloc(&M.colind, std::max<int64_t>(nnz, 1) * sizeof(int32_t)));
        KK_HIP(hipMalloc(&M.val, std::max<int64_t>(nnz, 1) * sizeof(double)));
        KK_HIP(hipMemcpy(M.rowptr, rp.data(), (nrows + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
        KK_HIP(hipMemcpy(M.colind, h.col.data(), nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        KK_HIP(hipMemcpy(M.val, h.val.data(), nnz * sizeof(double), hipMemcpyHostToDevice));
        const double avg = nrows ? (double)nnz / nrows : 1;
        int L = 2;
        while (L * 2 <= avg && L < 64) L *= 2;
        M.lanes_per_row = L;
        M.bytes = (nrows + 1) * 4 + nnz * 12;
    }
    return KK_OK;
}

static void transpose_csr(const kk_host_csr& a, kk_host_csr& t) {
    t.nrows = a.ncols; t.ncols = a.nrows;
    const int64_t nnz = a.rowptr[a.nrows];
    t.rowptr.assign(t.nrows + 1, 0);
    t.col.resize(nnz); t.val.resize(nnz);
    for (int64_t p = 0; p < nnz; ++p) t.rowptr[a.col[p] + 1]++;
    for (int64_t i = 0; i < t.nrows; ++i) t.rowptr[i + 1] += t.rowptr[i];
    std::vector<int64_t> cur(t.rowptr.begin(), t.rowptr.end() - 1);
    for (int64_t i = 0; i < a.nrows; ++i)
        for (int64_t p = a.rowptr[i]; p < a.rowptr[i + 1]; ++p) {
            const int64_t q = cur[a.col[p]]++;
            t.col[q] = (int32_t)i;
            t.val[q] = a.val[p];
        }
}

extern "C" int kk_csr_create(kk_ctx c, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* rowptr,
                             const int32_t* colind, const double* val, int index_base, int flags, kk_op* out) {
    KK_CHECK(c && out && rowptr && (nnz == 0 || (colind && val)), KK_ERR_INVALID, "kk_csr_create: null arg");
    KK_CHECK(nrows > 0 && ncols > 0 && nnz >= 0, KK_ERR_INVALID, "kk_csr_create: bad dimensions");
    KK_CHECK(index_base == 0 || index_base == 1, KK_ERR_INVALID, "index_base must be 0 or 1");
    KK_CHECK(rowptr[nrows] - index_base == nnz, KK_ERR_DIM, "kk_csr_create: rowptr[nrows] != nnz");
    KK_HIP(hipSetDevice(c->device));
    kk_op op = new kk_op_s();
    op->ctx = c; op->nrows = nrows; op->ncols = ncols; op->nnz = nnz; op->flags = flags;
    kk_host_csr& h = op->hA;
    h.nrows = nrows; h.ncols = ncols;
    h.rowptr.resize(nrows + 1);
    for (int64_t i = 0; i <= nrows; ++i) h.rowptr[i] = rowptr[i] - index_base;
    h.col.resize(nnz); h.val.assign(val, val + nnz);
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t cc = (int64_t)colind[p] - index_base;
        if (cc < 0 || cc >= ncols) {
            delete op;
            kk_set_error("kk_csr_create: column index %lld out of range at entry %lld", (long long)cc, (long long)p);
            return KK_ERR_DIM;
        }
        h.col[p] = (int32_t)cc;
    }
    int s = upload_sparse(c, h, op->A);
    if (s != KK_OK) { free_sparse(op->A); delete op; return s; }
    if (flags & KK_OP_SYMMETRIC) { kk_host_csr().rowptr.swap(h.rowptr); h.col.clear(); h.col.shrink_to_fit(); h.val.clear(); h.val.shrink_to_fit(); }
    *out = op;
    return KK_OK;
}

extern "C" int kk_csc_create(kk_ctx c, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* colptr,
                             const int64_t* rowval, const double* nzval, int index_base, int flags, kk_op* out) {
    KK_CHECK(c && out && colptr && (nnz == 0 || (rowval && nzval)), KK_ERR_INVALID, "kk_csc_create: null arg");
    KK_CHECK(nrows > 0 && ncols > 0 && nnz >= 0, KK_ERR_INVALID, "kk_csc_create: bad dimensions");
    KK_CHECK(index_base == 0 || index_base == 1, KK_ERR_INVALID, "index_base must be 0 or 1");
    KK_CHECK(colptr[ncols] - index_base == nnz, KK_ERR_DIM, "kk_csc_create: colptr[ncols] != nnz");
    KK_HIP(hipSetDevice(c->device));
    // the CSC arrays of A are the CSR arrays of A'
    kk_host_csr ht;
    ht.nrows = ncols; ht.ncols = nrows;
    ht.rowptr.resize(ncols + 1);
    for (int64_t i = 0; i <= ncols; ++i) ht.rowptr[i] = colptr[i] - index_base;
    ht.col.resize(nnz); ht.val.assign(nzval, nzval + nnz);
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t r = rowval[p] - index_base;
        if (r < 0 || r >= nrows) {
            kk_set_error("kk_csc_create: row index %lld out of range at entry %lld", (long long)r, (long long)p);
            return KK_ERR_DIM;
        }
        ht.col[p] = (int32_t)r;
    }
    kk_op op = new kk_op_s();
    op->ctx = c; op->nrows = nrows; op->ncols = ncols; op->nnz = nnz; op->flags = flags;
    int s;
    if (flags & KK_OP_SYMMETRIC) {
        s = upload_sparse(c, ht, op->A);  // A == A'
    } else {
        transpose_csr(ht, op->hA);
        s = upload_sparse(c, op->hA, op->A);
        if (s == KK_OK) {
            s = upload_sparse(c, ht, op->At);
            op->have_At = (s == KK_OK);
            kk_host_csr().rowptr.swap(op->hA.rowptr); op->hA.col.clear(); op->hA.val.clear();
        }
    }
    if (s != KK_OK) { free_sparse(op->A); free_sparse(op->At); delete op; return s; }
    *out = op;
    return KK_OK;
}

extern "C" int kk_op_free(kk_op op) {
    if (!op) return KK_OK;
    (void)hipDeviceSynchronize();  // see kk_basis_free
    free_sparse(op->A);
    free_sparse(op->At);
    delete op;
    return KK_OK;
}

extern "C" int kk_op_info(kk_op op, int64_t* nrows, int64_t* ncols, int64_t* nnz, int* format, int64_t* bytes) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    if (nrows) *nrows = op->nrows;
    if (ncols) *ncols = op->ncols;
    if (nnz) *nnz = op->nnz;
    if (format) *format = op->A.format;
    if (bytes) *bytes = op->A.bytes + op->At.bytes;
    return KK_OK;
}

extern "C" int kk_op_set_ghost(kk_op op, int64_t n_local_cols, int64_t n_ghost, void* device_ghost) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    KK_CHECK(n_local_cols >= 0 && n_ghost >= 0 && n_local_cols + n_ghost == op->ncols, KK_ERR_DIM,
             "kk_op_set_ghost: n_local (%lld) + n_ghost (%lld) != ncols (%lld)", (long long)n_local_cols,
             (long long)n_ghost, (long long)op->ncols);
    KK_CHECK(n_ghost == 0 || device_ghost, KK_ERR_INVALID, "kk_op_set_ghost: null ghost buffer");
    op->A.n_local = n_local_cols;
    op->A.n_ghost = n_ghost;
    op->A.ghost = (double*)device_ghost;  // caller-owned
    return KK_OK;
}

extern "C" int kk_op_set_halo_hook(kk_op op, kk_halo_fn fn, void* user) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    op->A.halo = fn;
    op->A.halo_user = user;
    return KK_OK;
}
extern "C" int kk_gather_ptr(kk_ctx c, const void* x_device, const int64_t* device_idx, int64_t count, void* device_out) {
    KK_CHECK(c && x_device && (count == 0 || (device_idx && device_out)), KK_ERR_INVALID, "kk_gather_ptr: null arg");
    return kk_launch_gather(c, (const double*)x_device, device_idx, count, (double*)device_out);
}

static int get_matrix(kk_op op, int transpose, const kk_sparse_dev** M) {
    if (!transpose || (op->flags & KK_OP_SYMMETRIC)) {
        *M = &op->A;
        return KK_OK;
    }
    if (!op->have_At) {
        KK_CHECK(!op->hA.rowptr.empty(), KK_ERR_INVALID, "transpose requested but host copy is gone");
        kk_host_csr ht;
        transpose_csr(op->hA, ht);
        KK_TRY(upload_sparse(op->ctx, ht, op->At));
        op->have_At = true;
        kk_host_csr().rowptr.swap(op->hA.rowptr); op->hA.col.clear(); op->hA.col.shrink_to_fit(); op->hA.val.clear(); op->hA.val.shrink_to_fit();
    }
    *M = &op->At;
    return KK_OK;
}

// dimension check of y = op(A) x ; with ghosts the x-vector holds the local columns only
static int check_apply(kk_op op, int transpose, kk_basis bx, kk_basis by) {
    const int64_t in = transpose ? op->nrows : (op->A.n_ghost > 0 ? op->A.n_local : op->ncols);
    const int64_t outn = transpose ? op->ncols : op->nrows;
    // ghost-only operator (n_local == 0): every column comes from the caller's gathered buffer, x is unused
    const bool ghost_only = !transpose && op->A.n_ghost > 0 && op->A.n_local == 0;
    KK_CHECK((ghost_only || bx->n == in) && by->n == outn, KK_ERR_DIM, "apply: operator is %lldx%lld%s, x has %lld rows, y has %lld rows",
             (long long)op->nrows, (long long)op->ncols, transpose ? " (adjoint)" : "", (long long)bx->n, (long long)by->n);
    KK_CHECK(bx->ctx == op->ctx && by->ctx == op->ctx, KK_ERR_INVALID, "apply: objects belong to different contexts");
    return KK_OK;
}

extern "C" int kk_spmv(kk_op op, int transpose, kk_basis bx, int cx, kk_basis by, int cy) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    CHECK_COL(bx, cx); CHECK_COL(by, cy);
    KK_TRY(check_apply(op, transpose, bx, by));
    KK_CHECK(!(bx == by && cx == cy), KK_ERR_INVALID, "kk_spmv: x and y must differ");
    const kk_sparse_dev* M;
    KK_TRY(get_matrix(op, transpose, &M));
    gram_touch(by, cy);
    kk_spmv_fuse f;
    return kk_launch_spmv(op->ctx, *M, bx->col(cx), by->col(cy), by->ld, f);
}

extern "C" int kk_spmv_affine(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, double a0, double a1) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    CHECK_COL(bx, cx); CHECK_COL(by, cy);
    KK_TRY(check_apply(op, 0, bx, by));
    KK_CHECK(op->nrows == op->ncols || op->A.n_ghost > 0, KK_ERR_DIM, "affine apply needs a square operator");
    KK_CHECK(!(bx == by && cx == cy), KK_ERR_INVALID, "kk_spmv_affine: x and y must differ");
    gram_touch(by, cy);
    kk_spmv_fuse f;
    f.a0 = a0; f.a1 = a1;
    return kk_launch_spmv(op->ctx, op->A, bx->col(cx), by->col(cy), by->ld, f);
}

// q = a0 p + a1 A p with the fused <p, q> (the CG / short-recurrence apply, linsolve/cg.jl:35-36,61-62)
extern "C" int kk_spmv_affine_dot(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, double a0, double a1, double* dot) {
    KK_CHECK(op && dot, KK_ERR_INVALID, "null arg");
    CHECK_COL(bx, cx); CHECK_COL(by, cy);
    KK_TRY(check_apply(op, 0, bx, by));
    KK_CHECK(!(bx == by && cx == cy), KK_ERR_INVALID, "kk_spmv_affine_dot: x and y must differ");
    kk_ctx c = op->ctx;
    gram_touch(by, cy);
    kk_spmv_fuse f;
    f.a0 = a0; f.a1 = a1;
    f.dot_mode = 2;
    f.dot_out = SCP(c, SC_DOT);
    KK_TRY(kk_launch_spmv(c, op->A, bx->col(cx), by->col(cy), by->ld, f));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_DOT, 1, 0));
    KK_TRY(stream_sync(c));
    *dot = pin(c, WS_SCAL + SC_DOT)[0];
    return KK_OK;
}
static int check_square_op(kk_op op, kk_basis b);
// One CG iteration body (linsolve/cg.jl:60-66) with ONE host synchronisation:
//   [p = r + beta p]  (skipped when beta_is_first)   q = a0 p + a1 A p with fused <p,q> (stays on the device)
//   alpha = rho / <p,q> formed inside the update kernel ; x += alpha p ; r -= alpha q ; |r|
// columns of `b`: cx, cr, cp, cq.  Returns <p,q> and |r|.
extern "C" int kk_cg_iterate(kk_op op, kk_basis b, int cx, int cr, int cp, int cq, double a0, double a1, double beta,
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
extern "C" int kk_cg_update(kk_basis bx, int cx, kk_basis bp, int cp, kk_basis br, int cr, kk_basis bq, int cq, double alpha,
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
static int fetch_mark(kk_ctx c);
static int fetch_wait(kk_ctx c);
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
extern "C" int kk_bicgstab_half(kk_op op, kk_basis b, const int* cols, double a0, double a1, int mode, double rho,
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
extern "C" int kk_bicgstab_full(kk_op op, kk_basis b, const int* cols, double a0, double a1, int redo_t,
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
extern "C" int kk_lsmr_step_u(kk_basis b, int c_av, int c_ah, int c_u, double c, double alpha, double* beta) {
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
extern "C" int kk_lsmr_update(kk_basis b, int ch, int chbar, int cx, kk_basis bv, int cv, double c1, double c2, double c3) {
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

extern "C" int kk_gather(kk_basis bx, int cx, const int64_t* device_idx, int64_t count, void* device_out) {
    CHECK_COL(bx, cx);
    return kk_launch_gather(bx->ctx, bx->col(cx), device_idx, count, (double*)device_out);
}

// ------------------------------------------------------------------------------------------
// L2 basis operations
// ------------------------------------------------------------------------------------------
#define CHECK_RANGE(b, c0, m) KK_CHECK((b) && (c0) >= 0 && (m) >= 0 && (c0) + (m) <= (b)->cap && (m) <= KK_MAX_M, KK_ERR_INVALID, "%s: column range [%d,%d) invalid (capacity %d, max %d per call)", __func__, (c0), (c0) + (m), (b) ? (b)->cap : 0, KK_MAX_M)

extern "C" int kk_project(kk_basis b, int c0, int m, kk_basis bx, int cx, double alpha, double beta, double* y) {
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

extern "C" int kk_unproject(kk_basis by, int cy, kk_basis b, int c0, int m, const double* x, double alpha, double beta) {
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

extern "C" int kk_rank1update(kk_basis b, int c0, int m, kk_basis by, int cy, const double* x, double alpha, double beta) {
    CHECK_RANGE(b, c0, m); CHECK_COL(by, cy); CHECK_SAME(b, by);
    KK_CHECK(x || m == 0, KK_ERR_INVALID, "null x");
    KK_CHECK(!(by == b && cy >= c0 && cy < c0 + m), KK_ERR_INVALID, "kk_rank1update: y aliases a basis column");
    if (m == 0) return KK_OK;
    gram_touch(b, c0);
    kk_coef ch;
    memset(&ch, 0, sizeof(ch));
    for (int j = 0; j < m; ++j) ch.v[j] = x[j];
    return kk_launch_rank1(b->ctx, b->col(c0), b->ld, m, by->col(cy), &ch, alpha, beta);
}

extern "C" int kk_basistransform(kk_basis b, int c0, int m, int n, const double* U, int ldu) {
    CHECK_RANGE(b, c0, m);
    KK_CHECK(U && n >= 0 && n <= m && ldu >= m, KK_ERR_DIM, "kk_basistransform: U must be m x n with n <= m, ldu >= m");
    if (n == 0 || m == 0) return KK_OK;
    kk_ctx c = b->ctx;
    gram_touch(b, c0);
    // pack U (m x n, leading dimension m) into the pinned staging area, then into device scratch
    KK_TRY(stream_sync(c));
    double* hp = c->h_U;  // pinned KK_MAX_M x KK_MAX_M staging
    for (int j = 0; j < n; ++j) memcpy(hp + (size_t)j * m, U + (size_t)j * ldu, m * sizeof(double));
    double* dU = c->partials;  // reuse the partial-sum buffer as U scratch (>= 2 MiB)
    KK_HIP(hipMemcpyAsync(dU, hp, (size_t)m * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return kk_launch_basistransform(c, b->col(c0), b->ld, m, n, dU);
}

extern "C" int kk_givens_rmul(kk_basis b, int i1, int i2, double cc, double s) {
    CHECK_COL(b, i1); CHECK_COL(b, i2);
    KK_CHECK(i1 != i2, KK_ERR_INVALID, "kk_givens_rmul: i1 == i2");
    gram_touch(b, std::min(i1, i2));
    return kk_launch_givens(b->ctx, b->col(i1), b->col(i2), b->ld, cc, s);
}

extern "C" int kk_householder_rmul(kk_basis b, int c0, int m, const double* v, double beta) {
    CHECK_RANGE(b, c0, m);
    KK_CHECK(v || m == 0, KK_ERR_INVALID, "null v");
    if (m == 0 || beta == 0.0) return KK_OK;  // iszero(beta) && return b  (reflector.jl:147)
    gram_touch(b, c0);
    kk_coef ch;
    memset(&ch, 0, sizeof(ch));
    for (int j = 0; j < m; ++j) ch.v[j] = v[j];
    return kk_launch_householder(b->ctx, b->col(c0), b->ld, m, &ch, beta);
}

// ---- Gram rows for the low-synchronisation MGS -------------------------------------------
// gram(i, j) = <b_i, b_j>, j < i, stored at b->gram[i*cap + j]; rows [0, gram_rows) valid.
static int block_inner_run(kk_ctx c, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
                           double* M, int ldm);
// device mirror of the host Gram rows (used by the on-device low-sync solve)
static int gram_device(kk_basis b) {
    if (b->gram.empty()) b->gram.assign((size_t)b->cap * b->cap, 0.0);
    if (!b->d_gram) {
        KK_HIP(hipMalloc(&b->d_gram, (size_t)b->cap * b->cap * sizeof(double)));
        KK_HIP(hipMemcpy(b->d_gram, b->gram.data(), (size_t)b->cap * b->cap * sizeof(double), hipMemcpyHostToDevice));
    }
    return KK_OK;
}
static int gram_upload_rows(kk_basis b, int lo, int hi) {
    if (!b->d_gram || hi <= lo) return KK_OK;
    // pageable source: the runtime stages it before returning, so the host vector may change afterwards
    KK_HIP(hipMemcpyAsync(b->d_gram + (size_t)lo * b->cap, b->gram.data() + (size_t)lo * b->cap,
                          (size_t)(hi - lo) * b->cap * sizeof(double), hipMemcpyHostToDevice, b->ctx->stream));
    return KK_OK;
}
static int gram_ensure_host(kk_basis b, int upto);
static int gram_ensure(kk_basis b, int upto /* exclusive */) {
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

// ---- one orthogonalisation pass; coefficient results land in pinned slot `slot` ------------
// CGS pass:  s = V'w ; w -= V s ; optional |w| (orthonormal.jl:378-384)
static int pass_cgs(kk_ctx c, const double* V, int64_t ld, int m, double* w, bool want_norm, int slot) {
    KK_TRY(kk_launch_project(c, V, ld, m, w, nullptr, nullptr, nullptr, WSP(c, WS_S), WSP(c, WS_G)));
    KK_TRY(ws_fetch_async(c, WS_S, m, slot));
    KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, c->ws + WS_S, -1.0, 1.0, -1, nullptr,
                               want_norm ? SCP(c, SC_NRM2) : nullptr));
    if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, slot));
    return KK_OK;
}
// strict MGS sweep (orthonormal.jl:414-423): `carry` = pending axpy (q, &s) left over from a
// previous sweep whose last subtraction is fused into this sweep's first dot.
static int pass_mgs_strict(kk_ctx c, const double* V, int64_t ld, int m, double* w, int64_t ws_s, bool want_norm,
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
static int lowsync_project_dev(kk_basis b, int m, const double* w, const double* pre_vec, const double* pre_a,
                               const double* a0_dev, int64_t ws_coef, int64_t ws_s, bool* rode) {
    kk_ctx c = b->ctx;
    const int newest = m - 1;
    if (b->gram_rows < newest) KK_TRY(gram_ensure(b, newest));  // only after the basis was transformed (restart)
    if (b->gram_rows < 1) b->gram_rows = 1;
    const bool ride = (b->gram_rows == newest && newest > 0);
    KK_TRY(gram_device(b));
    KK_TRY(kk_launch_project(c, b->col(0), b->ld, m, w, pre_vec, pre_a, ride ? b->col(newest) : nullptr, WSP(c, WS_S),
                             WSP(c, WS_G)));
    KK_TRY(kk_launch_lowsync_solve(c, WSP(c, WS_S), ride ? WSP(c, WS_G) : nullptr, b->d_gram, b->cap, m, newest, a0_dev,
                                   WSP(c, ws_coef), WSP(c, ws_s)));
    *rode = ride;
    return KK_OK;
}
static void lowsync_commit_row(kk_basis b, int m, const double* g_host) {
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
static int final_sync(kk_ctx c);
static int orth_run(kk_basis b, int c0, int m, double* w, kk_orth_t alg, double eta, double* x, double* nrm,
                    int* npasses, bool want_norm) {
    kk_ctx c = b->ctx;
    const double* V = b->col(c0);
    const int64_t ld = b->ld;
    int passes = 0;
    double nn = 0;
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
    const bool lowsync = c->mgs_mode == 1 && c0 == 0;
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
                KK_TRY(ws_fetch_async(c, WS_S, m, 0));
                KK_TRY(kk_launch_unproj_proj(c, V, ld, m, w, w, nullptr, WSP(c, WS_S), WSP(c, WS_G), nullptr));
                KK_TRY(ws_fetch_async(c, WS_G, m, 0));
                KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, WSP(c, WS_G), -1.0, 1.0, -1, nullptr,
                                           want_norm ? SCP(c, SC_NRM2) : nullptr));
                if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
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
                KK_TRY(ws_fetch_async(c, WS_Y, m, 0));
                if (rode) KK_TRY(ws_fetch_async(c, WS_G, m, 0));
                if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
                KK_TRY(final_sync(c));
                memcpy(x, pin(c, WS_Y, 0), m * sizeof(double));
                if (rode) lowsync_commit_row(b, m, pin(c, WS_G, 0));
            } else if (lowsync) {
                KK_TRY(pass_mgs_lowsync(b, c0, m, w, x, want_norm, 0));
                KK_TRY(final_sync(c));
            } else {
                KK_TRY(pass_mgs_strict(c, V, ld, m, w, WS_S, want_norm, 0, nullptr, nullptr, false));
                KK_TRY(final_sync(c));
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
                KK_TRY(ws_fetch_async(c, WS_Y, 2 * KK_MAX_M, 0));   // s1 (WS_Y) and s2 (WS_Z) are adjacent
                if (rode) KK_TRY(ws_fetch_async(c, WS_G, m, 0));
                if (want_norm) KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 2, 0));
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
                KK_TRY(pass_mgs_strict(c, V, ld, m, w, WS_S, false, 0, nullptr, nullptr, true));
                KK_TRY(pass_mgs_strict(c, V, ld, m, w, WS_G, want_norm, 0, V + (int64_t)(m - 1) * ld,
                                       c->ws + WS_S + m - 1, false));
                KK_TRY(final_sync(c));
                for (int j = 0; j < m; ++j) x[j] = pin(c, WS_S, 0)[j] + pin(c, WS_G, 0)[j];
            }
            nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            passes = 2;
        } break;
        case KK_MGSIR: {  // :440-452
            KK_TRY(kk_launch_nrm2(c, w, ld, SCP(c, SC_NRM2B)));
            KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2B, 2, 1));
            if (lowsync) KK_TRY(pass_mgs_lowsync(b, c0, m, w, x, true, 0));
            else KK_TRY(pass_mgs_strict(c, V, ld, m, w, WS_S, true, 0, nullptr, nullptr, false));
            KK_TRY(stream_sync(c));
            double nold = pin(c, WS_SCAL + SC_NRM2B, 1)[1];
            if (!lowsync) memcpy(x, pin(c, WS_S, 0), m * sizeof(double));
            nn = pin(c, WS_SCAL + SC_NRM2, 0)[1];
            passes = 1;
            while (KK_EPS < nn && nn < eta * nold) {
                nold = nn;
                if (lowsync) KK_TRY(pass_mgs_lowsync(b, c0, m, w, tmp.data(), true, 0));
                else KK_TRY(pass_mgs_strict(c, V, ld, m, w, WS_S, true, 0, nullptr, nullptr, false));
                KK_TRY(stream_sync(c));
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

extern "C" int kk_orthogonalize(kk_basis b, int c0, int m, kk_basis bw, int cw, kk_orth_t alg, double eta, double* x,
                                double* nrm, int* npasses) {
    CHECK_RANGE(b, c0, m); CHECK_COL(bw, cw); CHECK_SAME(b, bw);
    KK_CHECK(x || m == 0, KK_ERR_INVALID, "null x");
    KK_CHECK(!(bw == b && cw >= c0 && cw < c0 + m), KK_ERR_INVALID, "kk_orthogonalize: w aliases a basis column");
    gram_touch(bw, cw);
    return orth_run(b, c0, m, bw->col(cw), alg, eta, x, nrm, npasses, nrm != nullptr);
}

extern "C" int kk_orthonormalize(kk_basis b, int c0, int m, kk_basis bw, int cw, kk_orth_t alg, double eta, double* x,
                                 double* nrm, int* npasses) {
    CHECK_RANGE(b, c0, m); CHECK_COL(bw, cw); CHECK_SAME(b, bw);
    KK_CHECK(x || m == 0, KK_ERR_INVALID, "null x");
    KK_CHECK(!(bw == b && cw >= c0 && cw < c0 + m), KK_ERR_INVALID, "kk_orthonormalize: w aliases a basis column");
    gram_touch(bw, cw);
    double nn = 0;
    KK_TRY(orth_run(b, c0, m, bw->col(cw), alg, eta, x, &nn, npasses, true));
    if (nrm) *nrm = nn;
    return kk_launch_scal(b->ctx, bw->col(cw), bw->ld, 1.0 / nn, nullptr);  // scale!!(v, inv(beta))  :525
}

// _orthogonalize!!(v, q, alg) (orthonormal.jl:455-489)
static int orth_vec_run(kk_ctx c, const double* q, double* w, int64_t ld, kk_orth_t alg, double eta, double* s_out,
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

extern "C" int kk_orthogonalize_vec(kk_basis bq, int cq, kk_basis bw, int cw, kk_orth_t alg, double eta, double* s,
                                    double* nrm) {
    CHECK_COL(bq, cq); CHECK_COL(bw, cw); CHECK_SAME(bq, bw);
    KK_CHECK(!(bq == bw && cq == cw), KK_ERR_INVALID, "kk_orthogonalize_vec: q and w must differ");
    gram_touch(bw, cw);
    return orth_vec_run(bq->ctx, bq->col(cq), bw->col(cw), bw->ld, alg, eta, s, nrm, nrm != nullptr);
}

// ------------------------------------------------------------------------------------------
// L3 fused expand! steps
// ------------------------------------------------------------------------------------------
static int check_square_op(kk_op op, kk_basis b) {
    KK_CHECK(op && b, KK_ERR_INVALID, "null arg");
    KK_CHECK(op->ctx == b->ctx, KK_ERR_INVALID, "operator and basis belong to different contexts");
    const int64_t in = op->A.n_ghost > 0 ? op->A.n_local : op->ncols;
    KK_CHECK(op->nrows == b->n && in == b->n, KK_ERR_DIM, "operator is %lldx%lld but vectors have %lld rows",
             (long long)op->nrows, (long long)op->ncols, (long long)b->n);
    return KK_OK;
}

// initialize (factorizations/lanczos.jl:180-222 == arnoldi.jl:135-175): col c0 = x0 -> v ; col c0+1 = r
static int krylov_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta, double* alpha, double* beta) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(c0 >= 0 && c0 + 2 <= b->cap, KK_ERR_INVALID, "initialize: need columns %d..%d", c0, c0 + 1);
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

extern "C" int kk_lanczos_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta, double* alpha,
                                     double* beta) {
    return krylov_initialize(op, b, c0, orth, eta, alpha, beta);
}
extern "C" int kk_arnoldi_initialize(kk_op op, kk_basis b, int c0, kk_orth_t orth, double eta, double* alpha,
                                     double* beta) {
    return krylov_initialize(op, b, c0, orth, eta, alpha, beta);
}

// Speculative first half of the NEXT expand!: w' = A (r/beta) - beta v  with the scale applied on
// the fly from the device-resident beta, so the GPU keeps working while the host reads back
// (alpha, beta), returns to the caller and re-enters.  r itself is NOT modified; the next expand
// call normalises it in place (after this read) and skips its SpMV if (op, c0, k, beta) match.
// Bit-identical to the non-speculative order: r*(1/beta) is formed with the same operands.
static int speculate_next(kk_op op, kk_basis b, int c0, int k_next, int dot_mode, bool with_prev, double beta_host) {
    kk_ctx c = b->ctx;
    b->spec_valid = false;
    if (!c->speculate || c0 + k_next + 2 > b->cap || k_next + 1 > KK_MAX_M) return KK_OK;
    kk_spmv_fuse f;
    f.xscale_dev = SCP(c, SC_INVNRM);
    if (with_prev) { f.vprev = b->col(c0 + k_next - 1); f.bprev_dev = SCP(c, SC_NRM); }
    f.dot_mode = dot_mode;
    f.dot_out = SCP(c, SC_SPECA);
    KK_TRY(kk_launch_spmv(c, op->A, b->col(c0 + k_next), b->col(c0 + k_next + 1), b->ld, f));
    b->spec_valid = true; b->spec_op = op; b->spec_c0 = c0; b->spec_k = k_next; b->spec_dot_mode = dot_mode;
    b->spec_beta = beta_host;
    c->spec_owner = b;
    return KK_OK;
}
// true if the previous expand on this basis already enqueued exactly this step's SpMV; moves the
// speculative alpha into the regular slot
static int spec_take(kk_op op, kk_basis b, int c0, int k, int dot_mode, double beta_old, bool* hit) {
    kk_ctx c = b->ctx;
    *hit = b->spec_valid && c->spec_owner == b && b->spec_op == op && b->spec_c0 == c0 && b->spec_k == k &&
           b->spec_dot_mode == dot_mode && b->spec_beta == beta_old;
    if (*hit && dot_mode)
        KK_HIP(hipMemcpyAsync(SCP(c, SC_ALPHA0), SCP(c, SC_SPECA), sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return KK_OK;
}
// the last synchronisation of an expand: a pending speculation request (Arnoldi) is enqueued first
// The host waits only for the read-backs queued so far (event), NOT for the speculative SpMV that
// is enqueued behind them -- that one keeps the GPU busy during the host round trip.
static int fetch_mark(kk_ctx c) {
    KK_HIP(hipEventRecord(c->ev_fetch, c->stream));
    return KK_OK;
}
static int fetch_wait(kk_ctx c) {
    KK_HIP(hipEventSynchronize(c->ev_fetch));
    return KK_OK;
}
static int final_sync(kk_ctx c) {
    if (c->spec_req.active) {
        c->spec_req.active = false;
        KK_TRY(fetch_mark(c));
        KK_TRY(speculate_next(c->spec_req.op, c->spec_req.b, c->spec_req.c0, c->spec_req.k_next, 0, false, 0.0));
        return fetch_wait(c);
    }
    return stream_sync(c);
}

extern "C" int kk_lanczos_expand(kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, double eta, double beta_old,
                                 double* alpha, double* beta, int* npasses) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(k >= 1 && c0 >= 0 && c0 + k + 2 <= b->cap && k + 1 <= KK_MAX_M, KK_ERR_INVALID,
             "kk_lanczos_expand: need k >= 1 and columns %d..%d within capacity %d", c0, c0 + k + 1, b->cap);
    KK_CHECK(alpha && beta, KK_ERR_INVALID, "null output");
    KK_CHECK(beta_old != 0.0, KK_ERR_ZERO_NORM, "kk_lanczos_expand: residual norm is zero");
    kk_ctx c = b->ctx;
    const int64_t ld = b->ld;
    const int m = k + 1;                  // basis size after the push
    double* V = b->col(c0);
    double* v = b->col(c0 + k);           // holds r on entry
    const double* vprev = b->col(c0 + k - 1);
    double* w = b->col(c0 + k + 1);
    const bool cgs_order = (orth == KK_CGS || orth == KK_CGS2 || orth == KK_CGSIR);
    bool hit = false;
    KK_TRY(spec_take(op, b, c0, k, cgs_order ? 1 : 2, beta_old, &hit));
    gram_touch(b, c0 + k);
    int passes = 0;
    // V = push!(V, scale!!(r, 1/beta_old))   lanczos.jl:257
    KK_TRY(kk_launch_scal(c, v, ld, 1.0 / beta_old, nullptr));
    if (!hit) {
        // w = A v - beta_old v_prev with the fused alpha dot   lanczos.jl:297-299 / 306-308
        kk_spmv_fuse f;
        f.vprev = vprev; f.bprev = beta_old;
        f.dot_mode = cgs_order ? 1 : 2;
        f.dot_out = SCP(c, SC_ALPHA0);
        KK_TRY(kk_launch_spmv(c, op->A, v, w, ld, f));
    }  // else: the previous expand already enqueued exactly this SpMV (speculate_next)
    const double* a0_dev = c->ws + WS_SCAL + SC_ALPHA0;
    double a = 0, bt = 0;
    const bool lowsync = c->mgs_mode == 1;
    if (orth == KK_CGS || orth == KK_MGS || orth == KK_CGSIR || orth == KK_MGSIR) {
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
        }
    } else if (orth == KK_CGS2 || (orth == KK_MGS2 && lowsync && c0 == 0)) {
        // one projection pass with "w -= alpha0 v" folded in (read V twice in total):
        //   s = V'(w - alpha0 v) ; w <- w - V (s + alpha0 e_m) ; beta = |w|     lanczos.jl:318-322 / 329-336
        if (orth == KK_CGS2) {
            KK_TRY(kk_launch_project(c, V, ld, m, w, v, a0_dev, nullptr, WSP(c, WS_S), WSP(c, WS_G)));
            KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, c->ws + WS_S, -1.0, 1.0, m - 1, a0_dev,
                                       SCP(c, SC_NRM2)));
            KK_TRY(ws_fetch_async(c, WS_S, m, 0));
            KK_TRY(ws_fetch_async(c, WS_SCAL, 4, 0));
            KK_TRY(fetch_mark(c));
            KK_TRY(speculate_next(op, b, c0, k + 1, 1, true, 0.0));
            KK_TRY(fetch_wait(c));
            a = pin(c, WS_SCAL + SC_ALPHA0)[0] + pin(c, WS_S)[m - 1];
        } else {
            // low-sync MGS2: project (Gram row riding along) -> triangular solve ON THE DEVICE (alpha0 folded
            // into the last coefficient) -> update; one host synchronisation, as for CGS2
            bool rode = false;
            KK_TRY(lowsync_project_dev(b, m, w, v, a0_dev, a0_dev, WS_X, WS_Y, &rode));
            KK_TRY(kk_launch_unproject(c, V, ld, m, w, w, nullptr, WSP(c, WS_X), -1.0, 1.0, -1, nullptr, SCP(c, SC_NRM2)));
            KK_TRY(ws_fetch_async(c, WS_SCAL, 4, 0));
            KK_TRY(ws_fetch_async(c, WS_Y + m - 1, 1, 0));
            if (rode) KK_TRY(ws_fetch_async(c, WS_G, m, 0));
            KK_TRY(fetch_mark(c));
            KK_TRY(speculate_next(op, b, c0, k + 1, 2, true, 0.0));
            KK_TRY(fetch_wait(c));
            a = pin(c, WS_SCAL + SC_ALPHA0)[0] + pin(c, WS_Y)[m - 1];
            if (rode) lowsync_commit_row(b, m, pin(c, WS_G, 0));
        }
        bt = pin(c, WS_SCAL + SC_NRM)[0];
        passes = 1;
    } else if (orth == KK_MGS2) {
        // strict: w -= alpha0 v fused with the first dot of the sweep   lanczos.jl:329-334
        KK_TRY(pass_mgs_strict(c, V, ld, m, w, WS_S, true, 0, v, a0_dev, false));
        KK_TRY(ws_fetch_async(c, WS_SCAL, 1, 0));
        KK_TRY(stream_sync(c));
        a = pin(c, WS_SCAL + SC_ALPHA0)[0] + pin(c, WS_S)[m - 1];
        bt = pin(c, WS_SCAL + SC_NRM2)[1];
        passes = 1;
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

extern "C" int kk_arnoldi_expand(kk_op op, kk_basis b, int c0, int k, kk_orth_t orth, double eta, double beta_old,
                                 double* h, double* beta, int* npasses) {
    KK_TRY(check_square_op(op, b));
    KK_CHECK(k >= 1 && c0 >= 0 && c0 + k + 2 <= b->cap && k + 1 <= KK_MAX_M, KK_ERR_INVALID,
             "kk_arnoldi_expand: need k >= 1 and columns %d..%d within capacity %d", c0, c0 + k + 1, b->cap);
    KK_CHECK(h && beta, KK_ERR_INVALID, "null output");
    KK_CHECK(beta_old != 0.0, KK_ERR_ZERO_NORM, "kk_arnoldi_expand: residual norm is zero");
    kk_ctx c = b->ctx;
    const int m = k + 1;
    double* v = b->col(c0 + k);
    double* w = b->col(c0 + k + 1);
    bool hit = false;
    KK_TRY(spec_take(op, b, c0, k, 0, beta_old, &hit));
    gram_touch(b, c0 + k);
    KK_TRY(kk_launch_scal(c, v, b->ld, 1.0 / beta_old, nullptr));  // push!(V, scale(r, 1/beta))   arnoldi.jl:209
    if (!hit) {
        kk_spmv_fuse f;
        KK_TRY(kk_launch_spmv(c, op->A, v, w, b->ld, f));           // w = apply(operator, last(V))  :242
    }
    // ask orth_run to enqueue the NEXT step's SpMV right before its final host sync (non-IR variants)
    c->spec_req.active = (orth != KK_CGSIR && orth != KK_MGSIR);
    c->spec_req.op = op; c->spec_req.b = b; c->spec_req.c0 = c0; c->spec_req.k_next = k + 1;
    int st = orth_run(b, c0, m, w, orth, eta, h, beta, npasses, true);  // orthogonalize!! + norm      :243-244
    c->spec_req.active = false;
    if (st == KK_OK && b->spec_valid) b->spec_beta = *beta;
    return st;
}

// ---- GKL ----------------------------------------------------------------------------------
static int check_gkl(kk_op op, kk_basis bu, kk_basis bv) {
    KK_CHECK(op && bu && bv, KK_ERR_INVALID, "null arg");
    KK_CHECK(op->ctx == bu->ctx && op->ctx == bv->ctx, KK_ERR_INVALID, "objects belong to different contexts");
    KK_CHECK(op->nrows == bu->n && op->ncols == bv->n, KK_ERR_DIM,
             "GKL: operator is %lldx%lld, U vectors have %lld rows, V vectors %lld", (long long)op->nrows,
             (long long)op->ncols, (long long)bu->n, (long long)bv->n);
    KK_CHECK(op->A.n_ghost == 0, KK_ERR_UNSUPPORTED, "GKL on ghosted (row-sharded) operators goes through the split-phase API");
    return KK_OK;
}

extern "C" int kk_gkl_initialize(kk_op op, kk_basis bu, kk_basis bv, double* alpha, double* beta) {
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
    kk_spmv_fuse f1;
    f1.nrm_out = SCP(c, SC_NRM2);
    KK_TRY(kk_launch_spmv(c, *At, u0, v0, bv->ld, f1));
    kk_spmv_fuse f2;
    KK_TRY(kk_launch_spmv(c, op->A, v0, r, bu->ld, f2));
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

extern "C" int kk_gkl_expand(kk_op op, kk_basis bu, kk_basis bv, int k, kk_orth_t orth, double eta, double beta_old,
                             double* alpha, double* beta, int* npasses_v, int* npasses_u) {
    KK_TRY(check_gkl(op, bu, bv));
    KK_CHECK(k >= 1 && k + 2 <= bu->cap && k + 1 <= bv->cap && k + 1 <= KK_MAX_M, KK_ERR_INVALID,
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
    kk_spmv_fuse f1;
    f1.vprev = vlast; f1.bprev = beta_old;
    const bool v_sweep = (orth == KK_MGS2 || orth == KK_CGSIR || orth == KK_MGSIR);
    f1.nrm_out = SCP(c, SC_NRM2);
    KK_TRY(kk_launch_spmv(c, *At, u, v, bv->ld, f1));
    if (orth == KK_MGS2) {  // gkl.jl:330-336
        double nn = 0;
        KK_TRY(orth_run(bv, 0, k, v, KK_MGS, eta, tmp.data(), &nn, nullptr, true));
        a = nn;
        pv = 1;
        // publish alpha / 1/alpha on the device for the next kernels
        double hv[3] = {a * a, a, 1.0 / a};
        KK_HIP(hipMemcpyAsync(c->ws + WS_SCAL + SC_NRM2, hv, sizeof(hv), hipMemcpyHostToDevice, c->stream));
        KK_TRY(stream_sync(c));
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
    kk_spmv_fuse f2;
    f2.vprev = u; f2.bprev_dev = alpha_dev;
    f2.nrm_out = SCP(c, SC_NRM2B);
    KK_TRY(kk_launch_spmv(c, op->A, v, r, bu->ld, f2));
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

// ------------------------------------------------------------------------------------------
// BlockLanczos (src/factorizations/blocklanczos.jl)
// ------------------------------------------------------------------------------------------
#define CHECK_BLOCK(b, c0, p) KK_CHECK((b) && (c0) >= 0 && (p) >= 0 && (c0) + (p) <= (b)->cap, KK_ERR_INVALID, "%s: block [%d,%d) outside capacity %d", __func__, (c0), (c0) + (p), (b) ? (b)->cap : 0)

// M_host (p x q, ldm) = X' Y.  Panel mode: MFMA gram tiles into device scratch, ONE D2H + sync.
static int block_inner_run(kk_ctx c, const double* X, int64_t ldx, int p, const double* Y, int64_t ldy, int q, int64_t ld,
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

extern "C" int kk_block_inner(kk_basis bx, int cx, int p, kk_basis by, int cy, int q, double* M, int ldm) {
    CHECK_BLOCK(bx, cx, p); CHECK_BLOCK(by, cy, q); CHECK_SAME(bx, by);
    KK_CHECK(M && ldm >= p, KK_ERR_INVALID, "kk_block_inner: bad output");
    return block_inner_run(bx->ctx, bx->col(cx), bx->ld, p, by->col(cy), by->ld, q, bx->ld, M, ldm);
}

extern "C" int kk_block_apply(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, int nb) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    CHECK_BLOCK(bx, cx, nb); CHECK_BLOCK(by, cy, nb);
    KK_TRY(check_apply(op, 0, bx, by));
    KK_CHECK(!(bx == by && cx < cy + nb && cy < cx + nb), KK_ERR_INVALID, "kk_block_apply: blocks overlap");
    gram_touch(by, cy);
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
static int block_update_run(kk_ctx c, const double* V, int64_t ld, int m, double* W, int64_t ldw, int q, const double* S,
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

extern "C" int kk_block_update(kk_basis bw, int cw, int q, kk_basis b, int c0, int m, const double* S, int lds,
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

extern "C" int kk_block_reorthogonalize(kk_basis b, int c0, int m, int cw, int q) {
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
                        int* is_drift) {
    kk_ctx c = b->ctx;
    if (c->block_mode == 1 && c_out != c_in && p <= 64 && p >= 2) {
        const int64_t ld = b->ld;
        std::vector<double> G((size_t)p * p), R1, R2, Ri;
        KK_TRY(block_inner_run(c, b->col(c_in), ld, p, b->col(c_in), ld, p, ld, G.data(), p));
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

extern "C" int kk_block_qr(kk_basis b, int c_in, int p, int c_out, double tol, double* R, int ldr, int* good_idx,
                           int* ngood, int* is_drift) {
    CHECK_BLOCK(b, c_in, p); CHECK_BLOCK(b, c_out, p);
    KK_CHECK(p >= 1 && p <= KK_MAX_M, KK_ERR_INVALID, "kk_block_qr: p=%d", p);
    KK_CHECK(R && good_idx && ngood && ldr >= p, KK_ERR_INVALID, "kk_block_qr: bad output arguments");
    KK_CHECK(c_out == c_in || c_out + p <= c_in || c_in + p <= c_out, KK_ERR_INVALID, "kk_block_qr: partial overlap");
    gram_touch(b, std::min(c_in, c_out));
    return block_qr_run(b, c_in, p, c_out, tol, R, ldr, good_idx, ngood, is_drift);
}

extern "C" int kk_blocklanczos_initialize(kk_op op, kk_basis b, int c_x0, int bs0, int c_r, double qr_tol, int* bs,
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
    for (int j = 0; j < bs0; ++j) {
        KK_TRY(kk_launch_nrm2(c, b->col(c_x0 + j), b->ld, SCP(c, SC_NRM2)));
        KK_TRY(ws_fetch_async(c, WS_SCAL + SC_NRM2, 1, 0));
        KK_TRY(stream_sync(c));
        n2 += pin(c, WS_SCAL + SC_NRM2)[0];
    }
    if (n2 == 0.0) {
        kk_set_error("initial vector should not have norm zero");
        return KK_ERR_ZERO_NORM;
    }
    int ng = 0, drift = 0;
    KK_TRY(block_qr_run(b, c_x0, bs0, 0, qr_tol, R.data(), bs0, good.data(), &ng, &drift));  // X1 = block_qr!(X0)[good]  :175-177
    KK_CHECK(ng >= 1, KK_ERR_ZERO_NORM, "kk_blocklanczos_initialize: start block has numerical rank 0");
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

extern "C" int kk_blocklanczos_expand(kk_op op, kk_basis b, int k, int bs_r, int c_r, int c_rnext, double qr_tol,
                                      int* bs_next, double* B, int ldb, double* M, int ldm, double* norm_R, int* is_drift) {
    KK_TRY(check_square_op(op, b));
    CHECK_BLOCK(b, 0, k + bs_r); CHECK_BLOCK(b, c_r, bs_r); CHECK_BLOCK(b, c_rnext, bs_r);
    KK_CHECK(k >= bs_r && bs_r >= 1 && bs_next && B && M && norm_R && ldb >= bs_r && ldm >= bs_r, KK_ERR_INVALID,
             "kk_blocklanczos_expand: bad arguments");
    KK_CHECK(c_r >= k + bs_r && c_rnext >= k + bs_r && (c_rnext + bs_r <= c_r || c_r + bs_r <= c_rnext), KK_ERR_INVALID,
             "kk_blocklanczos_expand: residual blocks must lie beyond column k+bs_r and not overlap");
    kk_ctx c = b->ctx;
    gram_touch(b, k);
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

// ------------------------------------------------------------------------------------------
// split-phase API (row-sharded multi-GPU runs): local partials into caller-owned device buffers
// ------------------------------------------------------------------------------------------
extern "C" int kk_apply_fused_dev(kk_op op, kk_basis b, int col_v, int col_prev, int col_w, double beta_old,
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
extern "C" int kk_apply_fused_dev2(kk_op op, kk_basis b, int col_v, int col_prev, int col_w, const void* dev_xscale,
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
extern "C" int kk_unproject_devcoef(kk_basis by, int cy, kk_basis b, int c0, int m, const void* dev_coef, double alpha,
                                    double beta, void* dev_nrm) {
    CHECK_RANGE(b, c0, m); CHECK_COL(by, cy); CHECK_SAME(b, by);
    KK_CHECK(dev_coef || m == 0, KK_ERR_INVALID, "null coef");
    KK_CHECK(!(by == b && cy >= c0 && cy < c0 + m), KK_ERR_INVALID, "kk_unproject_devcoef: y aliases a basis column");
    gram_touch(by, cy);
    return kk_launch_unproject(b->ctx, b->col(c0), b->ld, m, by->col(cy), by->col(cy), nullptr, (const double*)dev_coef, alpha,
                               beta, -1, nullptr, (double*)dev_nrm);
}
extern "C" int kk_project_dev(kk_basis b, int c0, int m, kk_basis bx, int cx, int col_rhs2, void* dev_out) {
    CHECK_RANGE(b, c0, m); CHECK_COL(bx, cx); CHECK_SAME(b, bx);
    KK_CHECK(dev_out || m == 0, KK_ERR_INVALID, "null output");
    KK_CHECK(col_rhs2 < bx->cap, KK_ERR_INVALID, "kk_project_dev: rhs2 column out of range");
    if (m == 0) return KK_OK;
    double* o = (double*)dev_out;
    return kk_launch_project(b->ctx, b->col(c0), b->ld, m, bx->col(cx), nullptr, nullptr,
                             col_rhs2 >= 0 ? bx->col(col_rhs2) : nullptr, o, o + m);
}
extern "C" int kk_unproject_dev(kk_basis by, int cy, kk_basis b, int c0, int m, const double* coef, double alpha,
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
extern "C" int kk_dot_dev(kk_basis bx, int cx, kk_basis by, int cy, void* dev_out) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    KK_CHECK(dev_out, KK_ERR_INVALID, "null output");
    return kk_launch_dot(bx->ctx, bx->col(cx), by->col(cy), bx->ld, (double*)dev_out);
}
extern "C" int kk_nrm2_dev(kk_basis bx, int cx, void* dev_out3) {
    CHECK_COL(bx, cx);
    KK_CHECK(dev_out3, KK_ERR_INVALID, "null output");
    return kk_launch_nrm2(bx->ctx, bx->col(cx), bx->ld, (double*)dev_out3);
}
extern "C" int kk_axpy_dev(kk_basis by, int cy, kk_basis bx, int cx, const void* dev_a, double sign) {
    CHECK_COL(bx, cx); CHECK_COL(by, cy); CHECK_SAME(bx, by);
    KK_CHECK(dev_a, KK_ERR_INVALID, "null scalar");
    gram_touch(by, cy);
    return kk_launch_axpby(by->ctx, by->col(cy), bx->col(cx), by->ld, 0.0, 1.0, (const double*)dev_a, sign, 1);
}
extern "C" int kk_scal_rsqrt_dev(kk_basis bx, int cx, const void* dev_nrm2) {
    CHECK_COL(bx, cx);
    KK_CHECK(dev_nrm2, KK_ERR_INVALID, "null scalar");
    gram_touch(bx, cx);
    return kk_launch_scal(bx->ctx, bx->col(cx), bx->ld, 0.0, (const double*)dev_nrm2, 1);
}
