// libkrylov_hip.so, C ABI part 8: the multi-GPU layer.  One process (context) per GPU, every N-vector row-sharded;
// the library owns an RCCL communicator (kk_comm_init) and issues the collectives of the path itself, on the context
// stream, between its own kernels:
//   * ncclAllReduce(sum, f64) of the <= 2m+1 doubles of every inner-product-type result (all finalize sites),
//   * grouped ncclSend / ncclRecv of the ghost entries before a sparse apply (kk_csr_create_sharded),
//   * ncclAllGather / ncclReduceScatter of the short vectors of a rectangular map (kk_csr_create_sharded_rect, GKL).
// The reference has no communication layer at all; this is north_star's "basis row-sharded
// across the 8 GPUs of one node with RCCL allreduce over xGMI".  librccl is dlopen'ed at the first kk_comm_* call so
// single-GPU processes never map it (573 MB) and a torch-bundled copy loaded earlier in the process is reused.
#include "kk_host.h"
#include <dlfcn.h>
#include <mutex>
#include <unistd.h>
#include <rccl/rccl.h>

namespace {
struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};
rccl_api g_rccl;
std::mutex g_rccl_mutex;

int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.handle) return KK_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    const char* env = getenv("KK_RCCL_LIB");
    if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);  // already mapped?
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        kk_set_error("kk_comm: cannot load librccl (%s); set KK_RCCL_LIB", dlerror());
        return KK_ERR_UNSUPPORTED;
    }
#define KK_SYM(field, name)                                                      \
    do {                                                                         \
        *(void**)(&g_rccl.field) = dlsym(h, name);                               \
        if (!g_rccl.field) {                                                     \
            kk_set_error("kk_comm: librccl lacks the symbol %s", name);          \
            dlclose(h);                                                          \
            return KK_ERR_UNSUPPORTED;                                           \
        }                                                                        \
    } while (0)
    KK_SYM(GetUniqueId, "ncclGetUniqueId");
    KK_SYM(CommInitRank, "ncclCommInitRank");
    KK_SYM(CommDestroy, "ncclCommDestroy");
    KK_SYM(AllReduce, "ncclAllReduce");
    KK_SYM(AllGather, "ncclAllGather");
    KK_SYM(ReduceScatter, "ncclReduceScatter");
    KK_SYM(Send, "ncclSend");
    KK_SYM(Recv, "ncclRecv");
    KK_SYM(GroupStart, "ncclGroupStart");
    KK_SYM(GroupEnd, "ncclGroupEnd");
    KK_SYM(GetErrorString, "ncclGetErrorString");
    KK_SYM(GetVersion, "ncclGetVersion");
#undef KK_SYM
    g_rccl.handle = h;
    return KK_OK;
}
int rccl_fail(ncclResult_t r, const char* what, int line) {
    kk_set_error("RCCL error %d (%s) in %s at kk_comm.hip:%d", (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", what, line);
    return KK_ERR_HIP;
}
}  // namespace
#define KK_NCCL(call)                                                  \
    do {                                                               \
        ncclResult_t _r = (call);                                      \
        if (_r != ncclSuccess) return rccl_fail(_r, #call, __LINE__);  \
    } while (0)

static_assert(KK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "kk_comm id size must match ncclUniqueId");

// ------------------------------------------------------------------------------------------
// communicator
// ------------------------------------------------------------------------------------------
KK_API int kk_comm_get_unique_id(void* id128) {
    KK_CHECK(id128, KK_ERR_INVALID, "kk_comm_get_unique_id: null buffer");
    KK_TRY(rccl_load());
    ncclUniqueId id;
    KK_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return KK_OK;
}

// ------------------------------------------------------------------------------------------
// cross-rank in-kernel reduction (kk_xsync.h): every rank allocates a small sync area in fine-grained device memory,
// the ranks exchange its IPC handle through the communicator and map each other's area.  The feature is switched on only
// when EVERY rank succeeded (agreed by an all-reduce of the local status): the sequence of collectives issued here is the
// same on every rank whatever happens locally.
// ------------------------------------------------------------------------------------------
namespace {
struct xs_record {            // what a rank tells the others (16 x 8 bytes)
    int64_t ok, pid, dev, ptr;
    int64_t bus;              // hash of the PCI bus id of the rank's device: ranks that SHARE a GPU find each other by it
    int64_t cus, dev_cus, xcds;   // CUs the rank's context counts on ("num_cus"), CUs and XCDs of its device
    char handle[72];          // hipIpcMemHandle_t (64 bytes) + padding to a multiple of 8
};
static_assert(sizeof(hipIpcMemHandle_t) <= 72 && sizeof(xs_record) % 8 == 0, "xs_record layout");
}
static void xs_release(kk_ctx c) {
    kk_comm_s* k = c->comm;
    if (!k) return;
    for (int r = 0; r < KK_XS_MAX_RANKS; ++r) {
        if (k->xs_peer[r] && k->xs_peer[r] != (void*)k->xs_mine && k->xs_opened[r]) (void)hipIpcCloseMemHandle(k->xs_peer[r]);
        k->xs_peer[r] = nullptr; k->xs_opened[r] = false;
    }
    if (k->xs_table) (void)hipFree(k->xs_table);
    if (k->xs_mine) (void)hipFree(k->xs_mine);
    k->xs_table = nullptr; k->xs_mine = nullptr; k->xs_active = false;
    if (k->cus_before > 0) { c->num_cus = k->cus_before; k->cus_before = 0; }   // (the cut belongs to the cross-rank launches: single-rank work gets the chip back)
}
static int xs_setup(kk_ctx c) {
    kk_comm_s* k = c->comm;
    const char* env = getenv("KK_XSYNC");
    int local_ok = (k->world <= KK_XS_MAX_RANKS && !(env && atoi(env) == 0)) ? 1 : 0;
    xs_record mine;
    memset(&mine, 0, sizeof(mine));
    if (local_ok) {
        if (hipExtMallocWithFlags((void**)&k->xs_mine, KK_XS_BYTES, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); k->xs_mine = nullptr; local_ok = 0; }
    }
    if (local_ok && hipMemset(k->xs_mine, 0, KK_XS_BYTES) != hipSuccess) local_ok = 0;
    if (local_ok && k->world > 1) {
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, k->xs_mine) != hipSuccess) { (void)hipGetLastError(); local_ok = 0; }
        else memcpy(mine.handle, &h, sizeof(h));
    }
    mine.ok = local_ok; mine.pid = (int64_t)getpid(); mine.dev = c->device; mine.ptr = (int64_t)(uintptr_t)k->xs_mine;
    mine.cus = c->num_cus; mine.dev_cus = c->dev_cus; mine.xcds = c->dev_xcds;
    {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), c->device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof(bus), "dev%d-pid%d", c->device, (int)getpid()); }
        uint64_t h = 1469598103934665603ull;
        for (const char* q = bus; *q; ++q) h = (h ^ (uint64_t)(unsigned char)*q) * 1099511628211ull;
        mine.bus = (int64_t)(h & 0x7fffffffffffffffull);
    }
    // all-gather of the records (a world-1 communicator copies): staged in the block scratch, which nothing uses right now
    const int64_t nw = sizeof(xs_record) / 8;
    int64_t* d_send = (int64_t*)c->blk;
    int64_t* d_recv = d_send + 64;
    KK_HIP(hipMemcpyAsync(d_send, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
    KK_HIP(hipStreamSynchronize(c->stream));
    KK_TRY(kk_comm_allgather_i64(c, d_send, d_recv, nw));
    std::vector<xs_record> all((size_t)k->world);
    KK_HIP(hipMemcpyAsync(all.data(), d_recv, sizeof(xs_record) * (size_t)k->world, hipMemcpyDeviceToHost, c->stream));
    KK_HIP(hipStreamSynchronize(c->stream));
    int status = KK_OK;
    // Ranks that share ONE GPU (tests; a node with fewer GPUs than ranks): the persistent kernels launch one block per CU and need
    // the blocks of ALL ranks resident at once.  A launch deals its blocks to the XCDs round-robin, so W kernels of n blocks need
    // W * ceil(n / XCDs) CUs per XCD (profiles/r05_xsync_world3_residency.txt): unless the caller has set "num_cus" itself, every rank
    // takes its share of each XCD, less two CUs per XCD for the streaming kernels of the others.  And ALL ranks of the communicator
    // launch the SAME number of blocks, the minimum over the ranks: the route of a sweep and the panel width of k_mgs_panel follow
    // from (vector length, CUs), and every rank must arrive at the same ones -- the vector length is agreed per slab
    // (route_agree, kk_host.h).  Every rank derives the same numbers from the same records; they are APPLIED further down, once the
    // feature is known to be on (and taken back by xs_release).
    int cus_agreed = c->num_cus;
    {
        int n_share = 0;
        for (int r = 0; r < k->world; ++r) n_share += (all[r].bus == mine.bus);
        k->xs_share = n_share;
        cus_agreed = 1 << 30;
        for (int r = 0; r < k->world; ++r) {
            int share_r = 0;
            for (int q = 0; q < k->world; ++q) share_r += (all[q].bus == all[r].bus);
            int cus_r = (int)all[r].cus;
            const int xcds_r = all[r].xcds > 0 ? (int)all[r].xcds : 8;
            if (share_r > 1 && all[r].cus == all[r].dev_cus) cus_r = xcds_r * std::max(1, (int)(all[r].dev_cus / xcds_r) / share_r - 2);
            cus_agreed = std::min(cus_agreed, cus_r);
        }
        if (cus_agreed < 1 || cus_agreed > c->dev_cus) cus_agreed = c->num_cus;
    }
    for (int r = 0; r < k->world && local_ok; ++r) if (!all[r].ok) local_ok = 0;
    for (int r = 0; r < k->world && local_ok; ++r) {
        if (r == k->rank) { k->xs_peer[r] = k->xs_mine; continue; }
        if (all[r].pid == mine.pid) {   // another context of this process (one thread per GPU): same address space, no IPC mapping
            if ((int)all[r].dev != c->device) {
                hipError_t e = hipDeviceEnablePeerAccess((int)all[r].dev, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) local_ok = 0;
                (void)hipGetLastError();
            }
            k->xs_peer[r] = (void*)(uintptr_t)all[r].ptr;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, all[r].handle, sizeof(h));
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); local_ok = 0; }
        else { k->xs_peer[r] = p; k->xs_opened[r] = true; }
    }
    if (local_ok) {
        unsigned long long tab[KK_XS_MAX_RANKS] = {};
        for (int r = 0; r < k->world; ++r) tab[r] = (unsigned long long)(uintptr_t)k->xs_peer[r];
        if (hipMalloc((void**)&k->xs_table, sizeof(tab)) != hipSuccess || hipMemcpy(k->xs_table, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) local_ok = 0;
    }
    int worst = KK_OK;
    status = kk_comm_agree_status(c, local_ok ? KK_OK : KK_ERR_UNSUPPORTED, &worst);
    if (status != KK_OK || worst != KK_OK) { xs_release(c); return status; }   // (not an error of kk_comm_init: the RCCL route serves)
    // hand-shake through the mapped areas (every rank is here: the all-reduce above has just completed on all of them)
    kk_xs_dev a;
    a.table = k->xs_table; a.mine = k->xs_mine; a.tag0 = 1u; a.launch = 1u; a.rank = k->rank; a.world = k->world;
    int* d_out = (int*)(d_send + 512);
    KK_HIP(hipMemsetAsync(d_out, 0, sizeof(int), c->stream));
    KK_TRY(kk_launch_xs_selftest(c, a, d_out));
    int h_out = 0;
    KK_HIP(hipMemcpyAsync(&h_out, d_out, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    KK_HIP(hipStreamSynchronize(c->stream));
    k->xs_red = 4; k->xs_launch = 1;
    status = kk_comm_agree_status(c, h_out == 1 ? KK_OK : KK_ERR_UNSUPPORTED, &worst);
    if (status != KK_OK || worst != KK_OK) { xs_release(c); return status; }
    // What the two ways of summing over the ranks COST on this machine, measured here and now: XS_NRED
    // in-kernel reductions back to back (one round trip through every peer's area each) and XS_NAR RCCL all-reduces of 8 doubles
    // back to back on the context stream.  The slowest rank's figures are the communicator's (all-reduce max): the rule that
    // picks the route of a sweep (kk_xs_pays, kk_internal.h) reads the same numbers on every rank.
    {
        const int XS_NRED = 65, XS_NAR = 20;
        a.tag0 = k->xs_red + 1u; a.launch = ++k->xs_launch;
        k->xs_red += (unsigned)XS_NRED;
        long long* d_t = (long long*)(d_send + 520);
        KK_HIP(hipMemsetAsync(d_t, 0, 2 * sizeof(long long), c->stream));
        KK_TRY(kk_launch_xs_timing(c, a, XS_NRED, d_t));
        long long h_t[2] = {0, 0};
        KK_HIP(hipMemcpyAsync(h_t, d_t, sizeof(h_t), hipMemcpyDeviceToHost, c->stream));
        KK_HIP(hipStreamSynchronize(c->stream));
        double* d_ar = (double*)(d_send + 528);
        KK_HIP(hipMemsetAsync(d_ar, 0, 8 * sizeof(double), c->stream));
        KK_TRY(kk_comm_allreduce_sum(c, d_ar, 8));   // (warm-up: the first collective of a communicator sets up its channels)
        KK_HIP(hipEventRecord(c->t0, c->stream));
        for (int i = 0; i < XS_NAR; ++i) KK_TRY(kk_comm_allreduce_sum(c, d_ar, 8));
        KK_HIP(hipEventRecord(c->t1, c->stream));
        KK_HIP(hipEventSynchronize(c->t1));
        float ms = 0;
        KK_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
        k->n_allreduce -= XS_NAR + 1;   // (set-up traffic: not part of the caller's statistics)
        double fig[3] = {h_t[0] == 1 ? 0.0 : 1.0, (double)h_t[1] * 0.01 / (XS_NRED - 1), 1e3 * (double)ms / XS_NAR};   // failed?, us per in-kernel reduction, us per all-reduce
        KK_HIP(hipMemcpyAsync(d_ar, fig, sizeof(fig), hipMemcpyHostToDevice, c->stream));
        KK_NCCL(g_rccl.AllReduce(d_ar, d_ar, 3, ncclDouble, ncclMax, (ncclComm_t)k->nccl, c->stream));
        KK_HIP(hipMemcpyAsync(fig, d_ar, sizeof(fig), hipMemcpyDeviceToHost, c->stream));
        KK_HIP(hipStreamSynchronize(c->stream));
        if (fig[0] != 0.0) { xs_release(c); return KK_OK; }   // a rank's timing launch gave up: the RCCL routes serve
        k->xs_hop_us = fig[1]; k->ar_us = fig[2];
    }
    k->xs_active = true;
    if (cus_agreed != c->num_cus) { k->cus_before = c->num_cus; c->num_cus = cus_agreed; }
    return KK_OK;
}
void kk_xs_postmortem(kk_ctx ctx, const char* where) {
    static const bool on = getenv("KK_XSYNC_DEBUG") && atoi(getenv("KK_XSYNC_DEBUG")) != 0;
    kk_comm_s* k = ctx->comm;
    if (!on || !k || !k->xs_mine) return;
    unsigned h[16 + 64 + 2 * KK_XS_SET_BYTES / 4];
    if (hipMemcpy(h, k->xs_mine + KK_XS_DBG_OFFSET, (16 + 64) * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
    fprintf(stderr, "[kk_xsync rank %d] %s: host launch %u red %u | first give-up: launch %u tag %u nval %u why %u (1 peer abort, 2 local flag, 4 timeout) block %u red %u | tags seen:",
            k->rank, where, k->xs_launch, k->xs_red, h[0], h[1], h[2], h[3], h[4], h[5]);
    for (int l = 0; l < 64; ++l) if ((l >> 3) < (int)h[2] && (l & 7) < k->world) fprintf(stderr, " [v%d r%d]=%u", l >> 3, l & 7, h[16 + l]);
    fprintf(stderr, "\n");
}
kk_xs_dev kk_xs_launch_args(kk_ctx ctx, unsigned nred) {
    kk_xs_dev a;
    if (!kk_sharded(ctx) || !kk_xs_on(ctx)) return a;
    kk_comm_s* k = ctx->comm;
    a.table = k->xs_table; a.mine = k->xs_mine;
    if (k->xs_clear_word) {   // first persistent launch after a recovery: every store of the lost launch's id has landed (kk_xsync.h)
        k->xs_clear_word = false;
        (void)hipMemsetAsync(k->xs_mine + KK_XS_ERR_OFFSET, 0, sizeof(unsigned), ctx->stream);
    }
    a.tag0 = k->xs_red + 1u;
    k->xs_red += nred;
    if (++k->xs_launch == 0) ++k->xs_launch;   // (0 is the "no abort" value of the error word)
    a.launch = k->xs_launch;
    a.rank = k->rank; a.world = k->world;
    ++k->n_xs_launches;
    return a;
}

KK_API int kk_comm_init(kk_ctx c, const void* id128, int rank, int world, int flags) {
    KK_CHECK(c && id128, KK_ERR_INVALID, "kk_comm_init: null arg");
    KK_CHECK(world >= 1 && rank >= 0 && rank < world, KK_ERR_INVALID, "kk_comm_init: rank %d of %d", rank, world);
    KK_CHECK(!c->comm, KK_ERR_INVALID, "kk_comm_init: the context already has a communicator");
    KK_CHECK(!c->allreduce, KK_ERR_INVALID, "kk_comm_init: an all-reduce hook is installed (use one mechanism)");
    KK_TRY(rccl_load());
    KK_HIP(hipSetDevice(c->device));
    KK_HIP(hipStreamSynchronize(c->stream));
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    KK_NCCL(g_rccl.CommInitRank(&comm, world, id, rank));
    kk_comm_s* k = new kk_comm_s();
    static uint64_t next_uid = 0;
    k->nccl = comm; k->rank = rank; k->world = world; k->uid = ++next_uid;
    k->active = world > 1 || (flags & KK_COMM_FORCE_COLLECTIVES) != 0;
    c->comm = k;
    c->spec_owner = nullptr;   // a speculative apply enqueued before the switch carries an un-sharded alpha
    if (k->active) {           // in-kernel sum over the ranks for the persistent MGS kernels (falls back to the RCCL routes when unavailable)
        const int st = xs_setup(c);
        if (st != KK_OK) { (void)kk_comm_destroy(c); return st; }
    }
    return KK_OK;
}

KK_API int kk_comm_destroy(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (!c->comm) return KK_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    ncclComm_t comm = (ncclComm_t)c->comm->nccl;
    xs_release(c);
    delete c->comm;
    c->comm = nullptr;
    c->spec_owner = nullptr;
    if (comm && g_rccl.CommDestroy) KK_NCCL(g_rccl.CommDestroy(comm));
    return KK_OK;
}

KK_API int kk_comm_info(kk_ctx c, int* rank, int* world, int* rccl_version) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (world) *world = c->comm ? c->comm->world : 1;
    if (rccl_version) {
        *rccl_version = 0;
        if (c->comm && g_rccl.GetVersion) (void)g_rccl.GetVersion(rccl_version);
    }
    return KK_OK;
}

KK_API int kk_comm_stats(kk_ctx c, int64_t* n_allreduce, int64_t* n_p2p_groups, int64_t* n_gather) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (n_allreduce) *n_allreduce = c->comm ? c->comm->n_allreduce : 0;
    if (n_p2p_groups) *n_p2p_groups = c->comm ? c->comm->n_p2p : 0;
    if (n_gather) *n_gather = c->comm ? c->comm->n_gather : 0;
    return KK_OK;
}

int kk_comm_allreduce_sum(kk_ctx c, double* dev_ptr, int64_t count) {
    kk_comm_s* k = c->comm;
    if (!k || !k->active || count <= 0) return KK_OK;
    kk_prof_scope ps(c, "nccl_allreduce");   // stream time of the collective (waiting for the peers included) when profiling is on
    KK_NCCL(g_rccl.AllReduce(dev_ptr, dev_ptr, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)k->nccl, c->stream));
    ++k->n_allreduce;
    return KK_OK;
}

// caller-visible collective on the context stream (e.g. the max of a per-rank time); op: 0 sum, 1 max, 2 min
KK_API int kk_comm_allreduce(kk_ctx c, void* dev_ptr, int64_t count, int op) {
    KK_CHECK(c && (dev_ptr || count == 0), KK_ERR_INVALID, "kk_comm_allreduce: null arg");
    KK_CHECK(op >= 0 && op <= 2, KK_ERR_INVALID, "kk_comm_allreduce: op must be 0 (sum), 1 (max) or 2 (min)");
    kk_comm_s* k = c->comm;
    if (!k || !k->active || count <= 0) return KK_OK;
    const ncclRedOp_t ro = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    KK_NCCL(g_rccl.AllReduce(dev_ptr, dev_ptr, (size_t)count, ncclDouble, ro, (ncclComm_t)k->nccl, c->stream));
    ++k->n_allreduce;
    return KK_OK;
}

// *v = max over the ranks, on the host when the call returns (route_agree, kk_host.h)
int kk_comm_allreduce_max_host(kk_ctx c, double* v) {
    kk_comm_s* k = c->comm;
    if (!k || !k->active) return KK_OK;
    double* t = SCP(c, SC_TMP2);
    KK_HIP(hipMemcpyAsync(t, v, sizeof(double), hipMemcpyHostToDevice, c->stream));
    KK_NCCL(g_rccl.AllReduce(t, t, 1, ncclDouble, ncclMax, (ncclComm_t)k->nccl, c->stream));
    KK_HIP(hipMemcpyAsync(v, t, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    KK_HIP(hipStreamSynchronize(c->stream));
    return KK_OK;
}

// all ranks have reached this point and their streams are drained
KK_API int kk_comm_barrier(kk_ctx c) {
    KK_CHECK(c, KK_ERR_INVALID, "null ctx");
    if (c->comm && c->comm->world > 1) {
        double* t = SCP(c, SC_TMP2);
        KK_HIP(hipMemsetAsync(t, 0, sizeof(double), c->stream));
        KK_NCCL(g_rccl.AllReduce(t, t, 1, ncclDouble, ncclSum, (ncclComm_t)c->comm->nccl, c->stream));
    }
    KK_HIP(hipStreamSynchronize(c->stream));
    return KK_OK;
}

// A collective entry point validates its arguments locally and then agrees on the outcome BEFORE the first data
// collective: a rank that simply returned on bad input would leave its peers blocked in the exchange that follows.
int kk_comm_agree_status(kk_ctx c, int local, int* worst) {
    *worst = local;
    kk_comm_s* k = c->comm;
    if (!k || k->world == 1) return KK_OK;
    double* t = SCP(c, SC_TMP2);
    const double mine = (double)(-local);   // statuses are <= 0: the maximum of the negated codes is the worst one
    KK_HIP(hipMemcpyAsync(t, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
    KK_HIP(hipStreamSynchronize(c->stream));   // `mine` is a stack variable
    KK_NCCL(g_rccl.AllReduce(t, t, 1, ncclDouble, ncclMax, (ncclComm_t)k->nccl, c->stream));
    double w = 0;
    KK_HIP(hipMemcpyAsync(&w, t, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    KK_HIP(hipStreamSynchronize(c->stream));
    *worst = -(int)w;
    return KK_OK;
}

// ------------------------------------------------------------------------------------------
// internal: integer exchanges used while a sharded operator is set up, ghost exchange, gather / scatter
// ------------------------------------------------------------------------------------------
int kk_comm_allgather_i64(kk_ctx c, const int64_t* d_send, int64_t* d_recv, int64_t count) {
    kk_comm_s* k = c->comm;
    if (!k || k->world == 1) {
        KK_HIP(hipMemcpyAsync(d_recv, d_send, count * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
        return KK_OK;
    }
    KK_NCCL(g_rccl.AllGather(d_send, d_recv, (size_t)count, ncclInt64, (ncclComm_t)k->nccl, c->stream));
    return KK_OK;
}
// grouped point-to-point: segment q of d_send (send_counts[q] elements) goes to rank q, segment q of d_recv comes from it
static int p2p_group(kk_ctx c, const void* d_send, const int64_t* send_counts, void* d_recv, const int64_t* recv_counts,
                     ncclDataType_t dt) {
    kk_comm_s* k = c->comm;
    const size_t es = 8;
    if (!k || k->world == 1) {   // loop-back plan of a one-rank run: this rank serves itself with a device copy
        if (send_counts[0] > 0 && send_counts[0] == recv_counts[0])
            KK_HIP(hipMemcpyAsync(d_recv, d_send, (size_t)send_counts[0] * es, hipMemcpyDeviceToDevice, c->stream));
        return KK_OK;
    }
    kk_prof_scope ps(c, "nccl_p2p");
    KK_NCCL(g_rccl.GroupStart());
    int64_t so = 0, ro = 0;
    ncclResult_t bad = ncclSuccess;
    for (int q = 0; q < k->world && bad == ncclSuccess; ++q) {
        if (send_counts[q] > 0)
            bad = g_rccl.Send((const char*)d_send + so * es, (size_t)send_counts[q], dt, q, (ncclComm_t)k->nccl, c->stream);
        so += send_counts[q];
        if (bad == ncclSuccess && recv_counts[q] > 0)
            bad = g_rccl.Recv((char*)d_recv + ro * es, (size_t)recv_counts[q], dt, q, (ncclComm_t)k->nccl, c->stream);
        ro += recv_counts[q];
    }
    ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return rccl_fail(bad, "ncclSend/ncclRecv", __LINE__);
    if (end != ncclSuccess) return rccl_fail(end, "ncclGroupEnd", __LINE__);
    ++k->n_p2p;
    return KK_OK;
}
int kk_comm_exchange_i64(kk_ctx c, const int64_t* d_send, const int64_t* send_counts, int64_t* d_recv,
                         const int64_t* recv_counts) {
    return p2p_group(c, d_send, send_counts, d_recv, recv_counts, ncclInt64);
}

// fill the ghost buffer of a row-sharded operator from the vector x it is about to be applied to
int kk_halo_exchange(kk_ctx c, const kk_sparse_dev& M, const double* x) {
    const kk_halo_plan* p = M.plan;
    if (!p || (p->total_send == 0 && p->total_recv == 0)) return KK_OK;
    KK_CHECK((c->comm ? c->comm->world : 1) == (int)p->send_counts.size(), KK_ERR_INVALID,
             "ghost exchange: the operator was created for another communicator");
    if (p->total_send) KK_TRY(kk_launch_gather(c, x, p->d_send_idx, p->total_send, p->d_sendbuf));
    return p2p_group(c, p->d_sendbuf, p->send_counts.data(), p->d_ghost, p->recv_counts.data(), ncclDouble);
}

// ghost exchange for a block of nb <= 16 vectors: nb gathers, then ONE group with nb sends / receives per peer
int kk_halo_exchange_block(kk_ctx c, const kk_sparse_dev& M, const double* X, int64_t ldx, int nb, const double** G, int64_t* ldg) {
    kk_halo_plan* p = M.plan;
    KK_CHECK(p && nb >= 1 && nb <= 16, KK_ERR_INVALID, "block ghost exchange: bad arguments");
    const int64_t ng = std::max<int64_t>(p->total_recv, 1), ns = std::max<int64_t>(p->total_send, 1);
    if (!p->d_ghost_blk) {
        KK_HIP(hipMalloc(&p->d_ghost_blk, (size_t)16 * ng * sizeof(double)));
        KK_HIP(hipMalloc(&p->d_sendbuf_blk, (size_t)16 * ns * sizeof(double)));
        KK_HIP(hipMemsetAsync(p->d_ghost_blk, 0, (size_t)16 * ng * sizeof(double), c->stream));
    }
    *G = p->d_ghost_blk;
    *ldg = ng;
    if (p->total_send == 0 && p->total_recv == 0) return KK_OK;
    kk_comm_s* k = c->comm;
    KK_CHECK((k ? k->world : 1) == (int)p->send_counts.size(), KK_ERR_INVALID, "ghost exchange: the operator was created for another communicator");
    for (int j = 0; j < nb; ++j)
        if (p->total_send) KK_TRY(kk_launch_gather(c, X + (int64_t)j * ldx, p->d_send_idx, p->total_send, p->d_sendbuf_blk + (int64_t)j * ns));
    if (!k || k->world == 1) {   // loop-back plan: serve myself
        for (int j = 0; j < nb; ++j)
            KK_HIP(hipMemcpyAsync(p->d_ghost_blk + (int64_t)j * ng, p->d_sendbuf_blk + (int64_t)j * ns, (size_t)p->total_send * sizeof(double),
                                  hipMemcpyDeviceToDevice, c->stream));
        return KK_OK;
    }
    kk_prof_scope ps(c, "nccl_p2p");
    KK_NCCL(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int j = 0; j < nb && bad == ncclSuccess; ++j) {
        int64_t so = 0, ro = 0;
        for (int q = 0; q < k->world && bad == ncclSuccess; ++q) {
            if (p->send_counts[q] > 0)
                bad = g_rccl.Send(p->d_sendbuf_blk + (int64_t)j * ns + so, (size_t)p->send_counts[q], ncclDouble, q, (ncclComm_t)k->nccl, c->stream);
            so += p->send_counts[q];
            if (bad == ncclSuccess && p->recv_counts[q] > 0)
                bad = g_rccl.Recv(p->d_ghost_blk + (int64_t)j * ng + ro, (size_t)p->recv_counts[q], ncclDouble, q, (ncclComm_t)k->nccl, c->stream);
            ro += p->recv_counts[q];
        }
    }
    ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return rccl_fail(bad, "ncclSend/ncclRecv (block)", __LINE__);
    if (end != ncclSuccess) return rccl_fail(end, "ncclGroupEnd (block)", __LINE__);
    ++k->n_p2p;
    return KK_OK;
}

// vfull = all-gather of the `shard`-strided local pieces (stage); world 1: plain copy
int kk_comm_allgather_f64(kk_ctx c, const double* d_stage, double* d_full, int64_t shard) {
    kk_comm_s* k = c->comm;
    if (!k || !k->active) {
        KK_HIP(hipMemcpyAsync(d_full, d_stage, shard * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return KK_OK;
    }
    kk_prof_scope ps(c, "nccl_gather");
    KK_NCCL(g_rccl.AllGather(d_stage, d_full, (size_t)shard, ncclDouble, (ncclComm_t)k->nccl, c->stream));
    ++k->n_gather;
    return KK_OK;
}
// stage = my shard of the sum over ranks of d_full
int kk_comm_reducescatter_f64(kk_ctx c, const double* d_full, double* d_stage, int64_t shard) {
    kk_comm_s* k = c->comm;
    if (!k || !k->active) {
        KK_HIP(hipMemcpyAsync(d_stage, d_full, shard * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return KK_OK;
    }
    kk_prof_scope ps(c, "nccl_gather");
    KK_NCCL(g_rccl.ReduceScatter(d_full, d_stage, (size_t)shard, ncclDouble, ncclSum, (ncclComm_t)k->nccl, c->stream));
    ++k->n_gather;
    return KK_OK;
}
