// TEST INFRASTRUCTURE -- never shipped, never linked into libkrylov_hip.so.
//
// libfake_rccl.so: the twelve nccl* entry points csrc/kk_comm.hip binds (rccl_load), implemented over a memory-mapped
// file + host staging, so that TWO (or more) PROCESSES SHARING ONE GPU form a communicator of world size > 1.  Real RCCL
// refuses that layout ("invalid usage": two ranks on one device), and the gpurun boxes have exactly one GPU; with
// KK_RCCL_LIB=<this file> the library's native multi-rank path -- kk_comm_init with world > 1, the ghost-plan
// negotiation of kk_csr_create_sharded, the grouped send / recv of every sparse apply, all-gather / reduce-scatter of
// the rectangular map, the all-reduces of every finalize site -- executes for real, rank against rank, before an
// 8-GPU node ever sees it.  Semantics follow the NCCL API contract the library relies on:
//   * collectives are matched by call order on every rank; send/recv pairs are matched per (source, destination) in
//     FIFO order; operations inside ncclGroupStart/End are issued together and may be mutually dependent;
//   * results are stream-ordered.  Default mode: every call drains the stream, stages through host memory and is complete
//     on return (stronger than the contract, never weaker).  KK_FAKE_RCCL_ASYNC=1: the call only ENQUEUES -- a copy of the
//     send data into pinned staging, a host function on the caller's stream that waits KK_FAKE_RCCL_DELAY_US (default 300)
//     and then does the exchange with the peers, a copy of the result to the receive buffer -- and returns at once, as RCCL
//     does.  In that mode a host read of a "result" that is not behind a stream / event synchronisation, or a kernel on
//     another stream that consumes it without an event, sees stale data: the class of bug the synchronous mode hides;
//   * sum / max / min all-reduces combine the ranks in rank order 0..world-1 on every rank, so every rank receives
//     bit-identical results (RCCL's ring / tree all-reduce gives the same guarantee).
// Every wait is bounded (KK_FAKE_RCCL_TIMEOUT seconds, default 120): a missing peer produces ncclSystemError, not a hang.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct fake_comm;
typedef fake_comm* ncclComm_t;
}

namespace {
constexpr uint32_t kMagic = 0x4b4b4643;   // "KKFC"
constexpr size_t kHeaderBytes = 4096;
constexpr size_t kMboxHeader = 64;

struct shm_header {
    std::atomic<uint32_t> magic;
    std::atomic<int> attached;
    std::atomic<int> detached;
    std::atomic<uint32_t> bar_count;
    std::atomic<uint32_t> bar_gen;
    std::atomic<int> aborted;
};
struct mbox {   // one per ordered pair (src -> dst): a single-chunk FIFO
    std::atomic<uint64_t> written;
    std::atomic<uint64_t> consumed;
    uint64_t chunk_bytes;
};
static_assert(sizeof(shm_header) <= kHeaderBytes && sizeof(mbox) <= kMboxHeader, "layout");

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
double timeout_s() {
    static double t = [] {
        const char* e = getenv("KK_FAKE_RCCL_TIMEOUT");
        return e && *e ? atof(e) : 120.0;
    }();
    return t;
}
size_t env_mb(const char* name, size_t def_mb) {
    const char* e = getenv(name);
    return (e && *e ? (size_t)atoll(e) : def_mb) << 20;
}
size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
thread_local int g_group_depth = 0;
struct p2p_op {
    bool is_send;
    fake_comm* comm;
    void* dev;
    size_t bytes;
    int peer;
    hipStream_t stream;
    std::vector<char> host;
    size_t done = 0;
};
thread_local std::vector<p2p_op> g_group_ops;
std::atomic<uint64_t> g_id_counter{0};
}  // namespace

struct fake_comm {
    int rank = 0, world = 1;
    char path[128] = {0};
    char* base = nullptr;
    size_t total = 0, slot_bytes = 0, mbox_bytes = 0;
    std::vector<char> tmp;
    uint64_t n_coll = 0, n_p2p = 0;
    shm_header* hdr() const { return (shm_header*)base; }
    char* slot(int r) const { return base + kHeaderBytes + (size_t)r * slot_bytes; }
    mbox* mb(int src, int dst) const {
        return (mbox*)(base + kHeaderBytes + (size_t)world * slot_bytes + ((size_t)src * world + dst) * (kMboxHeader + mbox_bytes));
    }
    char* mb_data(int src, int dst) const { return (char*)mb(src, dst) + kMboxHeader; }
    bool dead() const { return hdr()->aborted.load(std::memory_order_acquire) != 0; }
    ncclResult_t fail(const char* what) {
        hdr()->aborted.store(1, std::memory_order_release);
        fprintf(stderr, "fake_rccl: rank %d of %d: %s (timeout %.0f s) -- communicator aborted\n", rank, world, what, timeout_s());
        fflush(stderr);
        return ncclSystemError;
    }
    ncclResult_t barrier() {
        if (world == 1) return ncclSuccess;
        shm_header* h = hdr();
        const uint32_t gen = h->bar_gen.load(std::memory_order_acquire);
        if (h->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
            h->bar_count.store(0, std::memory_order_relaxed);
            h->bar_gen.fetch_add(1, std::memory_order_acq_rel);
            return ncclSuccess;
        }
        const double t0 = now_s();
        int spins = 0;
        while (h->bar_gen.load(std::memory_order_acquire) == gen) {
            if (dead()) return ncclSystemError;
            if (++spins > 200) { sched_yield(); if ((spins & 1023) == 0 && now_s() - t0 > timeout_s()) return fail("a peer never reached the barrier"); }
        }
        return ncclSuccess;
    }
};

namespace {
#define FK_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            fprintf(stderr, "fake_rccl: %s failed: %s\n", #call, hipGetErrorString(_e));               \
            return ncclUnhandledCudaError;                                                             \
        }                                                                                              \
    } while (0)
#define FK_TRY(call)                          \
    do {                                      \
        ncclResult_t _r = (call);             \
        if (_r != ncclSuccess) return _r;     \
    } while (0)

// KK_FAKE_RCCL_HOSTMEM=1: the "device" pointers are host pointers (self-test of this shim on a box without a GPU)
bool hostmem() {
    static bool h = [] { const char* e = getenv("KK_FAKE_RCCL_HOSTMEM"); return e && *e == '1'; }();
    return h;
}
ncclResult_t d2h(void* host, const void* dev, size_t bytes, hipStream_t s) {
    if (!bytes) return ncclSuccess;
    if (hostmem()) { memcpy(host, dev, bytes); return ncclSuccess; }
    FK_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
    FK_HIP(hipStreamSynchronize(s));
    return ncclSuccess;
}
ncclResult_t h2d(void* dev, const void* host, size_t bytes, hipStream_t s) {
    if (!bytes) return ncclSuccess;
    if (hostmem()) { memcpy(dev, host, bytes); return ncclSuccess; }
    FK_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
    FK_HIP(hipStreamSynchronize(s));
    return ncclSuccess;
}

template <typename T>
void combine(T* acc, const T* x, size_t n, ncclRedOp_t op) {
    switch (op) {
        case ncclSum: for (size_t i = 0; i < n; ++i) acc[i] = acc[i] + x[i]; break;
        case ncclProd: for (size_t i = 0; i < n; ++i) acc[i] = acc[i] * x[i]; break;
        case ncclMax: for (size_t i = 0; i < n; ++i) acc[i] = x[i] > acc[i] ? x[i] : acc[i]; break;
        case ncclMin: for (size_t i = 0; i < n; ++i) acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
    }
}
ncclResult_t combine_any(void* acc, const void* x, size_t n, ncclDataType_t dt, ncclRedOp_t op) {
    switch (dt) {
        case ncclFloat64: combine((double*)acc, (const double*)x, n, op); return ncclSuccess;
        case ncclFloat32: combine((float*)acc, (const float*)x, n, op); return ncclSuccess;
        case ncclInt64: combine((int64_t*)acc, (const int64_t*)x, n, op); return ncclSuccess;
        case ncclUint64: combine((uint64_t*)acc, (const uint64_t*)x, n, op); return ncclSuccess;
        case ncclInt32: combine((int32_t*)acc, (const int32_t*)x, n, op); return ncclSuccess;
        case ncclUint32: combine((uint32_t*)acc, (const uint32_t*)x, n, op); return ncclSuccess;
        default: return ncclInvalidArgument;
    }
}

// run a batch of sends / receives to completion: every operation advances whenever its mailbox allows it, so mutually
// dependent operations of one group (A sends to B while B sends to A, several messages per peer) cannot deadlock
bool tracing() {
    static bool t = [] { const char* e = getenv("KK_FAKE_RCCL_TRACE"); return e && *e == '1'; }();
    return t;
}
ncclResult_t run_p2p(std::vector<p2p_op>& ops) {
    if (ops.empty()) return ncclSuccess;
    if (tracing()) {
        fprintf(stderr, "fake_rccl[%d]: p2p batch:", ops[0].comm->rank);
        for (p2p_op& o : ops) fprintf(stderr, " %s%d:%zu", o.is_send ? "S" : "R", o.peer, o.bytes);
        fprintf(stderr, "\n");
    }
    for (p2p_op& o : ops) {
        o.host.resize(o.bytes);
        if (o.is_send) FK_TRY(d2h(o.host.data(), o.dev, o.bytes, o.stream));
    }
    const double t0 = now_s();
    size_t open = ops.size();
    for (p2p_op& o : ops)
        if (o.bytes == 0) --open;
    int idle = 0;
    while (open) {
        bool progressed = false;
        // FIFO per (src, dst): only the first unfinished operation of each direction and peer may touch the mailbox
        for (size_t i = 0; i < ops.size(); ++i) {
            p2p_op& o = ops[i];
            if (o.done == o.bytes) continue;
            bool first = true;
            for (size_t j = 0; j < i && first; ++j)
                if (ops[j].is_send == o.is_send && ops[j].peer == o.peer && ops[j].comm == o.comm && ops[j].done < ops[j].bytes) first = false;
            if (!first) continue;
            fake_comm* c = o.comm;
            if (c->dead()) return ncclSystemError;
            const size_t chunk = o.bytes - o.done < c->mbox_bytes ? o.bytes - o.done : c->mbox_bytes;
            if (o.is_send) {
                mbox* m = c->mb(c->rank, o.peer);
                const uint64_t w = m->written.load(std::memory_order_relaxed);
                if (m->consumed.load(std::memory_order_acquire) != w) continue;   // previous chunk not taken yet
                memcpy(c->mb_data(c->rank, o.peer), o.host.data() + o.done, chunk);
                m->chunk_bytes = chunk;
                m->written.store(w + 1, std::memory_order_release);
            } else {
                mbox* m = c->mb(o.peer, c->rank);
                const uint64_t r = m->consumed.load(std::memory_order_relaxed);
                if (m->written.load(std::memory_order_acquire) == r) continue;    // nothing there yet
                if (m->chunk_bytes != chunk) {
                    fprintf(stderr, "fake_rccl: rank %d: recv from %d expects a %zu-byte chunk, the peer sent %llu (mismatched send/recv sizes)\n",
                            c->rank, o.peer, chunk, (unsigned long long)m->chunk_bytes);
                    c->hdr()->aborted.store(1);
                    return ncclInvalidUsage;
                }
                memcpy(o.host.data() + o.done, c->mb_data(o.peer, c->rank), chunk);
                m->consumed.store(r + 1, std::memory_order_release);
            }
            o.done += chunk;
            progressed = true;
            if (o.done == o.bytes) --open;
        }
        if (progressed) { idle = 0; continue; }
        if (++idle > 200) {
            sched_yield();
            if ((idle & 1023) == 0 && now_s() - t0 > timeout_s()) return ops[0].comm->fail("a send / recv never found its partner");
        }
    }
    for (p2p_op& o : ops)
        if (!o.is_send) FK_TRY(h2d(o.dev, o.host.data(), o.bytes, o.stream));
    ops[0].comm->n_p2p++;
    return ncclSuccess;
}
}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------------------------------
// asynchronous mode (KK_FAKE_RCCL_ASYNC=1)
// ---------------------------------------------------------------------------------------------------------------------
bool async_mode() {
    static bool a = [] { const char* e = getenv("KK_FAKE_RCCL_ASYNC"); return e && *e == '1' && !hostmem(); }();
    return a;
}
long delay_us() {
    static long d = [] { const char* e = getenv("KK_FAKE_RCCL_DELAY_US"); return e && *e ? atol(e) : 300L; }();
    return d;
}
// pinned staging: a ring of slots, each reused only after the stream has passed the operation that used it last
struct pin_slot { char* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
constexpr int kPinSlots = 64;
pin_slot g_pin[kPinSlots];
int g_pin_next = 0;
char* pin_get(size_t bytes, int* idx) {
    pin_slot& sl = g_pin[g_pin_next];
    *idx = g_pin_next;
    g_pin_next = (g_pin_next + 1) % kPinSlots;
    if (!sl.ev && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (sl.used && hipEventSynchronize(sl.ev) != hipSuccess) return nullptr;
    if (sl.cap < bytes) {
        if (sl.p) (void)hipHostFree(sl.p);
        sl.cap = (bytes + 4095) / 4096 * 4096;
        if (hipHostMalloc((void**)&sl.p, sl.cap, hipHostMallocDefault) != hipSuccess) { sl.p = nullptr; sl.cap = 0; return nullptr; }
    }
    sl.used = true;
    return sl.p;
}
ncclResult_t pin_done(int idx, hipStream_t s) {
    FK_HIP(hipEventRecord(g_pin[idx].ev, s));
    return ncclSuccess;
}

struct seg { bool is_send; int peer; size_t bytes; size_t off; size_t done; };   // p2p segment inside the job's staging
struct job {
    int kind;   // 0 all-reduce, 1 all-gather, 2 reduce-scatter, 3 p2p batch
    fake_comm* c;
    char* sendh; char* recvh;
    size_t count;   // elements (per rank for gather / scatter)
    ncclDataType_t dt; ncclRedOp_t op;
    std::vector<seg> segs;
};
void job_fail(fake_comm* c, const char* what) {
    c->hdr()->aborted.store(1, std::memory_order_release);
    fprintf(stderr, "fake_rccl(async): rank %d of %d: %s -- communicator aborted\n", c->rank, c->world, what);
    fflush(stderr);
}
// the exchange itself, on host memory, run by the stream's host function
void run_job(void* arg) {
    job* j = (job*)arg;
    fake_comm* c = j->c;
    if (delay_us() > 0) usleep((useconds_t)delay_us());
    const size_t es = dtype_size(j->dt);
    if (j->kind == 0) {
        const size_t per = c->slot_bytes / es;
        for (size_t off = 0; off < j->count || (j->count == 0 && off == 0); off += per) {
            const size_t n = j->count - off < per ? j->count - off : per;
            if (n) memcpy(c->slot(c->rank), j->sendh + off * es, n * es);
            if (c->barrier() != ncclSuccess) { job_fail(c, "all-reduce: a peer is missing"); break; }
            char* out = j->recvh + off * es;
            if (n) memcpy(out, c->slot(0), n * es);
            for (int r = 1; r < c->world; ++r) (void)combine_any(out, c->slot(r), n, j->dt, j->op);
            if (c->barrier() != ncclSuccess) { job_fail(c, "all-reduce: a peer is missing"); break; }
            if (j->count == 0) break;
        }
    } else if (j->kind == 1) {
        const size_t per = c->slot_bytes / es;
        for (size_t off = 0; off < j->count || (j->count == 0 && off == 0); off += per) {
            const size_t n = j->count - off < per ? j->count - off : per;
            if (n) memcpy(c->slot(c->rank), j->sendh + off * es, n * es);
            if (c->barrier() != ncclSuccess) { job_fail(c, "all-gather: a peer is missing"); break; }
            for (int r = 0; r < c->world; ++r)
                if (n) memcpy(j->recvh + ((size_t)r * j->count + off) * es, c->slot(r), n * es);
            if (c->barrier() != ncclSuccess) { job_fail(c, "all-gather: a peer is missing"); break; }
            if (j->count == 0) break;
        }
    } else if (j->kind == 2) {
        const size_t per = c->slot_bytes / es / (size_t)c->world;
        for (size_t off = 0; off < j->count || (j->count == 0 && off == 0); off += per) {
            const size_t n = j->count - off < per ? j->count - off : per;
            for (int q = 0; q < c->world; ++q)
                if (n) memcpy(c->slot(c->rank) + (size_t)q * n * es, j->sendh + ((size_t)q * j->count + off) * es, n * es);
            if (c->barrier() != ncclSuccess) { job_fail(c, "reduce-scatter: a peer is missing"); break; }
            char* out = j->recvh + off * es;
            if (n) memcpy(out, c->slot(0) + (size_t)c->rank * n * es, n * es);
            for (int r = 1; r < c->world; ++r) (void)combine_any(out, c->slot(r) + (size_t)c->rank * n * es, n, j->dt, j->op);
            if (c->barrier() != ncclSuccess) { job_fail(c, "reduce-scatter: a peer is missing"); break; }
            if (j->count == 0) break;
        }
    } else {
        // p2p batch: the progress loop of run_p2p on the staged segments (sends read sendh + off, receives fill recvh + off)
        std::vector<seg>& ops = j->segs;
        const double t0 = now_s();
        size_t open = 0;
        for (seg& o : ops) if (o.bytes) ++open;
        int idle = 0;
        while (open) {
            bool progressed = false;
            for (size_t i = 0; i < ops.size(); ++i) {
                seg& o = ops[i];
                if (o.done == o.bytes) continue;
                bool first = true;
                for (size_t k = 0; k < i && first; ++k)
                    if (ops[k].is_send == o.is_send && ops[k].peer == o.peer && ops[k].done < ops[k].bytes) first = false;
                if (!first) continue;
                if (c->dead()) { open = 0; break; }
                const size_t chunk = o.bytes - o.done < c->mbox_bytes ? o.bytes - o.done : c->mbox_bytes;
                if (o.is_send) {
                    mbox* m = c->mb(c->rank, o.peer);
                    const uint64_t w = m->written.load(std::memory_order_relaxed);
                    if (m->consumed.load(std::memory_order_acquire) != w) continue;
                    memcpy(c->mb_data(c->rank, o.peer), j->sendh + o.off + o.done, chunk);
                    m->chunk_bytes = chunk;
                    m->written.store(w + 1, std::memory_order_release);
                } else {
                    mbox* m = c->mb(o.peer, c->rank);
                    const uint64_t r = m->consumed.load(std::memory_order_relaxed);
                    if (m->written.load(std::memory_order_acquire) == r) continue;
                    if (m->chunk_bytes != chunk) { job_fail(c, "mismatched send / recv sizes"); open = 0; break; }
                    memcpy(j->recvh + o.off + o.done, c->mb_data(o.peer, c->rank), chunk);
                    m->consumed.store(r + 1, std::memory_order_release);
                }
                o.done += chunk;
                progressed = true;
                if (o.done == o.bytes) --open;
            }
            if (progressed) { idle = 0; continue; }
            if (++idle > 200) {
                sched_yield();
                if ((idle & 1023) == 0 && now_s() - t0 > timeout_s()) { job_fail(c, "a send / recv never found its partner"); break; }
            }
        }
    }
    delete j;
}
// enqueue: stage the send data, the host function, the delivery of the result
ncclResult_t enqueue_collective(int kind, const void* send, void* recv, size_t send_bytes, size_t recv_bytes, size_t count, ncclDataType_t dt,
                                ncclRedOp_t op, fake_comm* c, hipStream_t s) {
    int idx;
    char* st = pin_get(send_bytes + recv_bytes + 64, &idx);
    if (!st) return ncclUnhandledCudaError;
    job* j = new job();
    j->kind = kind; j->c = c; j->sendh = st; j->recvh = st + (send_bytes + 63) / 64 * 64; j->count = count; j->dt = dt; j->op = op;
    if (send_bytes) FK_HIP(hipMemcpyAsync(j->sendh, send, send_bytes, hipMemcpyDeviceToHost, s));
    FK_HIP(hipLaunchHostFunc(s, run_job, j));
    if (recv_bytes) FK_HIP(hipMemcpyAsync(recv, j->recvh, recv_bytes, hipMemcpyHostToDevice, s));
    return pin_done(idx, s);
}
ncclResult_t enqueue_p2p(std::vector<p2p_op>& ops) {
    if (ops.empty()) return ncclSuccess;
    size_t sb = 0, rb = 0;
    for (p2p_op& o : ops) (o.is_send ? sb : rb) += (o.bytes + 63) / 64 * 64;
    int idx;
    char* st = pin_get(sb + rb + 64, &idx);
    if (!st) return ncclUnhandledCudaError;
    job* j = new job();
    j->kind = 3; j->c = ops[0].comm; j->sendh = st; j->recvh = st + sb; j->count = 0; j->dt = ncclUint8; j->op = ncclSum;
    hipStream_t s = ops[0].stream;
    size_t so = 0, ro = 0;
    for (p2p_op& o : ops) {
        seg g{o.is_send, o.peer, o.bytes, o.is_send ? so : ro, 0};
        if (o.is_send) {
            if (o.bytes) FK_HIP(hipMemcpyAsync(j->sendh + so, o.dev, o.bytes, hipMemcpyDeviceToHost, s));
            so += (o.bytes + 63) / 64 * 64;
        } else {
            ro += (o.bytes + 63) / 64 * 64;
        }
        j->segs.push_back(g);
    }
    std::vector<seg> recvs;
    for (size_t i = 0; i < ops.size(); ++i)
        if (!ops[i].is_send) recvs.push_back(j->segs[i]);
    char* recvh = j->recvh;
    FK_HIP(hipLaunchHostFunc(s, run_job, j));
    size_t k = 0;
    for (p2p_op& o : ops)
        if (!o.is_send) {
            if (o.bytes) FK_HIP(hipMemcpyAsync(o.dev, recvh + recvs[k].off, o.bytes, hipMemcpyHostToDevice, s));
            ++k;
        }
    ops[0].comm->n_p2p++;
    return pin_done(idx, s);
}
}  // namespace

extern "C" {
#define FK_EXPORT __attribute__((visibility("default")))

FK_EXPORT ncclResult_t ncclGetVersion(int* v) {
    if (!v) return ncclInvalidArgument;
    *v = 29999;   // recognisable: no RCCL release carries this number
    return ncclSuccess;
}
FK_EXPORT const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclSystemError: return "system error / peer missing (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl)";
        default: return "internal error (fake_rccl)";
    }
}
FK_EXPORT ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof(id->internal));
    const char* dir = getenv("KK_FAKE_RCCL_DIR");
    if (!dir || !*dir) dir = "/tmp";
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "%s/kkfake_rccl_%d_%lld_%llu", dir, (int)getpid(),
             (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec, (unsigned long long)g_id_counter.fetch_add(1));
    return ncclSuccess;
}
FK_EXPORT ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    if (id.internal[sizeof(id.internal) - 1] != 0 || id.internal[0] != '/') return ncclInvalidArgument;
    fake_comm* c = new fake_comm();
    c->rank = rank; c->world = world;
    memcpy(c->path, id.internal, sizeof(c->path));
    c->slot_bytes = env_mb("KK_FAKE_RCCL_SLOT_MB", 16);
    c->mbox_bytes = env_mb("KK_FAKE_RCCL_MBOX_MB", 1);
    c->total = kHeaderBytes + (size_t)world * c->slot_bytes + (size_t)world * world * (kMboxHeader + c->mbox_bytes);
    const int fd = open(c->path, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { fprintf(stderr, "fake_rccl: open(%s): %s\n", c->path, strerror(errno)); delete c; return ncclSystemError; }
    // every rank sizes the (sparse) file identically; fresh pages read as zeros, which IS the initial state of all counters
    if (ftruncate(fd, (off_t)c->total) != 0) { fprintf(stderr, "fake_rccl: ftruncate: %s\n", strerror(errno)); close(fd); delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { fprintf(stderr, "fake_rccl: mmap: %s\n", strerror(errno)); delete c; return ncclSystemError; }
    c->base = (char*)p;
    shm_header* h = c->hdr();
    h->magic.store(kMagic);
    h->attached.fetch_add(1, std::memory_order_acq_rel);
    const double t0 = now_s();
    int spins = 0;
    while (h->attached.load(std::memory_order_acquire) < world) {
        if (++spins > 200) {
            sched_yield();
            if ((spins & 1023) == 0 && now_s() - t0 > timeout_s()) {
                fprintf(stderr, "fake_rccl: rank %d: only %d of %d ranks attached to %s\n", rank, h->attached.load(), world, c->path);
                munmap(c->base, c->total); unlink(c->path); delete c;
                return ncclSystemError;
            }
        }
    }
    ncclResult_t r = c->barrier();
    if (r != ncclSuccess) { munmap(c->base, c->total); delete c; return r; }
    if (rank == 0) unlink(c->path);   // everybody holds a mapping: the name can go, nothing is left behind after the run
    *out = c;
    return ncclSuccess;
}
FK_EXPORT ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    if (async_mode()) (void)hipDeviceSynchronize();   // (host functions still in a stream hold pointers into the mapping)
    c->hdr()->detached.fetch_add(1);
    munmap(c->base, c->total);
    delete c;
    return ncclSuccess;
}

FK_EXPORT ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c,
                                     hipStream_t s) {
    if (!c || (count && (!send || !recv))) return ncclInvalidArgument;
    if (g_group_depth) return ncclInvalidUsage;   // collectives inside a group are not needed by libkrylov_hip
    if (tracing()) fprintf(stderr, "fake_rccl[%d]: allreduce %zu x dtype %d op %d\n", c->rank, count, (int)dt, (int)op);
    if (async_mode()) { c->n_coll++; return enqueue_collective(0, send, recv, count * dtype_size(dt), count * dtype_size(dt), count, dt, op, c, s); }
    const size_t es = dtype_size(dt), per = c->slot_bytes / es;
    for (size_t off = 0; off < count || (count == 0 && off == 0); off += per) {
        const size_t n = count - off < per ? count - off : per;
        FK_TRY(d2h(c->slot(c->rank), (const char*)send + off * es, n * es, s));
        FK_TRY(c->barrier());
        c->tmp.resize(n * es);
        if (n) memcpy(c->tmp.data(), c->slot(0), n * es);
        for (int r = 1; r < c->world; ++r) FK_TRY(combine_any(c->tmp.data(), c->slot(r), n, dt, op));
        FK_TRY(c->barrier());   // every rank has read every slot: they may be overwritten
        FK_TRY(h2d((char*)recv + off * es, c->tmp.data(), n * es, s));
        if (count == 0) break;
    }
    c->n_coll++;
    return ncclSuccess;
}
FK_EXPORT ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t dt, ncclComm_t c, hipStream_t s) {
    if (!c || (sendcount && (!send || !recv))) return ncclInvalidArgument;
    if (g_group_depth) return ncclInvalidUsage;
    if (tracing()) fprintf(stderr, "fake_rccl[%d]: allgather %zu x dtype %d\n", c->rank, sendcount, (int)dt);
    if (async_mode()) { c->n_coll++; return enqueue_collective(1, send, recv, sendcount * dtype_size(dt), sendcount * dtype_size(dt) * c->world, sendcount, dt, ncclSum, c, s); }
    const size_t es = dtype_size(dt), per = c->slot_bytes / es;
    for (size_t off = 0; off < sendcount || (sendcount == 0 && off == 0); off += per) {
        const size_t n = sendcount - off < per ? sendcount - off : per;
        FK_TRY(d2h(c->slot(c->rank), (const char*)send + off * es, n * es, s));
        FK_TRY(c->barrier());
        c->tmp.resize(n * es * c->world);
        for (int r = 0; r < c->world; ++r)
            if (n) memcpy(c->tmp.data() + (size_t)r * n * es, c->slot(r), n * es);
        FK_TRY(c->barrier());
        for (int r = 0; r < c->world; ++r)
            FK_TRY(h2d((char*)recv + ((size_t)r * sendcount + off) * es, c->tmp.data() + (size_t)r * n * es, n * es, s));
        if (sendcount == 0) break;
    }
    c->n_coll++;
    return ncclSuccess;
}
FK_EXPORT ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c,
                                         hipStream_t s) {
    if (!c || (recvcount && (!send || !recv))) return ncclInvalidArgument;
    if (g_group_depth) return ncclInvalidUsage;
    if (tracing()) fprintf(stderr, "fake_rccl[%d]: reducescatter %zu x dtype %d\n", c->rank, recvcount, (int)dt);
    const size_t es = dtype_size(dt), per = c->slot_bytes / es / (size_t)c->world;
    if (per == 0) return ncclInternalError;
    if (async_mode()) { c->n_coll++; return enqueue_collective(2, send, recv, recvcount * es * c->world, recvcount * es, recvcount, dt, op, c, s); }
    for (size_t off = 0; off < recvcount || (recvcount == 0 && off == 0); off += per) {
        const size_t n = recvcount - off < per ? recvcount - off : per;
        // my contribution to every destination block, laid out [dest][n] in my slot
        for (int q = 0; q < c->world; ++q)
            FK_TRY(d2h(c->slot(c->rank) + (size_t)q * n * es, (const char*)send + ((size_t)q * recvcount + off) * es, n * es, s));
        FK_TRY(c->barrier());
        c->tmp.resize(n * es);
        if (n) memcpy(c->tmp.data(), c->slot(0) + (size_t)c->rank * n * es, n * es);
        for (int r = 1; r < c->world; ++r) FK_TRY(combine_any(c->tmp.data(), c->slot(r) + (size_t)c->rank * n * es, n, dt, op));
        FK_TRY(c->barrier());
        FK_TRY(h2d((char*)recv + off * es, c->tmp.data(), n * es, s));
        if (recvcount == 0) break;
    }
    c->n_coll++;
    return ncclSuccess;
}

FK_EXPORT ncclResult_t ncclGroupStart() {
    ++g_group_depth;
    return ncclSuccess;
}
FK_EXPORT ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    std::vector<p2p_op> ops;
    ops.swap(g_group_ops);
    return async_mode() ? enqueue_p2p(ops) : run_p2p(ops);
}
static ncclResult_t p2p(bool is_send, void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    if (!c || peer < 0 || peer >= c->world || (count && !buf)) return ncclInvalidArgument;
    p2p_op o;
    o.is_send = is_send; o.comm = c; o.dev = buf; o.bytes = count * dtype_size(dt); o.peer = peer; o.stream = s;
    if (g_group_depth) {
        g_group_ops.push_back(std::move(o));
        return ncclSuccess;
    }
    std::vector<p2p_op> one;
    one.push_back(std::move(o));
    return async_mode() ? enqueue_p2p(one) : run_p2p(one);
}
FK_EXPORT ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    return p2p(true, const_cast<void*>(buf), count, dt, peer, c, s);
}
FK_EXPORT ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    return p2p(false, buf, count, dt, peer, c, s);
}
// test hook: how many collectives / p2p batches this communicator carried
FK_EXPORT void kkfake_rccl_stats(ncclComm_t c, uint64_t* n_coll, uint64_t* n_p2p) {
    if (n_coll) *n_coll = c ? c->n_coll : 0;
    if (n_p2p) *n_p2p = c ? c->n_p2p : 0;
}
}
