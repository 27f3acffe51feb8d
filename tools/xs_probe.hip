// Probe for the cross-rank in-kernel reduction (kk_xsync): two PROCESSES on one GPU (the gpurun box has one), each with an
// IPC-shared fine-grained sync area, each running a persistent kernel of G blocks that needs a whole CU per block.
//   ./xs_probe <rank> <G> <rounds> <dir>     (run rank 0 and rank 1 concurrently; optional HSA_CU_MASK per process)
// Measures: does hipIpcGetMemHandle work on hipDeviceMallocFinegrained memory, do the two kernels run CONCURRENTLY (each
// needs the other's granules to finish), and the round trip of one tagged 16-byte granule pushed into the peer's area with
// sc0 sc1 stores and polled with sc0 sc1 loads (what one level-2 reduction step of k_mgs_persist costs on top of the local one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); exit(2); } } while (0)
static int g_rank = 0;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_sys(char* p, v4u v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ v4u ld_sys(const char* p) {
    v4u v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// every block: spin until the local slot carries the final tag (co-residency of all G blocks is needed for that only in the
// sense that block 0 must run).  block 0 / lane 0: ping-pong.  out[0] = ticks, out[1] = 1 on timeout
__global__ __launch_bounds__(512) void k_probe(char* mine, char* peer, int rank, int rounds, long long* out, int* arrive) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) { lds[0] = 1; atomicAdd(arrive, 1); }
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // wait until all my blocks are resident (arrive == gridDim.x): co-residency check
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) {
            if (wall_clock64() - t0 > 300000000ll) { out[1] = 2; break; }
        }
        const long long t1 = wall_clock64();
        int bad = 0;
        for (int i = 1; i <= rounds && !bad; ++i) {
            v4u t; t.x = (unsigned)i; t.y = 0x3ff00000u + rank; t.z = (unsigned)i * 7u; t.w = (unsigned)i;
            st_sys(peer + 64 * rank, t);     // my granule into the peer's area, slot [rank]
            for (;;) {
                v4u r = ld_sys(mine + 64 * (1 - rank));   // the peer's granule in my area
                if (r.x == (unsigned)i && r.w == (unsigned)i) { if (r.z != (unsigned)i * 7u) bad = 3; break; }
                if (wall_clock64() - t1 > 300000000ll) { bad = 1; break; }
            }
        }
        out[0] = wall_clock64() - t1;
        out[1] = out[1] ? out[1] : bad;
        // release the other blocks
        v4u f; f.x = 0xffffffffu; f.y = f.z = 0; f.w = 0xffffffffu;
        st_sys(mine + 1024, f);
    } else if (threadIdx.x == 0) {
        for (;;) {
            v4u r = ld_sys(mine + 1024);
            if (r.x == 0xffffffffu) break;
            if (wall_clock64() - t0 > 600000000ll) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: xs_probe rank G rounds dir\n"); return 1; }
    g_rank = atoi(argv[1]);
    const int G = atoi(argv[2]), rounds = atoi(argv[3]);
    const std::string dir = argv[4];
    const int fine = argc > 5 ? atoi(argv[5]) : 1;
    CK(hipSetDevice(0));
    char* mine = nullptr;
    hipError_t e = fine ? hipExtMallocWithFlags((void**)&mine, 4096, hipDeviceMallocFinegrained) : hipMalloc((void**)&mine, 4096);
    if (e != hipSuccess) { fprintf(stderr, "rank %d: alloc (fine=%d) failed: %s\n", g_rank, fine, hipGetErrorString(e)); return 2; }
    CK(hipMemset(mine, 0, 4096));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    CK(hipIpcGetMemHandle(&h, mine));
    {
        const std::string tmp = dir + "/h" + std::to_string(g_rank) + ".tmp", fin = dir + "/h" + std::to_string(g_rank);
        FILE* f = fopen(tmp.c_str(), "wb"); fwrite(&h, sizeof(h), 1, f); fclose(f); rename(tmp.c_str(), fin.c_str());
    }
    hipIpcMemHandle_t hp;
    {
        const std::string fin = dir + "/h" + std::to_string(1 - g_rank);
        FILE* f = nullptr;
        for (int i = 0; i < 3000 && !f; ++i) { f = fopen(fin.c_str(), "rb"); if (!f) usleep(10000); }
        if (!f) { fprintf(stderr, "rank %d: peer handle never appeared\n", g_rank); return 3; }
        if (fread(&hp, sizeof(hp), 1, f) != 1) return 3;
        fclose(f);
    }
    char* peer = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
    long long* out; int* arrive;
    CK(hipMalloc(&out, 64)); CK(hipMemset(out, 0, 64));
    CK(hipMalloc(&arrive, 64)); CK(hipMemset(arrive, 0, 64));
    const size_t lds = 150 * 1024;
    CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // rendezvous: both processes ready
    { const std::string r = dir + "/r" + std::to_string(g_rank); FILE* f = fopen(r.c_str(), "wb"); fclose(f); }
    { const std::string r = dir + "/r" + std::to_string(1 - g_rank); for (int i = 0; i < 3000 && access(r.c_str(), F_OK) != 0; ++i) usleep(10000); }
    hipLaunchKernelGGL(k_probe, dim3(G), dim3(512), lds, 0, mine, peer, g_rank, rounds, out, arrive);
    CK(hipDeviceSynchronize());
    long long ho[2];
    CK(hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost));
    const char* mask = getenv("HSA_CU_MASK");
    printf("{\"rank\": %d, \"G\": %d, \"rounds\": %d, \"fine\": %d, \"cu_mask\": \"%s\", \"status\": %lld, \"us_per_round_trip\": %.3f}\n", g_rank, G, rounds, fine,
           mask ? mask : "", ho[1], ho[1] == 0 ? ho[0] / 100.0 / rounds : -1.0);
    CK(hipIpcCloseMemHandle(peer));
    return ho[1] == 0 ? 0 : 4;
}
