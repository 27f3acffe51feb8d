// Exploration tool (not part of the product): the rate at which gfx950 serves RANDOM 8-byte gathers, as a function of the
// table size (L1 / L2 / Infinity Cache / HBM resident) and of the gathers in flight per lane.  It is the ceiling of every
// sparse apply whose column indices are random (config 4: 5M x 1M, 20 nnz/row): such a kernel streams (index, value)
// pairs and issues ONE gather per nonzero, nothing else.  The kernel below does exactly that and no more -- coalesced
// int32 index stream (+ optionally the 8-byte value stream), gather, FMA.
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o tools/bin/gather_rate ; run: tools/bin/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

template <int U, bool VALS>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ idx, const double* __restrict__ val, size_t n,
                                                const double* __restrict__ x, double* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (size_t)(U - 1) * stride < n; i += stride * U) {
        int c[U];
        double v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) c[k] = __builtin_nontemporal_load(idx + i + (size_t)k * stride);
        if (VALS) {
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(val + i + (size_t)k * stride);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) acc = fma(VALS ? v[k] : 1.0, x[c[k]], acc);
    }
    if (acc == 12345.678) out[0] = acc;
}

template <int U, bool VALS>
float run(const int* idx, const double* val, size_t n, const double* x, double* out, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_gather<U, VALS><<<grid, 256>>>(idx, val, n, x, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        k_gather<U, VALS><<<grid, 256>>>(idx, val, n, x, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

// fill idx with table-bounded pseudo-random columns on the device (--json mode: no 134M-entry host loop)
__global__ void k_fill_idx(int* __restrict__ idx, size_t n, unsigned table, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;   // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        idx[i] = (int)(z % table);
    }
}
// `gather_rate --json <table doubles>`: ONE table size -- the column tile of the library's tiled SELL format (393216 doubles = 3 MB,
// L2-resident) -- with the value stream, 8 / 16 gathers in flight, two grids; one JSON line with the best rate.  bench.py runs it at the
// start of a GKL measurement: `gather_ceiling_Ggathers_per_s` next to the leg's HBM fraction.
static int json_mode(size_t table) {
    const size_t n = (size_t)1 << 26;
    int* idx; double *val, *x, *out;
    if (hipMalloc(&idx, n * 4) != hipSuccess || hipMalloc(&val, n * 8) != hipSuccess || hipMalloc(&out, 8) != hipSuccess || hipMalloc(&x, table * 8) != hipSuccess) return 1;
    hipMemset(val, 0, n * 8); hipMemset(x, 0, table * 8);
    k_fill_idx<<<4096, 256>>>(idx, n, (unsigned)table, 7ull);
    hipDeviceSynchronize();
    float best = 1e30f; int bu = 0, bg = 0;
    for (int grid : {2048, 8192}) {
        const float m8 = run<8, true>(idx, val, n, x, out, grid), m16 = run<16, true>(idx, val, n, x, out, grid);
        if (m8 < best) { best = m8; bu = 8; bg = grid; }
        if (m16 < best) { best = m16; bu = 16; bg = grid; }
    }
    printf("{\"tool\": \"gather_rate\", \"table_bytes\": %zu, \"gathers_per_launch\": %zu, \"value_stream\": true, \"in_flight\": %d, \"grid\": %d, \"ms\": %.4f, "
           "\"Ggathers_per_s\": %.1f, \"stream_GBps\": %.0f}\n", table * 8, n, bu, bg, best, n / best / 1e6, n * 12.0 / best / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && std::string(argv[1]) == "--json") return json_mode(argc >= 3 ? (size_t)atoll(argv[2]) : (size_t)393216);
    const size_t n = (size_t)1 << 27;   // 134M gathers per launch (config 4 has 1e8 nonzeros per apply)
    int* idx; double *val, *x, *out;
    hipMalloc(&idx, n * 4); hipMalloc(&val, n * 8); hipMalloc(&out, 8);
    hipMemset(val, 0, n * 8);
    const size_t max_table = (size_t)1 << 26;   // doubles (512 MB)
    hipMalloc(&x, max_table * 8);
    hipMemset(x, 0, max_table * 8);
    std::vector<int> h(n);
    std::mt19937_64 rng(7);
    printf("{\"gathers_per_launch\": %zu, \"rows\": [\n", n);
    bool first = true;
    for (size_t table : {(size_t)4096, (size_t)1 << 16, (size_t)393216, (size_t)1 << 20, (size_t)5 << 20, (size_t)1 << 25}) {
        for (size_t i = 0; i < n; ++i) h[i] = (int)(rng() % table);
        hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int grid : {2048, 8192}) {
            struct { int u; bool vals; float ms; } res[] = {
                {1, false, run<1, false>(idx, val, n, x, out, grid)}, {4, false, run<4, false>(idx, val, n, x, out, grid)},
                {8, false, run<8, false>(idx, val, n, x, out, grid)}, {16, false, run<16, false>(idx, val, n, x, out, grid)},
                {4, true, run<4, true>(idx, val, n, x, out, grid)},   {8, true, run<8, true>(idx, val, n, x, out, grid)},
                {16, true, run<16, true>(idx, val, n, x, out, grid)}};
            for (auto& r : res) {
                printf("%s  {\"table_bytes\": %zu, \"grid\": %d, \"in_flight\": %d, \"value_stream\": %s, \"ms\": %.4f, \"Ggathers_per_s\": %.1f, \"stream_GBps\": %.0f}",
                       first ? "" : ",\n", table * 8, grid, r.u, r.vals ? "true" : "false", r.ms, n / r.ms / 1e6,
                       n * (r.vals ? 12.0 : 4.0) / r.ms / 1e6);
                first = false;
            }
        }
    }
    printf("\n]}\n");
    return 0;
}
