// SURVEY.md 8(f)-4 experiment: what would fp32 STORAGE of the Krylov basis (f64 arithmetic) buy on the two basis-streaming
// passes of an expand!?  Same simple tiling for both precisions (256-thread blocks, contiguous row ranges, 16-byte
// non-temporal loads, 4 columns in flight), so the ratio is apples to apples:
//   project  : s_j = sum_r V[r][j] * w[r]          (V once, w once)
//   unproject: w[r] -= sum_j V[r][j] * c[j]        (V once, w read + written)
// build: hipcc --offload-arch=gfx950 -O3 tools/fp32_basis.hip -o tools/bin/fp32_basis ; run: tools/bin/fp32_basis [rows]   (m = 52, the mean basis size of the headline sweep, and m = 100)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T> struct row4;   // 4 consecutive rows of one column as doubles
template <> struct row4<double> {
    static __device__ __forceinline__ void load(const double* p, double* o) {
        const d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + 2));
        o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
    }
};
template <> struct row4<float> {
    static __device__ __forceinline__ void load(const float* p, double* o) {
        const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    }
};
template <typename T, int M>
__global__ __launch_bounds__(256) void k_project(const T* __restrict__ V, long ld, const double* __restrict__ w, long rpb,
                                                 double* __restrict__ s) {
    // rows outer (the 4-row piece of w stays in registers), all M columns inner with one register accumulator per column
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < ld ? r0 + rpb : ld;
    double acc[M];
#pragma unroll
    for (int j = 0; j < M; ++j) acc[j] = 0;
    for (long r = r0 + threadIdx.x * 4; r < r1; r += 1024) {
        double wv[4];
        row4<double>::load(w + r, wv);
#pragma unroll
        for (int c = 0; c < M; c += 4) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) row4<T>::load(V + (long)(c + u) * ld + r, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[c + u] = fma(v[u][i], wv[i], acc[c + u]);
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        double t = acc[j];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(s + j, t);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_unproject(const T* __restrict__ V, long ld, int m, double* __restrict__ w, long rpb,
                                                   const double* __restrict__ cf) {
    const long r0 = (long)blockIdx.x * rpb, r1 = r0 + rpb < ld ? r0 + rpb : ld;
    for (long r = r0 + threadIdx.x * 4; r < r1; r += 1024) {
        double acc[4];
        row4<double>::load(w + r, acc);
        for (int c = 0; c + 4 <= m; c += 4) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) row4<T>::load(V + (long)(c + u) * ld + r, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double cc = cf[c + u];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fma(-cc, v[u][i], acc[i]);
            }
        }
        *reinterpret_cast<d2*>(w + r) = d2{acc[0], acc[1]};
        *reinterpret_cast<d2*>(w + r + 2) = d2{acc[2], acc[3]};
    }
}
template <typename T, int M>
void run(const char* name, long ld) {
    const int m = M;
    T* V; double *w, *s, *c;
    hipMalloc(&V, (size_t)m * ld * sizeof(T)); hipMalloc(&w, ld * 8); hipMalloc(&s, 1024 * 8); hipMalloc(&c, 1024 * 8);
    hipMemset(V, 0, (size_t)m * ld * sizeof(T)); hipMemset(w, 0, ld * 8); hipMemset(s, 0, 1024 * 8); hipMemset(c, 0, 1024 * 8);
    const long nsub = ld / 1024, target = 256L * 4;
    const long spb = (nsub + target - 1) / target;
    const int nblk = (int)((nsub + spb - 1) / spb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float bp = 1e30f, bu = 1e30f;
    for (int r = 0; r < 6; ++r) {
        float ms;
        hipEventRecord(a); k_project<T, M><<<nblk, 256>>>(V, ld, w, spb * 1024, s); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); if (r && ms < bp) bp = ms;
        hipEventRecord(a); k_unproject<T><<<nblk, 256>>>(V, ld, m, w, spb * 1024, c); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); if (r && ms < bu) bu = ms;
    }
    printf("%s basis, m=%d, %ld rows: project %.3f ms (%.2f TB/s of basis bytes)  unproject %.3f ms (%.2f TB/s)\n", name, m, ld, bp,
           (double)m * ld * sizeof(T) / bp / 1e9, bu, ((double)m * ld * sizeof(T) + 16.0 * ld) / bu / 1e9);
    hipFree(V); hipFree(w); hipFree(s); hipFree(c);
}
int main(int argc, char** argv) {
    const long ld = argc > 1 ? atol(argv[1]) : 10000896;
    run<double, 52>("f64", ld);
    run<float, 52>("f32", ld);
    run<double, 100>("f64", ld);
    run<float, 100>("f32", ld);
    return 0;
}
