// gfx950 restart-time kernels: in-place basis transform (MFMA f64 tall-skinny GEMM + LDS fallback), Givens and
// Householder on basis columns, rank-1 update.
#include "kk_device.h"

// ------------------------------------------------------------------------------------------
// restart-time kernels (per restart, not per iteration)
// ------------------------------------------------------------------------------------------
// basistransform! (orthonormal.jl:291-354): V[:, 0:n] <- V[:, 0:m] * U (m x n, column-major in
// device memory).  Row-local, so it is done in place: a block stages a 64-row x m tile in LDS,
// then each thread produces outputs for (row, 4 columns at a time).
#define BT_ROWS 64
__global__ __launch_bounds__(KK_TPB) void k_basistransform(double* __restrict__ V, int64_t ld, int m, int n,
                                                           const double* __restrict__ U) {
    extern __shared__ __attribute__((aligned(16))) double tile[];  // [m][BT_ROWS + 1]
    const int tid = threadIdx.x;
    const int TS = BT_ROWS + 1;
    for (int64_t rb = (int64_t)blockIdx.x * BT_ROWS; rb < ld; rb += (int64_t)gridDim.x * BT_ROWS) {
        for (int idx = tid; idx < m * BT_ROWS; idx += KK_TPB) {
            const int i = idx / BT_ROWS, r = idx % BT_ROWS;
            tile[i * TS + r] = V[(int64_t)i * ld + rb + r];
        }
        __syncthreads();
        const int r = tid % BT_ROWS, jg = tid / BT_ROWS;  // 4 column groups
        for (int j = jg; j < n; j += 4) {
            const double* Uj = U + (int64_t)j * m;
            double a = 0;
            for (int i = 0; i < m; ++i) a = fma(tile[i * TS + r], Uj[i], a);
            V[(int64_t)j * ld + rb + r] = a;
        }
        __syncthreads();
    }
}

// basistransform! as a tall-skinny GEMM on v_mfma_f64_16x16x4_f64 (thick restart,
// eigsolve/lanczos.jl:109): out[rows, 0:n] = V[rows, 0:m] * U, in place.
//   D[i][j] += sum_k A[i][k] B[k][j];  k = 4 basis columns per MFMA, j = 16 output columns per tile.
//   Lane (i = l&15, kq = l>>4) loads 4 CONSECUTIVE rows R0+4i..R0+4i+3 of column C+kq (32 B);
//   MFMA t uses element t, i.e. its 16 "rows" are R0 + 4i + t.  The result regs of lane l for
//   fixed (tile, r) and t = 0..3 are 4 consecutive rows R0 + 4((l>>4)+4r) + t -> 32 B stores.
// A wave owns 64-row chunks: it reads all m columns of a chunk before writing the n outputs of
// the same rows, so the transform is safely in place.  U sits in LDS (B operand).
template <int NJ>
__global__ __launch_bounds__(KK_TPB) void k_basistransform_mfma(double* V, int64_t ld, int m, int n, int npad,
                                                                const double* __restrict__ U, int64_t rpb) {
    extern __shared__ __attribute__((aligned(16))) double Us[];  // [m4][npad], zero padded
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m4 = (m + 3) & ~3;
    for (int idx = tid; idx < m4 * npad; idx += KK_TPB) {
        const int k = idx / npad, j = idx % npad;
        Us[idx] = (k < m && j < n) ? U[(int64_t)j * m + k] : 0.0;
    }
    __syncthreads();
    const int i = lane & 15, kq = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t R0 = r0 + wave * 64; R0 < r1; R0 += 256) {
        v4d acc[NJ][4];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[jt][t] = v4d{0.0, 0.0, 0.0, 0.0};
        const double* vin = V + R0 + 4 * i;
        for (int C = 0; C < m4; C += 4) {
            const int col = C + kq;
            double a[4];
            if (col < m) {
                const d2 x0 = ld2s(vin + (int64_t)col * ld), x1 = ld2s(vin + (int64_t)col * ld + 2);
                a[0] = x0.x; a[1] = x0.y; a[2] = x1.x; a[3] = x1.y;
            } else {
                a[0] = a[1] = a[2] = a[3] = 0.0;
            }
            const double* urow = Us + (C + kq) * npad + i;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const double bv = urow[jt * 16];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[jt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bv, acc[jt][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int col = jt * 16 + i;
            if (col < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double* o = V + (int64_t)col * ld + R0 + 4 * (kq + 4 * r);
                    st2(o, d2{acc[jt][0][r], acc[jt][1][r]});
                    st2(o + 2, d2{acc[jt][2][r], acc[jt][3][r]});
                }
            }
        }
    }
}

// rmul!(b, G::Givens) (dense/givens.jl:20-36): (q1,q2) <- (c q1 - s q2, s q1 + c q2)
__global__ __launch_bounds__(KK_TPB) void k_givens(double* __restrict__ q1, double* __restrict__ q2, int64_t ld,
                                                   int64_t rpb, double c, double s) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 a = ld2(q1 + r), b = ld2(q2 + r), o1, o2;
        o1.x = c * a.x - s * b.x; o1.y = c * a.y - s * b.y;
        o2.x = s * a.x + c * b.x; o2.y = s * a.y + c * b.y;
        st2(q1 + r, o1); st2(q2 + r, o2);
    }
}

// rmul!(b, H::Householder) (dense/reflector.jl:143-154), row-local and fused:
//   t = sum_j V[row,j] v[j];  V[row,j] -= beta * t * v[j]
// two sweeps over the m columns of the row tile; the second sweep hits L2.
__global__ __launch_bounds__(KK_TPB) void k_householder(double* __restrict__ V, int64_t ld, int m, kk_coef hv,
                                                        double beta, int64_t rpb) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 t{0.0, 0.0};
        for (int j = 0; j < m; ++j) {
            const d2 x = ld2(V + (int64_t)j * ld + r);
            t.x = fma(x.x, hv.v[j], t.x); t.y = fma(x.y, hv.v[j], t.y);
        }
        t.x *= beta; t.y *= beta;
        for (int j = 0; j < m; ++j) {
            d2 x = ld2(V + (int64_t)j * ld + r);
            x.x = fma(-t.x, hv.v[j], x.x); x.y = fma(-t.y, hv.v[j], x.y);
            st2(V + (int64_t)j * ld + r, x);
        }
    }
}

// the same with the reflector read from device memory: more than KK_MAX_M columns (the kernarg block holds 256 coefficients)
__global__ __launch_bounds__(KK_TPB) void k_householder_dev(double* __restrict__ V, int64_t ld, int m, const double* __restrict__ hv,
                                                            double beta, int64_t rpb) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        d2 t{0.0, 0.0};
        for (int j = 0; j < m; ++j) {
            const d2 x = ld2(V + (int64_t)j * ld + r);
            const double c = hv[j];
            t.x = fma(x.x, c, t.x); t.y = fma(x.y, c, t.y);
        }
        t.x *= beta; t.y *= beta;
        for (int j = 0; j < m; ++j) {
            d2 x = ld2(V + (int64_t)j * ld + r);
            const double c = hv[j];
            x.x = fma(-t.x, c, x.x); x.y = fma(-t.y, c, x.y);
            st2(V + (int64_t)j * ld + r, x);
        }
    }
}

// rank1update! (orthonormal.jl:210-275): V_j = beta*V_j + alpha * y * x[j]
__global__ __launch_bounds__(KK_TPB) void k_rank1(double* __restrict__ V, int64_t ld, int m,
                                                  const double* __restrict__ y, kk_coef xc, double alpha, double beta,
                                                  int64_t rpb) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t r = r0 + threadIdx.x * 2; r < r1; r += KK_SUB) {
        const d2 yv = ld2(y + r);
        for (int j = 0; j < m; ++j) {
            const double a = alpha * xc.v[j];
            d2 x;
            if (beta == 0.0) { x.x = a * yv.x; x.y = a * yv.y; }
            else {
                x = ld2(V + (int64_t)j * ld + r);
                x.x = fma(a, yv.x, beta * x.x); x.y = fma(a, yv.y, beta * x.y);
            }
            st2(V + (int64_t)j * ld + r, x);
        }
    }
}

// ---- launchers
int kk_launch_basistransform(kk_ctx ctx, double* V, int64_t ld, int m, int n, const double* U_dev) {
    kk_prof_scope ps(ctx, "k_basistransform");
    const int nj = (n + 15) / 16;
    if (nj <= 6 && !getenv("KK_BASISTRANSFORM_LDS")) {
        int npad = (n + 15) / 16 * 16;
        while (npad % 32 != 16) npad += 16;   // B-operand rows land on disjoint LDS banks
        const int m4 = (m + 3) & ~3;
        const size_t shm = (size_t)m4 * npad * sizeof(double);
        kk_part p = kk_partition(ctx, ld);
        dim3 g(p.nblk), b(KK_TPB);
#define BT_CASE(NJT)                                                                                                   \
        {                                                                                                              \
            KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_basistransform_mfma<NJT>),                      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                         \
            hipLaunchKernelGGL((k_basistransform_mfma<NJT>), g, b, shm, ctx->stream, V, ld, m, n, npad, U_dev, p.rpb);  \
        }
        if (shm <= 160 * 1024 - 256) {
            switch (nj) {
                case 1: BT_CASE(1) break;
                case 2: BT_CASE(2) break;
                case 3: BT_CASE(3) break;
                case 4: BT_CASE(4) break;
                case 5: BT_CASE(5) break;
                default: BT_CASE(6) break;
            }
            KK_HIP(hipGetLastError());
            return KK_OK;
        }
#undef BT_CASE
    }
    const size_t shm = (size_t)m * (BT_ROWS + 1) * sizeof(double);
    int nb = (int)std::min<int64_t>(ld / BT_ROWS, (int64_t)ctx->num_cus * 8);
    if (nb < 1) nb = 1;
    if (shm > 64 * 1024) {
        KK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_basistransform),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    }
    hipLaunchKernelGGL(k_basistransform, dim3(nb), dim3(KK_TPB), shm, ctx->stream, V, ld, m, n, U_dev);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_givens(kk_ctx ctx, double* q1, double* q2, int64_t ld, double c, double s) {
    kk_prof_scope ps(ctx, "k_givens");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_givens, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, q1, q2, ld, p.rpb, c, s);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_householder(kk_ctx ctx, double* V, int64_t ld, int m, const kk_coef* v, double beta) {
    kk_prof_scope ps(ctx, "k_householder");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_householder, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, V, ld, m, *v, beta, p.rpb);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_householder_dev(kk_ctx ctx, double* V, int64_t ld, int m, const double* v_dev, double beta) {
    kk_prof_scope ps(ctx, "k_householder");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_householder_dev, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, V, ld, m, v_dev, beta, p.rpb);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

int kk_launch_rank1(kk_ctx ctx, double* V, int64_t ld, int m, const double* y, const kk_coef* x, double alpha,
                    double beta) {
    kk_prof_scope ps(ctx, "k_rank1");
    kk_part p = kk_partition(ctx, ld);
    hipLaunchKernelGGL(k_rank1, dim3(p.nblk), dim3(KK_TPB), 0, ctx->stream, V, ld, m, y, *x, alpha, beta, p.rpb);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

