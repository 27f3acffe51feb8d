// gfx950 kernels of the short-recurrence solvers (CG, BiCGStab, LSMR): fused vector updates whose scalars live in the
// context's device workspace.  Same streaming design as kk_kernels_stream.hip.
#include "kk_device.h"

// CG update fused (linsolve/cg.jl:63-66): x += alpha p ; r -= alpha q ; partial |r|^2
__global__ __launch_bounds__(KK_TPB) void k_cg_update(double* __restrict__ x, const double* __restrict__ p, double* __restrict__ r,
                                                      const double* __restrict__ q, int64_t ld, int64_t rpb, double alpha,
                                                      const double* __restrict__ pq_dev, double* __restrict__ part) {
    __shared__ double sm[4];
    if (pq_dev) alpha = alpha / *pq_dev;   // alpha = rho / <p, q> with the inner product still on the device
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double acc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 xv = ld2(x + i), pv = ld2(p + i), rv = ld2(r + i), qv = ld2(q + i);
        xv.x = fma(alpha, pv.x, xv.x); xv.y = fma(alpha, pv.y, xv.y);
        rv.x = fma(-alpha, qv.x, rv.x); rv.y = fma(-alpha, qv.y, rv.y);
        st2(x + i, xv); st2(r + i, rv);
        acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
    }
    double t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---- BiCGStab (linsolve/bicgstab.jl:118-199) as three fused vector kernels; every scalar of the recurrence
// stays in the context's device scalars sc[] = {rho, rho_old, sigma, alpha, omega, <t,s>, <t,t>}
enum { BI_RHO = 0, BI_RHO_OLD = 1, BI_SIGMA = 2, BI_ALPHA = 3, BI_OMEGA = 4, BI_TS = 5, BI_TT = 6 /* triple 6..8 */,
       BI_ALPHA_OLD = 15 /* alpha of the last completed full step: a run-ahead half overwrites BI_ALPHA */ };
__global__ void k_set_scalar(double* dst, double v) { *dst = v; }
// p_out = r + beta (p - omega v),  beta = (rho/rho_old)(alpha/omega)          (:121-125)
__global__ __launch_bounds__(KK_TPB) void k_bicg_p(double* __restrict__ p_out, const double* __restrict__ p,
                                                   const double* __restrict__ r, const double* __restrict__ v, int64_t ld,
                                                   int64_t rpb, const double* __restrict__ sc) {
    const double omega = sc[BI_OMEGA];
    const double beta = (sc[BI_RHO] / sc[BI_RHO_OLD]) * (sc[BI_ALPHA_OLD] / omega);
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 pv = ld2(p + i), rv = ld2(r + i), vv = ld2(v + i);
        pv.x = fma(-omega, vv.x, pv.x); pv.y = fma(-omega, vv.y, pv.y);
        pv.x = fma(beta, pv.x, rv.x); pv.y = fma(beta, pv.y, rv.y);
        st2(p_out + i, pv);
    }
}
// alpha = rho/sigma ; s = r - alpha v ; partial |s|^2                          (:130-139)
__global__ __launch_bounds__(KK_TPB) void k_bicg_s(double* __restrict__ s, const double* __restrict__ r,
                                                   const double* __restrict__ v, int64_t ld, int64_t rpb,
                                                   double* __restrict__ sc, double* __restrict__ part) {
    __shared__ double sm[4];
    const double alpha = sc[BI_RHO] / sc[BI_SIGMA];
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[BI_ALPHA] = alpha;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double acc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 rv = ld2(r + i), vv = ld2(v + i);
        rv.x = fma(-alpha, vv.x, rv.x); rv.y = fma(-alpha, vv.y, rv.y);
        st2(s + i, rv);
        acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
    }
    double t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// omega = <t,s>/<t,t> ; x += alpha p + omega s ; r = s - omega t ; partials |r|^2 and <r_shadow, r>   (:160-169,120)
__global__ __launch_bounds__(KK_TPB) void k_bicg_xr(double* __restrict__ x, const double* __restrict__ p,
                                                    const double* __restrict__ s, const double* __restrict__ t,
                                                    double* __restrict__ r, const double* __restrict__ rs, int64_t ld,
                                                    int64_t rpb, double* __restrict__ sc, double* __restrict__ part_n,
                                                    double* __restrict__ part_d) {
    __shared__ double sm[4];
    const double alpha = sc[BI_ALPHA];
    const double omega = sc[BI_TS] / sc[BI_TT];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[BI_OMEGA] = omega;
        sc[BI_ALPHA_OLD] = alpha;
        sc[BI_RHO_OLD] = sc[BI_RHO];   // the finalize of <r_shadow, r> (next kernel in the stream) overwrites BI_RHO
    }
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double nacc = 0, dacc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 xv = ld2(x + i), pv = ld2(p + i), sv = ld2(s + i), tv = ld2(t + i), zv = ld2(rs + i);
        xv.x = fma(alpha, pv.x, xv.x); xv.y = fma(alpha, pv.y, xv.y);
        xv.x = fma(omega, sv.x, xv.x); xv.y = fma(omega, sv.y, xv.y);
        sv.x = fma(-omega, tv.x, sv.x); sv.y = fma(-omega, tv.y, sv.y);
        st2(x + i, xv); st2(r + i, sv);
        nacc = fma(sv.x, sv.x, nacc); nacc = fma(sv.y, sv.y, nacc);
        dacc = fma(zv.x, sv.x, dacc); dacc = fma(zv.y, sv.y, dacc);
    }
    double a = block_sum(nacc, sm);
    if (threadIdx.x == 0) part_n[blockIdx.x] = a;
    __syncthreads();
    double b = block_sum(dacc, sm);
    if (threadIdx.x == 0) part_d[blockIdx.x] = b;
}

// ---- LSMR (lssolve/lsmr.jl:61-110) vector updates, fused
// Ah = Av - c Ah ; u = Av - alpha u ; partial |u|^2                              (:64-68)
__global__ __launch_bounds__(KK_TPB) void k_lsmr_u(const double* __restrict__ av, double* __restrict__ ah,
                                                   double* __restrict__ u, int64_t ld, int64_t rpb, double c, double alpha,
                                                   double* __restrict__ part) {
    __shared__ double sm[4];
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    double acc = 0;
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        const d2 a = ld2(av + i);
        d2 h = ld2(ah + i), uv = ld2(u + i);
        h.x = fma(-c, h.x, a.x); h.y = fma(-c, h.y, a.y);
        uv.x = fma(-alpha, uv.x, a.x); uv.y = fma(-alpha, uv.y, a.y);
        st2(ah + i, h); st2(u + i, uv);
        acc = fma(uv.x, uv.x, acc); acc = fma(uv.y, uv.y, acc);
    }
    double t = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// hbar = h - c1 hbar ; x += c2 hbar ; [h = v - c3 h when v != nullptr]           (:121-128)
__global__ __launch_bounds__(KK_TPB) void k_lsmr_hx(double* __restrict__ h, double* __restrict__ hbar, double* __restrict__ x,
                                                    const double* __restrict__ v, int64_t ld, int64_t rpb, double c1,
                                                    double c2, double c3) {
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = imin(r0 + rpb, ld);
    for (int64_t i = r0 + threadIdx.x * 2; i < r1; i += KK_SUB) {
        d2 hv = ld2(h + i), hb = ld2(hbar + i), xv = ld2(x + i);
        hb.x = fma(-c1, hb.x, hv.x); hb.y = fma(-c1, hb.y, hv.y);
        xv.x = fma(c2, hb.x, xv.x); xv.y = fma(c2, hb.y, xv.y);
        st2(hbar + i, hb); st2(x + i, xv);
        if (v) {
            const d2 vv = ld2(v + i);
            hv.x = fma(-c3, hv.x, vv.x); hv.y = fma(-c3, hv.y, vv.y);
            st2(h + i, hv);
        }
    }
}

// ---- launchers
int kk_launch_cg_update(kk_ctx ctx, double* x, const double* p, double* r, const double* q, int64_t ld, double alpha,
                        const double* pq_dev, double* nrm_out3) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_cg_update");
        hipLaunchKernelGGL(k_cg_update, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, x, p, r, q, ld, pt.rpb, alpha, pq_dev,
                           part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true);
}

int kk_launch_bicg_p(kk_ctx ctx, double* p_out, const double* p, const double* r, const double* v, int64_t ld,
                     const double* sc) {
    kk_part pt = kk_partition(ctx, ld);
    kk_prof_scope ps(ctx, "k_bicg_p");
    hipLaunchKernelGGL(k_bicg_p, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, p_out, p, r, v, ld, pt.rpb, sc);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_set_scalar(kk_ctx ctx, double* dst, double v) {
    hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, ctx->stream, dst, v);
    KK_HIP(hipGetLastError());
    return KK_OK;
}
int kk_launch_bicg_s(kk_ctx ctx, double* s, const double* r, const double* v, int64_t ld, double* sc, double* nrm_out3) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_bicg_s");
        hipLaunchKernelGGL(k_bicg_s, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, s, r, v, ld, pt.rpb, sc,
                           part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true);
}
int kk_launch_bicg_xr(kk_ctx ctx, double* x, const double* p, const double* s, const double* t, double* r,
                      const double* rs, int64_t ld, double* sc, double* nrm_out3, double* rho_out) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_bicg_xr");
        hipLaunchKernelGGL(k_bicg_xr, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, x, p, s, t, r, rs, ld, pt.rpb, sc,
                           part_row(ctx, PART_SCAL_A), part_row(ctx, PART_SCAL_B));
    }
    KK_HIP(hipGetLastError());
    KK_TRY(finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true));
    return finalize_scalar(ctx, PART_SCAL_B, pt.nblk, rho_out, false);
}

int kk_launch_lsmr_u(kk_ctx ctx, const double* av, double* ah, double* u, int64_t ld, double c, double alpha,
                     double* nrm_out3) {
    kk_part pt = kk_partition(ctx, ld);
    {
        kk_prof_scope ps(ctx, "k_lsmr_u");
        hipLaunchKernelGGL(k_lsmr_u, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, av, ah, u, ld, pt.rpb, c, alpha,
                           part_row(ctx, PART_SCAL_A));
    }
    KK_HIP(hipGetLastError());
    return finalize_scalar(ctx, PART_SCAL_A, pt.nblk, nrm_out3, true);
}
int kk_launch_lsmr_hx(kk_ctx ctx, double* h, double* hbar, double* x, const double* v, int64_t ld, double c1, double c2,
                      double c3) {
    kk_part pt = kk_partition(ctx, ld);
    kk_prof_scope ps(ctx, "k_lsmr_hx");
    hipLaunchKernelGGL(k_lsmr_hx, dim3(pt.nblk), dim3(KK_TPB), 0, ctx->stream, h, hbar, x, v, ld, pt.rpb, c1, c2, c3);
    KK_HIP(hipGetLastError());
    return KK_OK;
}

