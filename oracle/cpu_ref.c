/*
 * cpu_ref.c -- C twin of oracle/krylov_oracle.py for the expand! paths of Lanczos (config 2), Arnoldi / GMRES (config 3),
 * GKL (config 4) and BlockLanczos (config 5).
 *
 * TEST INFRASTRUCTURE ONLY (tests/ and the cpu_baseline leg of bench.py).  It restates the
 * reference's CPU path *as the reference issues it*: one separately allocated vector per basis
 * element (OrthonormalBasis{T} = Vector{T}, src/orthonormal.jl:26-28), one BLAS-1 style pass
 * per inner/add!!/norm call (no fusion), and Julia's column-oriented, single-threaded
 * SparseMatrixCSC * vector product with Int64 indices (stdlib SparseArrays, reached through
 * src/apply.jl:1).  BLAS-1 loops are OpenMP-parallel, standing in for the threaded OpenBLAS
 * that LinearAlgebra.dot/axpy!/norm reach through libblastrampoline.
 *
 * Parity status: bit-level unpinned (see oracle/krylov_oracle.py header); this file is
 * validated against the NumPy oracle in tests/test_oracle_cpu_ref.py.
 *
 * Orthogonalizer codes match include/krylov_hip.h: 0 CGS, 1 MGS, 2 CGS2, 3 MGS2, 4 CGSIR, 5 MGSIR.
 */
#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* inner(x,y).  Deterministic whatever the number of threads and their scheduling: KK_DOT_CHUNKS fixed chunks, each summed
 * front to back by one thread, the chunk sums added in chunk order -- the twin gives the same bits on every box, so the
 * parity figures of bench.py / the full-size tests do not depend on how OpenMP combined a reduction that day. */
#define KK_DOT_CHUNKS 1024
static double ddot(int64_t n, const double* x, const double* y) {
    double part[KK_DOT_CHUNKS];
    const int64_t len = (n + KK_DOT_CHUNKS - 1) / KK_DOT_CHUNKS;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < KK_DOT_CHUNKS; ++c) {
        const int64_t lo = (int64_t)c * len, hi = lo + len < n ? lo + len : n;
        double s = 0;
#pragma omp simd reduction(+ : s)
        for (int64_t i = lo; i < hi; ++i) s += x[i] * y[i];
        part[c] = s;
    }
    double s = 0;
    for (int c = 0; c < KK_DOT_CHUNKS; ++c) s += part[c];
    return s;
}
static double dnrm2(int64_t n, const double* x) { return sqrt(ddot(n, x, x)); } /* norm(x) */
static void daxpy(int64_t n, double a, const double* x, double* y) {               /* add!!(y,x,a) */
#pragma omp parallel for simd schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
}
static void dscal(int64_t n, double a, double* x) { /* scale!!(x,a) */
#pragma omp parallel for simd schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] *= a;
}

/* y = A*x, A in CSC with 1-based Int64 indices (Julia's layout); serial scatter as in SparseArrays */
static void csc_mul(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                    const double* x, double* y) {
    memset(y, 0, (size_t)nrows * sizeof(double));
    for (int64_t j = 0; j < ncols; ++j) {
        const double xj = x[j];
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) y[rowval[p] - 1] += nzval[p] * xj;
    }
}

/* orthogonalize!!(w, V[0..m), x, CGS) -- src/orthonormal.jl:378-384 (generic, un-threaded-kernel path) */
static void cgs_pass(int64_t n, int m, double** V, double* w, double* x) {
    for (int j = 0; j < m; ++j) x[j] = ddot(n, V[j], w);       /* project!!   :107-115 */
    for (int j = 0; j < m; ++j) daxpy(n, -x[j], V[j], w);      /* unproject!! :146-148 */
}
/* MGS sweep -- src/orthonormal.jl:414-423 ; returns last coefficient */
static double mgs_pass(int64_t n, int m, double** V, double* w) {
    double s = 0;
    for (int j = 0; j < m; ++j) {
        s = ddot(n, V[j], w);
        daxpy(n, -s, V[j], w);
    }
    return s;
}

/* lanczosrecurrence (src/factorizations/lanczos.jl:295-376). V has m vectors (after the push),
 * w is a fresh vector receiving A*V[m-1]. Returns alpha, beta; *passes counts full passes. */
static void lanczos_recurrence(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, int m,
                               double** V, double* w, double beta_old, int orth, double eta, double* alpha_out,
                               double* beta_out, int* passes, double* scratch) {
    const double* v = V[m - 1];
    double alpha, beta, s;
    csc_mul(n, n, colptr, rowval, nzval, v, w);
    const int cgs_order = (orth == 0 || orth == 2 || orth == 4);
    if (cgs_order) {
        alpha = ddot(n, v, w);
        daxpy(n, -beta_old, V[m - 2], w);
        daxpy(n, -alpha, v, w);
    } else {
        daxpy(n, -beta_old, V[m - 2], w);
        alpha = ddot(n, v, w);
        daxpy(n, -alpha, v, w);
    }
    switch (orth) {
        case 0:
        case 1:
            beta = dnrm2(n, w);
            break;
        case 2: /* :320-322 */
            cgs_pass(n, m, V, w, scratch);
            alpha += scratch[m - 1];
            beta = dnrm2(n, w);
            ++*passes;
            break;
        case 3: /* :331-336 */
            s = mgs_pass(n, m, V, w);
            alpha += s;
            beta = dnrm2(n, w);
            ++*passes;
            break;
        default: { /* IR :346-354 / :363-374 */
            const double ab2 = alpha * alpha + beta_old * beta_old;
            beta = dnrm2(n, w);
            double nold = sqrt(beta * beta + ab2);
            while (DBL_EPSILON < beta && beta < eta * nold) {
                nold = beta;
                if (orth == 4) {
                    cgs_pass(n, m, V, w, scratch);
                    alpha += scratch[m - 1];
                } else {
                    alpha += mgs_pass(n, m, V, w);
                }
                beta = dnrm2(n, w);
                ++*passes;
            }
        }
    }
    *alpha_out = alpha;
    *beta_out = beta;
}

/*
 * initialize (lanczos.jl:180-222) + `steps` expand! calls (lanczos.jl:250-272).
 * alphas/betas must hold steps+1 doubles.  If basis_out != NULL it receives the steps+1 basis
 * vectors followed by the residual, column-major (n x (steps+2)).
 * Returns 0, or -1 on allocation failure / zero start vector.
 */
int kkref_lanczos(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* x0,
                  int steps, int orth, double eta, int nthreads, double* alphas, double* betas, int* total_passes,
                  double* basis_out) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    double** V = (double**)calloc((size_t)steps + 2, sizeof(double*));
    double* scratch = (double*)malloc(((size_t)steps + 2) * sizeof(double));
    if (!V || !scratch) return -1;
    int passes = 0, rc = 0;
    /* initialize */
    double* v = (double*)malloc((size_t)n * sizeof(double));
    double* r = (double*)malloc((size_t)n * sizeof(double));
    if (!v || !r) return -1;
    const double beta0 = dnrm2(n, x0);
    if (beta0 == 0) return -1;
    csc_mul(n, n, colptr, rowval, nzval, x0, r);
    double alpha = ddot(n, x0, r) / (beta0 * beta0);
    memcpy(v, x0, (size_t)n * sizeof(double));
    dscal(n, 1.0 / beta0, v);
    dscal(n, 1.0 / beta0, r);
    double beta_old = dnrm2(n, r);
    daxpy(n, -alpha, v, r);
    double beta = dnrm2(n, r);
    if (orth == 2 || orth == 3) {
        double da = ddot(n, v, r);
        alpha += da;
        daxpy(n, -da, v, r);
        beta = dnrm2(n, r);
    } else if (orth >= 4) {
        while (DBL_EPSILON < beta && beta < eta * beta_old) {
            beta_old = beta;
            double da = ddot(n, v, r);
            alpha += da;
            daxpy(n, -da, v, r);
            beta = dnrm2(n, r);
        }
    }
    V[0] = v;
    alphas[0] = alpha;
    betas[0] = beta;
    int k = 1;
    for (int it = 0; it < steps; ++it) {
        beta_old = betas[k - 1];
        dscal(n, 1.0 / beta_old, r); /* V = push!(V, scale!!(r, 1/beta_old))  :257 */
        V[k] = r;
        double* w = (double*)malloc((size_t)n * sizeof(double)); /* A*v allocates its result */
        if (!w) { rc = -1; break; }
        lanczos_recurrence(n, colptr, rowval, nzval, k + 1, V, w, beta_old, orth, eta, &alphas[k], &betas[k], &passes,
                           scratch);
        r = w;
        ++k;
    }
    if (basis_out && rc == 0) {
        for (int j = 0; j < k; ++j) memcpy(basis_out + (size_t)j * n, V[j], (size_t)n * sizeof(double));
        memcpy(basis_out + (size_t)k * n, r, (size_t)n * sizeof(double));
    }
    if (total_passes) *total_passes = passes;
    for (int j = 0; j < k; ++j) free(V[j]);
    free(r);
    free(V);
    free(scratch);
    return rc;
}


/* ------------------------------------------------------------------------------------------------
 * Arnoldi / restarted GMRES  (config 3)
 * ------------------------------------------------------------------------------------------------ */
/* orthogonalize!!(w, V[0..m), x, alg) with coefficient accumulation -- src/orthonormal.jl:378-452 */
static void orth_full(int64_t n, int m, double** V, double* w, double* x, int orth, double eta, double* tmp, int* passes) {
    int j;
    switch (orth) {
        case 0: /* CGS :378-384 */
            cgs_pass(n, m, V, w, x);
            ++*passes;
            break;
        case 2: /* CGS2 :394-399: orthogonalize, then reorthogonalize!! (:385-393, x .+= s) */
            cgs_pass(n, m, V, w, x);
            cgs_pass(n, m, V, w, tmp);
            for (j = 0; j < m; ++j) x[j] += tmp[j];
            *passes += 2;
            break;
        case 4: { /* CGSIR :400-412 */
            double nold = dnrm2(n, w);
            cgs_pass(n, m, V, w, x);
            double nnew = dnrm2(n, w);
            ++*passes;
            while (DBL_EPSILON < nnew && nnew < eta * nold) {
                nold = nnew;
                cgs_pass(n, m, V, w, tmp);
                for (j = 0; j < m; ++j) x[j] += tmp[j];
                nnew = dnrm2(n, w);
                ++*passes;
            }
        } break;
        case 1: /* MGS :414-423 */
            for (j = 0; j < m; ++j) { x[j] = ddot(n, V[j], w); daxpy(n, -x[j], V[j], w); }
            ++*passes;
            break;
        case 3: /* MGS2 :434-439 (second sweep accumulates, :424-433) */
            for (j = 0; j < m; ++j) { x[j] = ddot(n, V[j], w); daxpy(n, -x[j], V[j], w); }
            for (j = 0; j < m; ++j) { double s = ddot(n, V[j], w); daxpy(n, -s, V[j], w); x[j] += s; }
            *passes += 2;
            break;
        default: { /* MGSIR :440-452 */
            double nold = dnrm2(n, w);
            for (j = 0; j < m; ++j) { x[j] = ddot(n, V[j], w); daxpy(n, -x[j], V[j], w); }
            double nnew = dnrm2(n, w);
            ++*passes;
            while (DBL_EPSILON < nnew && nnew < eta * nold) {
                nold = nnew;
                for (j = 0; j < m; ++j) { double s = ddot(n, V[j], w); daxpy(n, -s, V[j], w); x[j] += s; }
                nnew = dnrm2(n, w);
                ++*passes;
            }
        }
    }
}
/* _orthogonalize!!(v, q, alg) -- src/orthonormal.jl:455-473 (non-IR variants); returns the coefficient */
static double orth_vec(int64_t n, const double* q, double* v, int orth, double eta) {
    double s = ddot(n, q, v);
    daxpy(n, -s, q, v);
    if (orth == 2 || orth == 3) { /* :465-473 */
        double ds = ddot(n, q, v);
        daxpy(n, -ds, q, v);
        s += ds;
    }
    return s;
}
static double orth_vec_ir(int64_t n, const double* q, double* v, double eta) { /* :474-489 */
    double nold = dnrm2(n, v);
    double s = ddot(n, q, v);
    daxpy(n, -s, q, v);
    double nnew = dnrm2(n, v);
    while (DBL_EPSILON < nnew && nnew < eta * nold) {
        nold = nnew;
        double ds = ddot(n, q, v);
        daxpy(n, -ds, q, v);
        s += ds;
        nnew = dnrm2(n, v);
    }
    return s;
}
/* LinearAlgebra.givensAlgorithm, real case (LAPACK dlartg semantics): [c s; -s c][f; g] = [r; 0] */
static void givens_fg(double f, double g, double* c, double* s, double* r) {
    if (g == 0) { *c = 1; *s = 0; *r = f; return; }
    if (f == 0) { *c = 0; *s = 1; *r = g; return; }
    double rr = hypot(f, g);
    double cc = f / rr, ss = g / rr;
    if (fabs(f) > fabs(g) && cc < 0) { cc = -cc; ss = -ss; rr = -rr; }
    *c = cc; *s = ss; *r = rr;
}
/* packed Hessenberg H[i,j], 1-based, i <= j+1  (dense/packedhessenberg.jl:32-39) */
static inline size_t hidx(int i, int j) { return (size_t)(((j * j + j - 2) >> 1) + i - 1); }

/*
 * linsolve(A, b, x0, GMRES(orth; krylovdim, maxiter, tol), a0, a1)  -- src/linsolve/gmres.jl:1-151 on the Arnoldi
 * factorization of src/factorizations/arnoldi.jl:135-245.  A in CSC with 1-based Int64 indices.  x receives the
 * solution (x0 on entry may be NULL = zero).  info = {converged, numiter, numops}; *normres = final residual norm.
 * trace (optional, trace_cap doubles) receives the residual estimate beta after every inner step (gmres.jl:53,94),
 * *trace_len their number.  Returns 0, -1 on allocation failure.
 */
int kkref_gmres(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b,
                const double* x0, double a0, double a1, int krylovdim, int maxiter, double tol, int orth, double eta,
                int nthreads, double* x, int* info, double* normres, double* trace, int trace_cap, int* trace_len) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const int K = krylovdim;
    double** V = (double**)calloc((size_t)K + 2, sizeof(double*));
    double* H = (double*)calloc((size_t)(K + 2) * (K + 3) / 2 + 8, sizeof(double));
    double* R = (double*)calloc((size_t)K * K, sizeof(double)); /* column-major K x K */
    double* y = (double*)calloc((size_t)K + 2, sizeof(double));
    double* gc = (double*)calloc((size_t)K + 1, sizeof(double));
    double* gsn = (double*)calloc((size_t)K + 1, sizeof(double));
    int* g1 = (int*)calloc((size_t)K + 1, sizeof(int));
    int* g2 = (int*)calloc((size_t)K + 1, sizeof(int));
    double* hx = (double*)calloc((size_t)K + 2, sizeof(double));
    double* tmp = (double*)calloc((size_t)K + 2, sizeof(double));
    double* r = (double*)malloc((size_t)n * sizeof(double));
    double* t = (double*)malloc((size_t)n * sizeof(double));
    if (!V || !H || !R || !y || !gc || !gsn || !g1 || !g2 || !hx || !tmp || !r || !t) return -1;
    int passes = 0, ntrace = 0, rc = 0;
    int64_t i;
    /* r = b - a0 x0 - a1 A x0 ; x = x0   :3-12 */
    if (x0) memcpy(x, x0, (size_t)n * sizeof(double)); else memset(x, 0, (size_t)n * sizeof(double));
    csc_mul(n, n, colptr, rowval, nzval, x, t);
    memcpy(r, b, (size_t)n * sizeof(double));
    if (a0 != 0) daxpy(n, -a0, x, r);
    daxpy(n, -a1, t, r);
    double beta = dnrm2(n, r);
    int numiter = 0, numops = 1, converged = 0, k = 0, nv = 0;
    if (beta < tol) { converged = 1; goto done; } /* :21-27 */
    /* fact = initialize(ArnoldiIterator(A, r, orth))   arnoldi.jl:135-175 */
    {
        double* v = (double*)malloc((size_t)n * sizeof(double));
        double* w = (double*)malloc((size_t)n * sizeof(double));
        if (!v || !w) { rc = -1; goto done; }
        const double beta0 = beta; /* norm(r) */
        csc_mul(n, n, colptr, rowval, nzval, r, w);
        double alpha = ddot(n, r, w) / (beta0 * beta0);
        memcpy(v, r, (size_t)n * sizeof(double));
        dscal(n, 1.0 / beta0, v);
        dscal(n, 1.0 / beta0, w);
        double bold = dnrm2(n, w);
        daxpy(n, -alpha, v, w);
        double bn = dnrm2(n, w);
        if (orth == 2 || orth == 3) {
            double da = ddot(n, v, w); alpha += da; daxpy(n, -da, v, w); bn = dnrm2(n, w);
        } else if (orth >= 4) {
            while (DBL_EPSILON < bn && bn < eta * bold) {
                bold = bn;
                double da = ddot(n, v, w); alpha += da; daxpy(n, -da, v, w); bn = dnrm2(n, w);
            }
        }
        V[0] = v; nv = 1;
        V[1] = w; /* residual lives in slot nv */
        H[0] = alpha; H[1] = bn;
        k = 1;
        numops += 1;
    }
    for (;;) { /* restart loop :44-149 */
        numiter += 1;
        y[0] = beta;
        k = 1;
        double nres = fabs(H[hidx(2, 1)]);
        R[0] = a0 + a1 * H[hidx(1, 1)];
        givens_fg(R[0], a1 * nres, &gc[0], &gsn[0], &R[0]);
        g1[0] = 0; g2[0] = 1;
        y[1] = 0;
        { double y1 = y[0], y2 = y[1]; y[0] = gc[0] * y1 + gsn[0] * y2; y[1] = -gsn[0] * y1 + gc[0] * y2; }
        beta = fabs(y[1]);
        if (trace && ntrace < trace_cap) trace[ntrace++] = beta;
        int len = 1;
        while (R[(size_t)(k - 1) * K + (k - 1)] != 0 && beta > tol && len < K) { /* :55 */
            /* expand!  arnoldi.jl:199-219 : push!(V, scale(r, 1/beta)) ; w = A v ; orthogonalize ; norm */
            double* vnew = V[nv];
            dscal(n, 1.0 / nres, vnew); /* scale(r, 1/beta): non-mutating in the reference, same values */
            nv += 1;
            double* w = (double*)malloc((size_t)n * sizeof(double));
            if (!w) { rc = -1; goto done; }
            csc_mul(n, n, colptr, rowval, nzval, vnew, w);
            orth_full(n, nv, V, w, hx, orth, eta, tmp, &passes);
            double bn = dnrm2(n, w);
            V[nv] = w;
            len = nv; k = len;
            for (int ii = 1; ii <= k; ++ii) H[hidx(ii, k)] = hx[ii - 1];
            H[hidx(k + 1, k)] = bn;
            nres = bn;
            numops += 1;
            double* Rk = R + (size_t)(k - 1) * K;
            for (int ii = 1; ii <= k - 1; ++ii) Rk[ii - 1] = a1 * H[hidx(ii, k)];
            Rk[k - 1] = a0 + a1 * H[hidx(k, k)];
            for (int ii = 0; ii < k - 1; ++ii) { /* lmul!(gs[i], Rk) :72-76 */
                double u1 = Rk[g1[ii]], u2 = Rk[g2[ii]];
                Rk[g1[ii]] = gc[ii] * u1 + gsn[ii] * u2;
                Rk[g2[ii]] = -gsn[ii] * u1 + gc[ii] * u2;
            }
            if (hypot(Rk[k - 1], a1 * nres) < tol) { /* :79-86 */
                double rr;
                givens_fg(0.0, y[k - 1], &gc[k - 1], &gsn[k - 1], &rr);
                y[k] = rr;
                g1[k - 1] = k; g2[k - 1] = k - 1;
                y[k - 1] = 0; Rk[k - 1] = 0;
            } else { /* :88-90 */
                givens_fg(Rk[k - 1], a1 * nres, &gc[k - 1], &gsn[k - 1], &Rk[k - 1]);
                g1[k - 1] = k - 1; g2[k - 1] = k;
                y[k] = 0;
                double y1 = y[k - 1], y2 = y[k];
                y[k - 1] = gc[k - 1] * y1 + gsn[k - 1] * y2;
                y[k] = -gsn[k - 1] * y1 + gc[k - 1] * y2;
            }
            beta = fabs(y[k]);
            if (trace && ntrace < trace_cap) trace[ntrace++] = beta;
        }
        /* triangular solve :98-102 (ldiv!, dense/linalg.jl:96-106) */
        int kk2 = (R[(size_t)(k - 1) * K + (k - 1)] == 0 && y[k - 1] == 0) ? k - 1 : k;
        for (int j = kk2 - 1; j >= 0; --j) {
            y[j] = y[j] / R[(size_t)j * K + j];
            for (int ii = 0; ii < j; ++ii) y[ii] -= R[(size_t)j * K + ii] * y[j];
        }
        for (int ii = 0; ii < k; ++ii) daxpy(n, y[ii], V[ii], x); /* :105-108 */
        if (beta > tol && numiter < maxiter) { /* :110-117 */
            dscal(n, 1.0 / nres, V[nv]); /* push!(V, scale!!(w, 1/normres)) */
            for (int ii = 0; ii < k; ++ii) { /* rmul!(V, gs[i]')  dense/givens.jl:30-36 with s -> -s */
                double* q1 = V[g1[ii]]; double* q2 = V[g2[ii]];
                const double c = gc[ii], s = -gsn[ii];
#pragma omp parallel for simd schedule(static)
                for (i = 0; i < n; ++i) { double u1 = q1[i], u2 = q2[i]; q1[i] = c * u1 - s * u2; q2[i] = s * u1 + c * u2; }
            }
            const double yk = y[k];
            const double* vk = V[k];
#pragma omp parallel for simd schedule(static)
            for (i = 0; i < n; ++i) r[i] = vk[i] * yk; /* r = scale!!(r, V[k+1], y[k+1]) */
        } else { /* :119-132 */
            memcpy(r, b, (size_t)n * sizeof(double));
            csc_mul(n, n, colptr, rowval, nzval, x, t);
#pragma omp parallel for simd schedule(static)
            for (i = 0; i < n; ++i) r[i] -= a0 * x[i] + a1 * t[i]; /* add!!(r, apply(op, x, a0, a1), -1) */
            numops += 1;
            beta = dnrm2(n, r);
            if (beta < tol) { converged = 1; goto done; }
        }
        if (numiter >= maxiter) goto done;
        /* fact = initialize!(ArnoldiIterator(A, r, orth), fact)   arnoldi.jl:176-198 */
        for (int j = 1; j <= nv; ++j) { free(V[j]); V[j] = NULL; }
        {
            double* v = V[0];
            const double nr = dnrm2(n, r);
            memcpy(v, r, (size_t)n * sizeof(double));
            dscal(n, 1.0 / nr, v); /* V[1] = scale!!(V[1], x0, 1/norm(x0)) */
            double* w = (double*)malloc((size_t)n * sizeof(double));
            if (!w) { rc = -1; goto done; }
            csc_mul(n, n, colptr, rowval, nzval, v, w);
            double al = (orth >= 4) ? orth_vec_ir(n, v, w, eta) : orth_vec(n, v, w, orth, eta);
            double bn = dnrm2(n, w);
            nv = 1; V[1] = w;
            H[0] = al; H[1] = bn; /* the reference does NOT count this apply in numops (gmres.jl:147-148) */
        }
    }
done:
    if (info) { info[0] = converged; info[1] = numiter; info[2] = numops; }
    if (normres) *normres = beta;
    if (trace_len) *trace_len = ntrace;
    for (int j = 0; j < K + 2; ++j) free(V[j]);
    free(V); free(H); free(R); free(y); free(gc); free(gsn); free(g1); free(g2); free(hx); free(tmp); free(r); free(t);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Sparse products for the two twins below.  y = A*x on a Julia SparseMatrixCSC runs as a serial scatter over the columns
 * (csc_mul above), i.e. y[i] collects its terms a_ij*x_j in ascending j starting from zero; the same sum formed per row
 * from the CSR image with ascending column indices gives the SAME bits and parallelises over rows.  adjoint(A)*x in
 * SparseArrays is a per-column dot over the stored entries in storage order -- also one independent sum per output.
 * Both products below are therefore bit-identical to the serial Julia loops whatever the thread count.
 * ------------------------------------------------------------------------------------------------ */
/* y[i] = sum over the stored entries of segment i (ptr/idx/val 1-based), in storage order, of val * x[idx] */
static void seg_dot_mul(int64_t nseg, const int64_t* ptr, const int64_t* idx, const double* val, const double* x, double* y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nseg; ++i) {
        double s = 0;
        for (int64_t p = ptr[i] - 1; p < ptr[i + 1] - 1; ++p) s += val[p] * x[idx[p] - 1];
        y[i] = s;
    }
}

/* ------------------------------------------------------------------------------------------------
 * GKL (config 4): initialize (src/factorizations/gkl.jl:183-215) + `steps` expand! calls (:246-269) with the five
 * gklrecurrence methods (:294-404).  A is nr x nc, handed over twice: as Julia's CSC (colptr/rowval/nzval: used for
 * A'u, one dot per column) and as its CSR image with ascending columns (rowptr/colval/rval: used for A v, see above).
 * alphas / betas hold steps+1 doubles; U_out (nr x (steps+2): the U vectors followed by the residual) and
 * V_out (nc x (steps+1)) are optional, column-major.  Returns 0, -1 allocation / zero start vector, -2 if the
 * compatibility test of :192 fails.
 * ------------------------------------------------------------------------------------------------ */
static void mgs_sweep(int64_t n, int m, double** Q, double* w) { /* for q in Q: orthogonalize!!(w, q, MGS)   orthonormal.jl:458-464 */
    for (int j = 0; j < m; ++j) {
        const double s = ddot(n, Q[j], w);
        daxpy(n, -s, Q[j], w);
    }
}
int kkref_gkl(int64_t nr, int64_t nc, const int64_t* colptr, const int64_t* rowval, const double* nzval,
              const int64_t* rowptr, const int64_t* colval, const double* rval, const double* u0, int steps, int orth,
              double eta, int nthreads, double* alphas, double* betas, double* U_out, double* V_out) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    double** U = (double**)calloc((size_t)steps + 2, sizeof(double*));
    double** V = (double**)calloc((size_t)steps + 2, sizeof(double*));
    double* x = (double*)malloc(((size_t)steps + 2) * sizeof(double));
    if (!U || !V || !x) return -1;
    int rc = 0, k = 0;
    double* r = NULL;
    {   /* initialize :183-215 */
        const double beta0 = dnrm2(nr, u0);
        if (beta0 == 0) return -1;
        double* v0 = (double*)malloc((size_t)nc * sizeof(double));
        double* Av0 = (double*)malloc((size_t)nr * sizeof(double));
        double* u = (double*)malloc((size_t)nr * sizeof(double));
        if (!v0 || !Av0 || !u) return -1;
        seg_dot_mul(nc, colptr, rowval, nzval, u0, v0);            /* v0 = apply_adjoint(A, u0) */
        const double alpha = dnrm2(nc, v0) / beta0;
        seg_dot_mul(nr, rowptr, colval, rval, v0, Av0);            /* Av0 = apply_normal(A, v0) */
        const double alpha2 = ddot(nr, u0, Av0) / (beta0 * beta0);
        if (!(fabs(alpha2 - alpha * alpha) <= sqrt(DBL_EPSILON) * fmax(fabs(alpha2), alpha * alpha))) return -2;   /* isapprox :192 */
        memcpy(u, u0, (size_t)nr * sizeof(double));
        dscal(nr, 1.0 / beta0, u);                                  /* u = scale(u0, 1/beta0) */
        dscal(nc, 1.0 / (alpha * beta0), v0);                       /* v = scale(v0, 1/(alpha beta0)) */
        dscal(nr, 1.0 / (alpha * beta0), Av0);                      /* r = scale!!(Av0, 1/(alpha beta0)) */
        daxpy(nr, -alpha, u, Av0);                                  /* r = add!!(r, u, -alpha) */
        U[0] = u; V[0] = v0; r = Av0;
        alphas[0] = alpha; betas[0] = dnrm2(nr, r);
        k = 1;
    }
    for (int it = 0; it < steps && rc == 0; ++it) {   /* expand! :246-269 */
        const double bold = betas[k - 1];
        dscal(nr, 1.0 / bold, r);                                   /* U = push!(U, scale!!(r, 1/beta_old)) */
        U[k] = r;
        const double* u = U[k];
        const int mU = k + 1, mV = k;
        double* v = (double*)malloc((size_t)nc * sizeof(double));
        double* rn = (double*)malloc((size_t)nr * sizeof(double));
        if (!v || !rn) { rc = -1; break; }
        seg_dot_mul(nc, colptr, rowval, nzval, u, v);               /* v = apply_adjoint(A, u) */
        daxpy(nc, -bold, V[mV - 1], v);                             /* v = add!!(v, V[end], -beta) */
        double alpha, beta;
        if (orth == 3) mgs_sweep(nc, mV, V, v);                     /* MGS2 :330-335 */
        alpha = dnrm2(nc, v);
        if (orth >= 4) {                                            /* CGSIR :353-358 (no eps guard) / MGSIR :380-386 */
            double nold = sqrt(alpha * alpha + bold * bold);
            while ((orth == 4 || DBL_EPSILON < alpha) && alpha < eta * nold) {
                nold = alpha;
                if (orth == 4) cgs_pass(nc, mV, V, v, x); else mgs_sweep(nc, mV, V, v);
                alpha = dnrm2(nc, v);
            }
        }
        dscal(nc, 1.0 / alpha, v);                                  /* v = scale!!(v, inv(alpha)) */
        seg_dot_mul(nr, rowptr, colval, rval, v, rn);               /* r = apply_normal(A, v) */
        daxpy(nr, -alpha, u, rn);                                   /* r = add!!(r, u, -alpha) */
        if (orth == 2) cgs_pass(nr, mU, U, rn, x);                  /* CGS2 :320 */
        else if (orth == 3) mgs_sweep(nr, mU, U, rn);               /* MGS2 :341-343 */
        beta = dnrm2(nr, rn);
        if (orth >= 4) {                                            /* :363-372 / :391-402 */
            double nold = sqrt(alpha * alpha + beta * beta);
            while (DBL_EPSILON < beta && beta < eta * nold) {
                nold = beta;
                if (orth == 4) cgs_pass(nr, mU, U, rn, x); else mgs_sweep(nr, mU, U, rn);
                beta = dnrm2(nr, rn);
            }
        }
        V[k] = v;
        r = rn;
        alphas[k] = alpha; betas[k] = beta;
        ++k;
    }
    if (rc == 0 && U_out) {
        for (int j = 0; j < k; ++j) memcpy(U_out + (size_t)j * nr, U[j], (size_t)nr * sizeof(double));
        memcpy(U_out + (size_t)k * nr, r, (size_t)nr * sizeof(double));
    }
    if (rc == 0 && V_out)
        for (int j = 0; j < k; ++j) memcpy(V_out + (size_t)j * nc, V[j], (size_t)nc * sizeof(double));
    for (int j = 0; j < k; ++j) { free(U[j]); free(V[j]); }
    free(r); free(U); free(V); free(x);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * BlockLanczos (config 5): initialize (src/factorizations/blocklanczos.jl:159-198) + expand! (:200-240) until the basis
 * holds at least `target_dim` vectors or `max_steps` block steps were taken, with block_qr! (:312-353, MGS with the DGKS
 * correction, rank drop), block_lanczosrecurrence (:242-263) and block_reorthogonalize! (:277-284).  A symmetric, CSC =
 * CSR (one gather per row = column: the bits of Julia's scatter, see above).  X0: n x bs0 column-major.  H (ldh x ldh,
 * column-major, zeroed here) receives the block-tridiagonal matrix; info = {k, R_size, number of block steps, numops
 * (vector applies incl. the probe of :171)}; sizes[j] (optional, max_steps+1 ints) = block size pushed in step j.
 * V_out (optional) n x (k + R_size): basis followed by the residual block.  Returns 0 / -1.
 * ------------------------------------------------------------------------------------------------ */
/* block_qr!(block, tol) :312-353 -> R (p x p column-major, full, rows of dropped vectors zero), good[], *ngood, *drift */
static void block_qr(int64_t n, int p, double** blk, double tol, double* R, int* good, int* ngood, int* drift) {
    *drift = 0;
    memset(R, 0, (size_t)p * p * sizeof(double));
    char idx[64];
    for (int j = 0; j < p; ++j) idx[j] = 1;
    double beta = sqrt(ddot(n, blk[0], blk[0]));
    if (beta > tol) { R[0] = beta; dscal(n, 1.0 / beta, blk[0]); }
    else { memset(blk[0], 0, (size_t)n * sizeof(double)); idx[0] = 0; }
    for (int j = 1; j < p; ++j) {
        for (int i = 0; i < j; ++i) {                       /* first MGS */
            const double rij = ddot(n, blk[i], blk[j]);
            R[(size_t)j * p + i] = rij;
            daxpy(n, -rij, blk[i], blk[j]);
        }
        beta = dnrm2(n, blk[j]);
        if (tol < beta && beta < 100 * tol) {               /* DGKS reorthogonalization */
            *drift = 1;
            for (int i = 0; i < j; ++i) {
                const double d = ddot(n, blk[i], blk[j]);
                R[(size_t)j * p + i] += d;
                daxpy(n, -d, blk[i], blk[j]);
            }
            beta = dnrm2(n, blk[j]);
        }
        if (beta < tol) { memset(blk[j], 0, (size_t)n * sizeof(double)); idx[j] = 0; }
        else { R[(size_t)j * p + j] = beta; dscal(n, 1.0 / beta, blk[j]); }
    }
    int g = 0;
    for (int j = 0; j < p; ++j) if (idx[j]) good[g++] = j;
    *ngood = g;
}
static void block_reorth(int64_t n, int p, double** Rb, int m, double** V) { /* :277-284 */
    for (int i = 0; i < p; ++i) mgs_sweep(n, m, V, Rb[i]);
}
int kkref_blocklanczos(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* X0,
                       int bs0, int target_dim, int max_steps, double qr_tol, int nthreads, double* H, int ldh, int* info,
                       int* sizes, double* normR_out, double* V_out) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    if (bs0 < 1 || bs0 > 64) return -1;
    const int cap = target_dim + 2 * bs0 + 2;
    double** V = (double**)calloc((size_t)cap, sizeof(double*));
    double* Rm = (double*)malloc((size_t)bs0 * bs0 * sizeof(double));
    double* B = (double*)calloc((size_t)bs0 * bs0, sizeof(double));    /* bs_next x bs, column-major ld bs0 */
    double* M = (double*)malloc((size_t)bs0 * bs0 * sizeof(double));
    double *blk[64], *cpy[64], *AX[64];
    int good[64];
    if (!V || !Rm || !B || !M) return -1;
    memset(H, 0, (size_t)ldh * ldh * sizeof(double));
    int nV = 0, numops = 0, nsteps = 0, rc = 0;
    /* initialize :159-198 */
    {
        double* probe = (double*)malloc((size_t)n * sizeof(double));
        if (!probe) return -1;
        seg_dot_mul(n, colptr, rowval, nzval, X0, probe);   /* Ax0 = apply(A, x0): only its type matters (:171-172) */
        free(probe);
        numops += 1;
    }
    for (int j = 0; j < bs0; ++j) {
        blk[j] = (double*)malloc((size_t)n * sizeof(double));
        if (!blk[j]) return -1;
        memcpy(blk[j], X0 + (size_t)j * n, (size_t)n * sizeof(double));   /* X1 = scale.(X0, one(alpha)) */
    }
    int ng = 0, drift = 0;
    block_qr(n, bs0, blk, qr_tol, Rm, good, &ng, &drift);
    for (int j = 0, g = 0; j < bs0; ++j) {
        if (g < ng && good[g] == j) { V[nV++] = blk[j]; ++g; } else free(blk[j]);
    }
    int bs = ng;
    if (bs == 0) return -1;
    for (int j = 0; j < bs; ++j) {
        AX[j] = (double*)malloc((size_t)n * sizeof(double));
        if (!AX[j]) return -1;
        seg_dot_mul(n, colptr, rowval, nzval, V[j], AX[j]);
    }
    numops += bs;
    for (int j = 0; j < bs; ++j)
        for (int i = 0; i < bs; ++i) M[(size_t)j * bs0 + i] = ddot(n, V[i], AX[j]);      /* M1 = block_inner(X1, AX1) */
    for (int j = 0; j < bs; ++j)
        for (int i = 0; i < bs; ++i) H[(size_t)j * ldh + i] = M[(size_t)j * bs0 + i];
    for (int j = 0; j < bs; ++j)
        for (int i = 0; i < bs; ++i) daxpy(n, -M[(size_t)j * bs0 + i], V[i], AX[j]);     /* first residual :187-191 */
    int k = bs, R_size = bs;
    if (sizes) sizes[0] = bs;
    /* expand! :200-240 */
    while (k < target_dim && nsteps < max_steps && rc == 0) {
        const int p = R_size;
        if (k + p > ldh || nV + p > cap) break;
        for (int j = 0; j < p; ++j) {                        /* Rcopy = copy(R) */
            cpy[j] = (double*)malloc((size_t)n * sizeof(double));
            if (!cpy[j]) { rc = -1; break; }
            memcpy(cpy[j], AX[j], (size_t)n * sizeof(double));
        }
        if (rc) break;
        block_qr(n, p, AX, qr_tol, Rm, good, &ng, &drift);
        for (int j = 0; j < p; ++j)
            for (int g = 0; g < ng; ++g) B[(size_t)j * bs0 + g] = Rm[(size_t)j * p + good[g]];          /* B = R[good_idx, :] */
        if (drift) {                                         /* :211-215 */
            block_reorth(n, p, AX, nV, V);
            block_qr(n, p, AX, qr_tol, Rm, good, &ng, &drift);
            for (int j = 0; j < p; ++j)
                for (int g = 0; g < ng; ++g) B[(size_t)j * bs0 + g] = ddot(n, AX[good[g]], cpy[j]);     /* block_inner(R[good], Rcopy) */
        }
        for (int j = 0; j < p; ++j) free(cpy[j]);
        const int bsn = ng;
        if (bsn == 0) break;
        for (int j = 0, g = 0; j < p; ++j) {                 /* push!(V, R[good_idx]) */
            if (g < ng && good[g] == j) { V[nV++] = AX[j]; ++g; } else free(AX[j]);
        }
        for (int j = 0; j < p; ++j)                          /* H[k+1:k+bsn, k-bs+1:k] = B ; transpose block :220-221 */
            for (int g = 0; g < bsn; ++g) {
                H[(size_t)(k - p + j) * ldh + (k + g)] = B[(size_t)j * bs0 + g];
                H[(size_t)(k + g) * ldh + (k - p + j)] = B[(size_t)j * bs0 + g];
            }
        /* block_lanczosrecurrence :242-263: X = last bsn vectors, Xprev = the p before them */
        double** X = V + (nV - bsn);
        double** Xp = V + (nV - bsn - p);
        for (int j = 0; j < bsn; ++j) {
            AX[j] = (double*)malloc((size_t)n * sizeof(double));
            if (!AX[j]) { rc = -1; break; }
            seg_dot_mul(n, colptr, rowval, nzval, X[j], AX[j]);
        }
        if (rc) break;
        numops += bsn;
        for (int j = 0; j < bsn; ++j)
            for (int i = 0; i < bsn; ++i) M[(size_t)j * bs0 + i] = ddot(n, X[i], AX[j]);
        for (int j = 0; j < bsn; ++j) {
            for (int i = 0; i < bsn; ++i) daxpy(n, -M[(size_t)j * bs0 + i], X[i], AX[j]);
            for (int i = 0; i < p; ++i) daxpy(n, -B[(size_t)i * bs0 + j], Xp[i], AX[j]);     /* -conj(B[j, i]) */
        }
        block_reorth(n, bsn, AX, nV, V);
        for (int j = 0; j < bsn; ++j)
            for (int i = 0; i < bsn; ++i) H[(size_t)(k + j) * ldh + (k + i)] = M[(size_t)j * bs0 + i];
        k += bsn;
        R_size = bsn;
        ++nsteps;
        if (sizes) sizes[nsteps] = bsn;
    }
    double nr2 = 0;
    for (int j = 0; j < R_size; ++j) { const double t = dnrm2(n, AX[j]); nr2 += t * t; }       /* norm(Block) = norm of the norms */
    if (normR_out) *normR_out = sqrt(nr2);
    if (info) { info[0] = k; info[1] = R_size; info[2] = nsteps; info[3] = numops; }
    if (rc == 0 && V_out) {
        for (int j = 0; j < nV; ++j) memcpy(V_out + (size_t)j * n, V[j], (size_t)n * sizeof(double));
        for (int j = 0; j < R_size; ++j) memcpy(V_out + (size_t)(nV + j) * n, AX[j], (size_t)n * sizeof(double));
    }
    for (int j = 0; j < nV; ++j) free(V[j]);
    for (int j = 0; j < R_size; ++j) free(AX[j]);
    free(V); free(Rm); free(B); free(M);
    return rc;
}

int kkref_num_threads(void) { return omp_get_max_threads(); }
