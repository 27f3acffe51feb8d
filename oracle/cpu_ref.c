/*
 * cpu_ref.c -- C twin of oracle/krylov_oracle.py for the Lanczos expand! path.
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

static double ddot(int64_t n, const double* x, const double* y) { /* inner(x,y) */
    double s = 0;
#pragma omp parallel for simd reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += x[i] * y[i];
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

int kkref_num_threads(void) { return omp_get_max_threads(); }
