/* A plain-C99 client of include/krylov_hip.h (missing 5): what an FFI binding sees -- the header and -lkrylov_hip,
 * nothing else (no ctypes table, no Python).  Builds a 2-D 5-point Laplacian (nx x ny, natural ordering, CSR, 0-based), runs
 * initialize + NSTEPS expand! of the Lanczos factorization (src/factorizations/lanczos.jl:180-222, 250-291) with
 * ModifiedGramSchmidt2 through the C ABI and prints alpha / beta as hex floats, one pair per line, for the test to compare with
 * the oracle.  The calling convention mirrors the reference's own ccall wrappers (src/dense/linalg.jl:428-454: status code out,
 * results through pointers).  TEST INFRASTRUCTURE.
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include tests/c_client/client.c -L krylovkit.jl_amd/lib -lkrylov_hip -lm -o client */
#include <stdio.h>
#include <stdlib.h>
#include "krylov_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int st_ = (call);                                                                  \
        if (st_ != KK_OK) {                                                                \
            fprintf(stderr, "%s failed with status %d: %s\n", #call, st_, kk_last_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 40, ny = argc > 2 ? atoi(argv[2]) : 30, nsteps = argc > 3 ? atoi(argv[3]) : 3;
    const int64_t n = (int64_t)nx * ny;
    int64_t* rowptr = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
    int32_t* col = (int32_t*)malloc((size_t)(5 * n) * sizeof(int32_t));
    double* val = (double*)malloc((size_t)(5 * n) * sizeof(double));
    double* x0 = (double*)malloc((size_t)n * sizeof(double));
    if (!rowptr || !col || !val || !x0) return 2;
    int64_t nnz = 0;
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            const int64_t r = (int64_t)j * nx + i;
            rowptr[r] = nnz;
            if (j > 0) { col[nnz] = (int32_t)(r - nx); val[nnz++] = -1.0; }
            if (i > 0) { col[nnz] = (int32_t)(r - 1); val[nnz++] = -1.0; }
            col[nnz] = (int32_t)r; val[nnz++] = 4.0;
            if (i < nx - 1) { col[nnz] = (int32_t)(r + 1); val[nnz++] = -1.0; }
            if (j < ny - 1) { col[nnz] = (int32_t)(r + nx); val[nnz++] = -1.0; }
        }
    rowptr[n] = nnz;
    for (int64_t r = 0; r < n; ++r) x0[r] = 1.0 + (double)((r * 7919) % 1000) / 1000.0;   /* any fixed, non-degenerate start vector */

    if (kk_version() < 100) { fprintf(stderr, "unexpected library version %d\n", kk_version()); return 3; }
    kk_ctx ctx = NULL;
    kk_op A = NULL;
    kk_basis V = NULL;
    CHECK(kk_ctx_create(0, &ctx));
    CHECK(kk_csr_create(ctx, n, n, nnz, rowptr, col, val, 0, KK_OP_SYMMETRIC, &A));
    CHECK(kk_basis_create(ctx, n, nsteps + 3, &V));
    CHECK(kk_basis_upload(V, 0, x0));
    double alpha = 0.0, beta = 0.0;
    CHECK(kk_lanczos_initialize(A, V, 0, KK_MGS2, 0.0, &alpha, &beta));
    printf("%a %a\n", alpha, beta);
    for (int k = 1; k <= nsteps; ++k) {
        int npasses = 0;
        const double beta_old = beta;
        CHECK(kk_lanczos_expand(A, V, 0, k, KK_MGS2, 0.0, beta_old, &alpha, &beta, &npasses));
        printf("%a %a\n", alpha, beta);
    }
    /* an error path through the same boundary: a column out of range comes back as a status + message, not as a crash */
    if (kk_basis_upload(V, nsteps + 100, x0) == KK_OK) { fprintf(stderr, "out-of-range upload was accepted\n"); return 4; }
    if (kk_last_error()[0] == '\0') { fprintf(stderr, "no error message\n"); return 5; }
    CHECK(kk_basis_free(V));
    CHECK(kk_op_free(A));
    CHECK(kk_ctx_destroy(ctx));
    free(rowptr); free(col); free(val); free(x0);
    return 0;
}
