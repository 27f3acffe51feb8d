// libkrylov_hip.so, C ABI part 2: sparse operators -- host CSR/CSC intake, device formats (ELL, CSR, SELL-64-sigma,
// column-tiled SELL), apply / apply_adjoint / affine apply, ghost columns and halo hooks of row-sharded operators.
#include "kk_host.h"

// ------------------------------------------------------------------------------------------
// operators
// ------------------------------------------------------------------------------------------
static void free_sparse(kk_sparse_dev& M) {
    (void)hipFree(M.ell_col); (void)hipFree(M.ell_val); (void)hipFree(M.dia_val);
    (void)hipFree(M.rowptr); (void)hipFree(M.colind); (void)hipFree(M.val);
    (void)hipFree(M.sell_off); (void)hipFree(M.sell_perm); (void)hipFree(M.sell_col); (void)hipFree(M.sell_val);
    for (int t = 0; t < M.ntiles; ++t) free_sparse(M.tiles[t]);
    delete[] M.tiles;
    M = kk_sparse_dev();
}

// SELL-64-sigma image of a host CSR matrix on the device (fills the sell_* fields of M, sets format = 2)
static int build_sell(const kk_host_csr& h, kk_sparse_dev& M, int64_t sigma = 64 * 64) {
    const int64_t nrows = h.nrows;
    // SELL-64-sigma: sort rows by length inside windows of sigma rows, slice into chunks of 64
    M.format = 2;
    M.sell_sigma = sigma;
    const int64_t C = 64;
    const int64_t nchunks = (nrows + C - 1) / C;
    std::vector<int32_t> perm((size_t)nchunks * C, -1);
    std::vector<int32_t> order(nrows);
    for (int64_t i = 0; i < nrows; ++i) order[i] = (int32_t)i;
    for (int64_t w0 = 0; w0 < nrows; w0 += sigma) {
        const int64_t w1 = std::min(nrows, w0 + sigma);
        std::stable_sort(order.begin() + w0, order.begin() + w1, [&](int32_t a, int32_t b) {
            return (h.rowptr[a + 1] - h.rowptr[a]) > (h.rowptr[b + 1] - h.rowptr[b]);
        });
    }
    for (int64_t i = 0; i < nrows; ++i) perm[i] = order[i];
    std::vector<int64_t> coff(nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) {
        int64_t wmax = 0;
        for (int64_t l = 0; l < C; ++l) {
            const int32_t r = perm[c * C + l];
            if (r >= 0) wmax = std::max(wmax, h.rowptr[r + 1] - h.rowptr[r]);
        }
        coff[c + 1] = coff[c] + wmax * C;
    }
    const int64_t total = coff[nchunks];
    std::vector<int32_t> sc((size_t)std::max<int64_t>(total, 1), 0);
    std::vector<double> sv((size_t)std::max<int64_t>(total, 1), 0.0);
    for (int64_t c = 0; c < nchunks; ++c)
        for (int64_t l = 0; l < C; ++l) {
            const int32_t r = perm[c * C + l];
            if (r < 0) continue;
            int64_t k = 0;
            for (int64_t p = h.rowptr[r]; p < h.rowptr[r + 1]; ++p, ++k) {
                sc[coff[c] + k * C + l] = h.col[p];
                sv[coff[c] + k * C + l] = h.val[p];
            }
        }
    M.sell_nchunks = nchunks;
    KK_HIP(hipMalloc(&M.sell_off, (nchunks + 1) * sizeof(int64_t)));
    KK_HIP(hipMalloc(&M.sell_perm, perm.size() * sizeof(int32_t)));
    KK_HIP(hipMalloc(&M.sell_col, sc.size() * sizeof(int32_t)));
    KK_HIP(hipMalloc(&M.sell_val, sv.size() * sizeof(double)));
    KK_HIP(hipMemcpy(M.sell_off, coff.data(), (nchunks + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    KK_HIP(hipMemcpy(M.sell_perm, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    KK_HIP(hipMemcpy(M.sell_col, sc.data(), sc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    KK_HIP(hipMemcpy(M.sell_val, sv.data(), sv.size() * sizeof(double), hipMemcpyHostToDevice));
    M.bytes = (nchunks + 1) * 8 + perm.size() * 4 + sc.size() * 12;
    return KK_OK;
}

// Column-tiled SELL for operators whose gathers have no locality (rows touching columns all over a vector that does
// not fit the 4 MB L2 of an XCD, e.g. the random rectangular map of the GKL configuration and its transpose): the
// columns are cut into tiles of `tile_cols`, every tile is its own SELL matrix over ALL rows, and an apply runs the
// tiles one after the other accumulating into y, so that each launch gathers from one L2-resident slice of x.
static int build_tiled(const kk_host_csr& h, int64_t tile_cols, kk_sparse_dev& M) {
    const int64_t nrows = h.nrows, nnz = h.rowptr[nrows];
    const int T = (int)((h.ncols + tile_cols - 1) / tile_cols);
    M.format = 3;
    M.ntiles = T;
    M.tiles = new kk_sparse_dev[T];
    M.tile_cols = tile_cols;
    // pass 1: entries per (tile, row); pass 2: scatter into per-tile CSR
    std::vector<kk_host_csr> ht(T);
    for (int t = 0; t < T; ++t) {
        ht[t].nrows = nrows; ht[t].ncols = h.ncols;
        ht[t].rowptr.assign(nrows + 1, 0);
    }
    for (int64_t i = 0; i < nrows; ++i)
        for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) ht[h.col[p] / tile_cols].rowptr[i + 1]++;
    for (int t = 0; t < T; ++t) {
        for (int64_t i = 0; i < nrows; ++i) ht[t].rowptr[i + 1] += ht[t].rowptr[i];
        ht[t].col.resize(ht[t].rowptr[nrows]);
        ht[t].val.resize(ht[t].rowptr[nrows]);
    }
    {
        std::vector<int64_t> cur(T);
        for (int64_t i = 0; i < nrows; ++i) {
            for (int t = 0; t < T; ++t) cur[t] = ht[t].rowptr[i];
            for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) {
                const int t = (int)(h.col[p] / tile_cols);
                const int64_t q = cur[t]++;
                ht[t].col[q] = h.col[p];
                ht[t].val[q] = h.val[p];
            }
        }
    }
    M.bytes = 0;
    for (int t = 0; t < T; ++t) {
        kk_sparse_dev& S = M.tiles[t];
        S.nrows = nrows; S.ncols = h.ncols; S.nnz = ht[t].rowptr[nrows];
        // sorting window = R rounds of the 256 rows of one thread block (k_spmv_sellw<R>): the wider the window the less
        // padding (= fewer gather instructions, which is what bounds these applies), while the y update stays coalesced
        int R = 4;
        if (const char* rs = getenv("KK_SELLW_ROUNDS")) R = atoi(rs);
        if (R != 1 && R != 2 && R != 4 && R != 8) R = 4;
        KK_TRY(build_sell(ht[t], S, (int64_t)KK_TPB * R));
        M.bytes += S.bytes;
        kk_host_csr().rowptr.swap(ht[t].rowptr);
        std::vector<int32_t>().swap(ht[t].col);
        std::vector<double>().swap(ht[t].val);
    }
    (void)nnz;
    return KK_OK;
}

// mean distance between the smallest and the largest column index of a row (sampled): small for stencils / banded
// operators whose gathers are cache friendly as they are, ~ncols for random sparsity
static double mean_row_span(const kk_host_csr& h) {
    const int64_t step = std::max<int64_t>(1, h.nrows / 65536);
    double sum = 0;
    int64_t cnt = 0;
    for (int64_t i = 0; i < h.nrows; i += step) {
        if (h.rowptr[i + 1] == h.rowptr[i]) continue;
        int32_t lo = h.col[h.rowptr[i]], hi = lo;
        for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) { lo = std::min(lo, h.col[p]); hi = std::max(hi, h.col[p]); }
        sum += (double)(hi - lo);
        ++cnt;
    }
    return cnt ? sum / cnt : 0.0;
}

// Grid-stencil structure of a square operator: every column offset col - row lies in {-D, 0, +D} + {-1, 0, +1} with one far
// offset D >= 64 (5-point / 9-point discretisations on an nx x ny grid in natural ordering, D = nx; the coefficients may vary
// from row to row).  Such an operator gets dense diagonals next to its ELL arrays; the multi-column apply then sweeps the grid
// lines with a three-line register window instead of gathering (k_spmm_dia).  Returns quietly when the structure is absent.
static int detect_stencil(const kk_host_csr& h, kk_sparse_dev& M) {
    M.dia_D = 0;
    if (getenv("KK_NO_DIA")) return KK_OK;
    const int64_t n = h.nrows, nnz = h.rowptr[n];
    if (n != h.ncols || n < 4096 || nnz < n) return KK_OK;
    if (M.dia_val) { (void)hipFree(M.dia_val); M.dia_val = nullptr; }
    int64_t offs[9];
    int no = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) {
            const int64_t o = (int64_t)h.col[p] - i;
            bool seen = false;
            for (int q = 0; q < no; ++q) seen |= (offs[q] == o);
            if (!seen) {
                if (no == 9) return KK_OK;
                offs[no++] = o;
            }
        }
    int64_t amax = 0;
    for (int q = 0; q < no; ++q) amax = std::max<int64_t>(amax, offs[q] < 0 ? -offs[q] : offs[q]);
    auto fits = [&](int64_t Dc) {   // every offset = a * Dc + s with a, s in {-1, 0, 1}
        for (int q = 0; q < no; ++q) {
            const int64_t a = offs[q] < 0 ? -offs[q] : offs[q];
            if (!(a <= 1 || (a >= Dc - 1 && a <= Dc + 1))) return false;
        }
        return true;
    };
    int64_t D = 0;
    if (amax >= 64 && fits(amax)) D = amax;
    else if (amax - 1 >= 64 && fits(amax - 1)) D = amax - 1;
    else return KK_OK;
    bool corner = false;
    for (int q = 0; q < no; ++q) {
        const int64_t a = offs[q] < 0 ? -offs[q] : offs[q];
        if (a == D - 1 || a == D + 1) corner = true;
    }
    if (D < 64 || D >= n) return KK_OK;
    const int pts = corner ? 9 : 5;
    if ((double)pts * n > 1.6 * (double)nnz + 4096) return KK_OK;   // diagonals mostly empty: the gather format is denser
    const int64_t dld = (n + 63) / 64 * 64;
    std::vector<double> dv((size_t)pts * dld, 0.0);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p) {
            const int64_t o = (int64_t)h.col[p] - i;
            int slot;
            if (pts == 5) slot = o == -D ? 0 : (o == -1 ? 1 : (o == 0 ? 2 : (o == 1 ? 3 : 4)));
            else {
                const int a = o < -1 ? 0 : (o > 1 ? 2 : 1);                  // line -1 / 0 / +1
                const int64_t sft = o - (int64_t)(a - 1) * D;                // -1 / 0 / +1 inside the line
                slot = a * 3 + (int)(sft + 1);
            }
            dv[(size_t)slot * dld + i] += h.val[p];   // duplicate entries of a row add up, as in the CSR apply
        }
    KK_HIP(hipMalloc(&M.dia_val, dv.size() * sizeof(double)));
    KK_HIP(hipMemcpy(M.dia_val, dv.data(), dv.size() * sizeof(double), hipMemcpyHostToDevice));
    M.dia_D = D; M.dia_pts = pts; M.dia_ld = dld;
    M.bytes += (int64_t)dv.size() * 8;
    // Constant coefficients?  (Laplacians, constant convection-diffusion operators ...)  Diagonal q must hold ONE value c_q
    // wherever its neighbour exists -- inside the matrix and on the same grid line for the +-1 shifts -- and nothing elsewhere.
    // The position of a row inside its line is (i + phase) % D; the phase is read off the first gap of the "+1" diagonal
    // (a row-sharded block need not start at a line boundary).
    M.dia_const = false;
    {
        int64_t o[9]; int b[9];
        if (pts == 5) { const int64_t o5[5] = {-D, -1, 0, 1, D}; const int b5[5] = {0, -1, 0, 1, 0}; for (int q = 0; q < 5; ++q) { o[q] = o5[q]; b[q] = b5[q]; } }
        else { for (int q = 0; q < 9; ++q) { b[q] = q % 3 - 1; o[q] = (int64_t)(q / 3 - 1) * D + b[q]; } }
        const int qp = pts == 5 ? 3 : 5;   // the slot of offset +1
        int64_t phase = -1;
        for (int64_t i = 0; i + 1 < n && phase < 0; ++i)
            if (dv[(size_t)qp * dld + i] == 0.0) phase = ((D - 1 - i) % D + D) % D;
        bool ok = phase >= 0;
        double cq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < pts && ok; ++q) {
            bool have = false;
            const double* dq = dv.data() + (size_t)q * dld;
            for (int64_t i = 0; i < n; ++i) {
                const int64_t j = i + o[q], ix = (i + phase) % D;
                const bool valid = j >= 0 && j < n && ix + b[q] >= 0 && ix + b[q] < D;
                if (valid) {
                    if (!have) { cq[q] = dq[i]; have = true; }
                    if (dq[i] != cq[q]) { ok = false; break; }
                } else if (dq[i] != 0.0) { ok = false; break; }
            }
        }
        if (ok) {
            M.dia_const = true;
            M.dia_phase = phase;
            for (int q = 0; q < 9; ++q) M.dia_c[q] = cq[q];
        }
    }
    return KK_OK;
}

// Row-sharded stencil: the rows of the local block that reference no ghost column form (for a partition along grid lines)
// one long run [int_lo, int_hi); if the local square part of the block has the grid-stencil structure, those rows can use
// the diagonal kernels and only the boundary strips need the gather kernels with the ghost buffer.
static int detect_stencil_sharded(const kk_host_csr& h, int64_t n_local, kk_sparse_dev& M) {
    M.int_lo = M.int_hi = 0;
    if (getenv("KK_NO_DIA") || M.format != 0 || n_local < 4096) return KK_OK;
    // longest run of rows without ghost references
    int64_t best_lo = 0, best_hi = 0, run_lo = 0;
    for (int64_t i = 0; i <= n_local; ++i) {
        bool ghost = (i == n_local);
        if (!ghost)
            for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1] && !ghost; ++p) ghost = h.col[p] >= n_local;
        if (ghost) {
            if (i - run_lo > best_hi - best_lo) { best_lo = run_lo; best_hi = i; }
            run_lo = i + 1;
        }
    }
    best_lo = (best_lo + 1) & ~(int64_t)1;   // both ends even: the kernels handle rows in pairs
    best_hi = best_hi & ~(int64_t)1;
    if (best_hi - best_lo < n_local / 2) return KK_OK;
    // local square part (ghost entries dropped: the rows that have any are outside the interior and never use the diagonals)
    kk_host_csr sq;
    sq.nrows = n_local; sq.ncols = n_local;
    sq.rowptr.assign(n_local + 1, 0);
    sq.col.reserve(h.col.size()); sq.val.reserve(h.val.size());
    for (int64_t i = 0; i < n_local; ++i) {
        for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p)
            if (h.col[p] < n_local) { sq.col.push_back(h.col[p]); sq.val.push_back(h.val[p]); }
        sq.rowptr[i + 1] = (int64_t)sq.col.size();
    }
    KK_TRY(detect_stencil(sq, M));
    if (M.dia_D > 0) { M.int_lo = best_lo; M.int_hi = best_hi; }
    return KK_OK;
}

static int upload_sparse(kk_ctx c, const kk_host_csr& h, kk_sparse_dev& M) {
    const int64_t nrows = h.nrows, nnz = h.rowptr[nrows];
    M.nrows = nrows; M.ncols = h.ncols; M.nnz = nnz;
    // dimensions stay below 2^31 (int32 column indices on the device); the number of stored entries does not: every element
    // offset of the ELL / SELL / tiled / diagonal images is 64-bit on the host and in the kernels.  Only the plain CSR image
    // (KK_SPMV_FORMAT=csr, a debugging format) keeps int32 row pointers.
    KK_CHECK(h.ncols < (int64_t)1 << 31 && nrows < (int64_t)1 << 31, KK_ERR_UNSUPPORTED, "dimension >= 2^31 not supported");
    int64_t maxw = 0;
    for (int64_t i = 0; i < nrows; ++i) maxw = std::max(maxw, h.rowptr[i + 1] - h.rowptr[i]);
    const bool force_csr = getenv("KK_SPMV_FORMAT") && !strcmp(getenv("KK_SPMV_FORMAT"), "csr");
    KK_CHECK(!force_csr || nnz < (int64_t)1 << 31, KK_ERR_UNSUPPORTED, "KK_SPMV_FORMAT=csr: nnz >= 2^31 not supported (int32 row pointers)");
    const bool force_ell = getenv("KK_SPMV_FORMAT") && !strcmp(getenv("KK_SPMV_FORMAT"), "ell");
    const bool force_sell = getenv("KK_SPMV_FORMAT") && !strcmp(getenv("KK_SPMV_FORMAT"), "sell");
    const bool ell = !force_csr && !force_sell && (force_ell || (maxw <= 64 && (double)maxw * nrows <= 1.25 * (double)nnz + 4096));
    // column tiling: default tile = 3 MB of the gathered vector (an XCD's L2 is 4 MB; measured optimum on the
    // config-4 operator, flat between 2 and 4 MB); KK_SPMV_TILE_COLS overrides (0 = never)
    int64_t tile_cols = 393216;
    if (const char* tc = getenv("KK_SPMV_TILE_COLS")) tile_cols = atoll(tc);
    const bool fmt_forced = force_csr || force_ell || force_sell;
    if (!fmt_forced && tile_cols > 0 && 2 * h.ncols > 3 * tile_cols && nnz > 0 && mean_row_span(h) > 1.5 * (double)tile_cols)
        return build_tiled(h, tile_cols, M);
    if (ell) {
        M.format = 0;
        M.width = (int)std::max<int64_t>(maxw, 1);
        M.ell_ld = (nrows + 63) / 64 * 64;
        std::vector<int32_t> ec((size_t)M.ell_ld * M.width, 0);
        std::vector<double> ev((size_t)M.ell_ld * M.width, 0.0);
        for (int64_t i = 0; i < nrows; ++i) {
            int k = 0;
            for (int64_t p = h.rowptr[i]; p < h.rowptr[i + 1]; ++p, ++k) {
                ec[(size_t)k * M.ell_ld + i] = h.col[p];
                ev[(size_t)k * M.ell_ld + i] = h.val[p];
            }
        }
        KK_HIP(hipMalloc(&M.ell_col, ec.size() * sizeof(int32_t)));
        KK_HIP(hipMalloc(&M.ell_val, ev.size() * sizeof(double)));
        KK_HIP(hipMemcpy(M.ell_col, ec.data(), ec.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        KK_HIP(hipMemcpy(M.ell_val, ev.data(), ev.size() * sizeof(double), hipMemcpyHostToDevice));
        M.bytes = ec.size() * 4 + ev.size() * 8;
        KK_TRY(detect_stencil(h, M));
    } else if (!force_csr) {
        KK_TRY(build_sell(h, M));
    } else {
        M.format = 1;
        std::vector<int32_t> rp(nrows + 1);
        for (int64_t i = 0; i <= nrows; ++i) rp[i] = (int32_t)h.rowptr[i];
        KK_HIP(hipMalloc(&M.rowptr, (nrows + 1) * sizeof(int32_t)));
        KK_HIP(hipMalloc(&M.colind, std::max<int64_t>(nnz, 1) * sizeof(int32_t)));
        KK_HIP(hipMalloc(&M.val, std::max<int64_t>(nnz, 1) * sizeof(double)));
        KK_HIP(hipMemcpy(M.rowptr, rp.data(), (nrows + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
        KK_HIP(hipMemcpy(M.colind, h.col.data(), nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        KK_HIP(hipMemcpy(M.val, h.val.data(), nnz * sizeof(double), hipMemcpyHostToDevice));
        const double avg = nrows ? (double)nnz / nrows : 1;
        int L = 2;
        while (L * 2 <= avg && L < 64) L *= 2;
        M.lanes_per_row = L;
        M.bytes = (nrows + 1) * 4 + nnz * 12;
    }
    return KK_OK;
}

// pointer array of a compressed format: starts at index_base, monotone, ends at nnz (a malformed array would send the
// packing loops out of bounds on the host)
static int check_ptr_array(const char* who, const int64_t* ptr, int64_t n, int64_t nnz, int index_base) {
    KK_CHECK(ptr[0] - index_base == 0, KK_ERR_DIM, "%s: pointer array does not start at %d", who, index_base);
    for (int64_t i = 0; i < n; ++i)
        KK_CHECK(ptr[i] <= ptr[i + 1], KK_ERR_DIM, "%s: pointer array decreases at position %lld", who, (long long)i);
    KK_CHECK(ptr[n] - index_base == nnz, KK_ERR_DIM, "%s: last pointer entry != nnz", who);
    return KK_OK;
}

static void transpose_csr(const kk_host_csr& a, kk_host_csr& t) {
    t.nrows = a.ncols; t.ncols = a.nrows;
    const int64_t nnz = a.rowptr[a.nrows];
    t.rowptr.assign(t.nrows + 1, 0);
    t.col.resize(nnz); t.val.resize(nnz);
    for (int64_t p = 0; p < nnz; ++p) t.rowptr[a.col[p] + 1]++;
    for (int64_t i = 0; i < t.nrows; ++i) t.rowptr[i + 1] += t.rowptr[i];
    std::vector<int64_t> cur(t.rowptr.begin(), t.rowptr.end() - 1);
    for (int64_t i = 0; i < a.nrows; ++i)
        for (int64_t p = a.rowptr[i]; p < a.rowptr[i + 1]; ++p) {
            const int64_t q = cur[a.col[p]]++;
            t.col[q] = (int32_t)i;
            t.val[q] = a.val[p];
        }
}

KK_API int kk_csr_create(kk_ctx c, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* rowptr,
                             const int32_t* colind, const double* val, int index_base, int flags, kk_op* out) {
    KK_CHECK(c && out && rowptr && (nnz == 0 || (colind && val)), KK_ERR_INVALID, "kk_csr_create: null arg");
    KK_CHECK(nrows > 0 && ncols > 0 && nnz >= 0, KK_ERR_INVALID, "kk_csr_create: bad dimensions");
    KK_CHECK(index_base == 0 || index_base == 1, KK_ERR_INVALID, "index_base must be 0 or 1");
    KK_TRY(check_ptr_array("kk_csr_create", rowptr, nrows, nnz, index_base));
    KK_CHECK(nrows < (int64_t)1 << 31 && ncols < (int64_t)1 << 31, KK_ERR_UNSUPPORTED,
             "kk_csr_create: dimensions >= 2^31 are not supported (int32 column indices on the device)");
    KK_HIP(hipSetDevice(c->device));
    kk_op op = new kk_op_s();
    op->ctx = c; op->nrows = nrows; op->ncols = ncols; op->nnz = nnz; op->flags = flags;
    kk_host_csr& h = op->hA;
    h.nrows = nrows; h.ncols = ncols;
    h.rowptr.resize(nrows + 1);
    for (int64_t i = 0; i <= nrows; ++i) h.rowptr[i] = rowptr[i] - index_base;
    h.col.resize(nnz); h.val.assign(val, val + nnz);
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t cc = (int64_t)colind[p] - index_base;
        if (cc < 0 || cc >= ncols) {
            delete op;
            kk_set_error("kk_csr_create: column index %lld out of range at entry %lld", (long long)cc, (long long)p);
            return KK_ERR_DIM;
        }
        h.col[p] = (int32_t)cc;
    }
    int s = upload_sparse(c, h, op->A);
    if (s != KK_OK) { free_sparse(op->A); delete op; return s; }
    if (flags & KK_OP_SYMMETRIC) { kk_host_csr().rowptr.swap(h.rowptr); h.col.clear(); h.col.shrink_to_fit(); h.val.clear(); h.val.shrink_to_fit(); }
    *out = op;
    return KK_OK;
}

KK_API int kk_csc_create(kk_ctx c, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* colptr,
                             const int64_t* rowval, const double* nzval, int index_base, int flags, kk_op* out) {
    KK_CHECK(c && out && colptr && (nnz == 0 || (rowval && nzval)), KK_ERR_INVALID, "kk_csc_create: null arg");
    KK_CHECK(nrows > 0 && ncols > 0 && nnz >= 0, KK_ERR_INVALID, "kk_csc_create: bad dimensions");
    KK_CHECK(index_base == 0 || index_base == 1, KK_ERR_INVALID, "index_base must be 0 or 1");
    KK_TRY(check_ptr_array("kk_csc_create", colptr, ncols, nnz, index_base));
    KK_CHECK(nrows < (int64_t)1 << 31 && ncols < (int64_t)1 << 31, KK_ERR_UNSUPPORTED,
             "kk_csc_create: dimensions >= 2^31 are not supported (int32 row indices on the device)");
    KK_HIP(hipSetDevice(c->device));
    // the CSC arrays of A are the CSR arrays of A'
    kk_host_csr ht;
    ht.nrows = ncols; ht.ncols = nrows;
    ht.rowptr.resize(ncols + 1);
    for (int64_t i = 0; i <= ncols; ++i) ht.rowptr[i] = colptr[i] - index_base;
    ht.col.resize(nnz); ht.val.assign(nzval, nzval + nnz);
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t r = rowval[p] - index_base;
        if (r < 0 || r >= nrows) {
            kk_set_error("kk_csc_create: row index %lld out of range at entry %lld", (long long)r, (long long)p);
            return KK_ERR_DIM;
        }
        ht.col[p] = (int32_t)r;
    }
    kk_op op = new kk_op_s();
    op->ctx = c; op->nrows = nrows; op->ncols = ncols; op->nnz = nnz; op->flags = flags;
    int s;
    if (flags & KK_OP_SYMMETRIC) {
        s = upload_sparse(c, ht, op->A);  // A == A'
    } else {
        transpose_csr(ht, op->hA);
        s = upload_sparse(c, op->hA, op->A);
        if (s == KK_OK) {
            s = upload_sparse(c, ht, op->At);
            op->have_At = (s == KK_OK);
            kk_host_csr().rowptr.swap(op->hA.rowptr); op->hA.col.clear(); op->hA.val.clear();
        }
    }
    if (s != KK_OK) { free_sparse(op->A); free_sparse(op->At); delete op; return s; }
    *out = op;
    return KK_OK;
}

KK_API int kk_op_free(kk_op op) {
    if (!op) return KK_OK;
    (void)hipDeviceSynchronize();  // see kk_basis_free
    if (kk_halo_plan* p = op->A.plan) {
        (void)hipFree(p->d_send_idx); (void)hipFree(p->d_sendbuf); (void)hipFree(p->d_ghost);
        (void)hipFree(p->d_sendbuf_blk); (void)hipFree(p->d_ghost_blk);
        delete p;
        op->A.plan = nullptr;
    }
    if (kk_gather_plan* g = op->gather) {
        (void)hipFree(g->vfull); (void)hipFree(g->zfull); (void)hipFree(g->stage);
        delete g;
    }
    free_sparse(op->A);
    free_sparse(op->At);
    delete op;
    return KK_OK;
}

// ------------------------------------------------------------------------------------------
// row-sharded operators with a native (RCCL) exchange plan
// ------------------------------------------------------------------------------------------
static void comm_shape(kk_ctx c, int* rank, int* world) {
    *rank = c->comm ? c->comm->rank : 0;
    *world = c->comm ? c->comm->world : 1;
}

// Row block [row_offsets[rank], row_offsets[rank+1]) of a square global operator with GLOBAL column indices.  Columns
// owned by other ranks become ghost columns; the request lists are exchanged here (all-gather of the counts, grouped
// send / recv of the indices) so that the caller needs no communication code of its own.
KK_API int kk_csr_create_sharded(kk_ctx c, int64_t nrows_local, const int64_t* row_offsets, int64_t nnz,
                                 const int64_t* rowptr, const int64_t* colind, const double* val, int index_base,
                                 int flags, kk_op* out) {
    KK_CHECK(c && out && row_offsets && rowptr && (nnz == 0 || (colind && val)), KK_ERR_INVALID, "kk_csr_create_sharded: null arg");
    int rank, world;
    comm_shape(c, &rank, &world);
    KK_HIP(hipSetDevice(c->device));
    // Loop-back mode for one-GPU test boxes (KK_LOOPBACK_GHOST_FROM=r, world size 1 only): the columns >= r, although
    // owned by this rank, are routed through the ghost machinery (request list, gather into the send buffer, exchange
    // -- here a device copy to self --, ghost-indexed reads in the SpMV / SpMM kernels), so that every piece of the
    // multi-GPU data path except the wire itself runs where no second GPU exists.
    int64_t loop_from = -1, loop_below = 0;   // KK_LOOPBACK_GHOST_BELOW = r2: the columns < r2 as well (a middle rank's layout)
    int64_t lo = 0, hi = 0, n_global = 0, n_ghost = 0;
    std::vector<int64_t> needed, recv_counts(world, 0), send_counts(world, 0);
    kk_op op = nullptr;
    // ---- part 1, LOCAL: validation, renumbering, upload.  This is a collective entry point: a rank that returned from
    // here on bad input would leave its peers blocked in the all-gather below, so the local status is agreed on first.
    auto local_part = [&]() -> int {
        KK_CHECK(index_base == 0 || index_base == 1, KK_ERR_INVALID, "index_base must be 0 or 1");
        for (int q = 0; q < world; ++q)
            KK_CHECK(row_offsets[q] <= row_offsets[q + 1], KK_ERR_DIM, "kk_csr_create_sharded: row_offsets must be non-decreasing");
        lo = row_offsets[rank]; hi = row_offsets[rank + 1]; n_global = row_offsets[world];
        KK_CHECK(row_offsets[0] == 0 && hi - lo == nrows_local && nrows_local > 0, KK_ERR_DIM,
                 "kk_csr_create_sharded: rank %d owns rows [%lld,%lld) but nrows_local = %lld", rank, (long long)lo, (long long)hi,
                 (long long)nrows_local);
        KK_TRY(check_ptr_array("kk_csr_create_sharded", rowptr, nrows_local, nnz, index_base));
        if (world == 1) {
            const char* lb = getenv("KK_LOOPBACK_GHOST_FROM");
            if (lb && *lb) loop_from = atoll(lb);
            const char* lb2 = getenv("KK_LOOPBACK_GHOST_BELOW");
            if (lb2 && *lb2) { loop_below = atoll(lb2); if (loop_from < 0) loop_from = n_global; }
        }
        auto is_ghost = [&](int64_t g) { return g < lo || g >= hi || (loop_from >= 0 && (g >= loop_from || g < loop_below)); };
        // ghost columns: sorted unique global ids outside [lo, hi) -> grouped by owner
        for (int64_t p = 0; p < nnz; ++p) {
            const int64_t g = colind[p] - index_base;
            KK_CHECK(g >= 0 && g < n_global, KK_ERR_DIM, "kk_csr_create_sharded: column index %lld out of range at entry %lld",
                     (long long)g, (long long)p);
            if (is_ghost(g)) needed.push_back(g);
        }
        std::sort(needed.begin(), needed.end());
        needed.erase(std::unique(needed.begin(), needed.end()), needed.end());
        n_ghost = (int64_t)needed.size();
        KK_CHECK(nrows_local + n_ghost < (int64_t)1 << 31, KK_ERR_UNSUPPORTED,
                 "kk_csr_create_sharded: local block too large for int32 indices");
        for (int64_t g : needed) {
            const int q = (int)(std::upper_bound(row_offsets, row_offsets + world + 1, g) - row_offsets) - 1;
            recv_counts[q]++;
        }
        op = new kk_op_s();
        op->ctx = c; op->nrows = nrows_local; op->ncols = nrows_local + n_ghost; op->nnz = nnz; op->flags = flags;
        kk_host_csr h;
        h.nrows = nrows_local; h.ncols = nrows_local + n_ghost;
        h.rowptr.resize(nrows_local + 1);
        for (int64_t i = 0; i <= nrows_local; ++i) h.rowptr[i] = rowptr[i] - index_base;
        h.col.resize(nnz); h.val.assign(val, val + nnz);
        for (int64_t p = 0; p < nnz; ++p) {
            const int64_t g = colind[p] - index_base;
            h.col[p] = !is_ghost(g) ? (int32_t)(g - lo)
                                    : (int32_t)(nrows_local + (std::lower_bound(needed.begin(), needed.end(), g) - needed.begin()));
        }
        KK_TRY(upload_sparse(c, h, op->A));
        if (n_ghost > 0) KK_TRY(detect_stencil_sharded(h, nrows_local, op->A));
        kk_halo_plan* plan = new kk_halo_plan();
        op->A.plan = plan;
        hipError_t e = hipMalloc(&plan->d_ghost, std::max<int64_t>(n_ghost, 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(plan->d_ghost, 0, std::max<int64_t>(n_ghost, 1) * sizeof(double), c->stream);
        if (e != hipSuccess) return kk_hip_fail(e, "hipMalloc (ghost buffer)", __FILE__, __LINE__);
        return KK_OK;
    };
    int s = local_part();
    auto fail = [&](int st) { if (op) kk_op_free(op); return st; };
    {
        int worst = s;
        const int sa = kk_comm_agree_status(c, s, &worst);
        if (sa != KK_OK) return fail(sa);
        if (s != KK_OK) return fail(s);
        if (worst != KK_OK) {
            kk_set_error("kk_csr_create_sharded: another rank rejected its arguments (status %d); no operator was created on any rank", worst);
            return fail(worst);
        }
    }
    kk_halo_plan* plan = op->A.plan;
    int64_t *d_a = nullptr, *d_b = nullptr;
    if (world > 1) {
        // counts[r * world + q] = number of entries rank r needs from rank q
        hipError_t e1 = hipMalloc(&d_a, world * sizeof(int64_t)), e2 = hipMalloc(&d_b, (size_t)world * world * sizeof(int64_t));
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(d_a); (void)hipFree(d_b); return fail(kk_hip_fail(e1 != hipSuccess ? e1 : e2, "hipMalloc", __FILE__, __LINE__)); }
        std::vector<int64_t> all((size_t)world * world);
        (void)hipMemcpyAsync(d_a, recv_counts.data(), world * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
        s = kk_comm_allgather_i64(c, d_a, d_b, world);
        if (s == KK_OK && hipMemcpyAsync(all.data(), d_b, all.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess) s = KK_ERR_HIP;
        if (s == KK_OK && hipStreamSynchronize(c->stream) != hipSuccess) s = KK_ERR_HIP;
        (void)hipFree(d_a); (void)hipFree(d_b);
        if (s != KK_OK) return fail(s);
        for (int q = 0; q < world; ++q) send_counts[q] = all[(size_t)q * world + rank];
        send_counts[rank] = 0;
    } else if (loop_from >= 0) {
        send_counts[0] = recv_counts[0];   // loop-back: this rank serves its own requests
    }
    plan->send_counts = send_counts; plan->recv_counts = recv_counts;
    for (int q = 0; q < world; ++q) { plan->total_send += send_counts[q]; plan->total_recv += recv_counts[q]; }
    {
        hipError_t e = hipMalloc(&plan->d_sendbuf, std::max<int64_t>(plan->total_send, 1) * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&plan->d_send_idx, std::max<int64_t>(plan->total_send, 1) * sizeof(int64_t));
        // (an allocation failure here is fatal for the run either way: the peers are already committed to the exchange below)
        if (e != hipSuccess) return fail(kk_hip_fail(e, "hipMalloc (ghost plan)", __FILE__, __LINE__));
    }
    if (world > 1) {
        // my request list (global ids grouped by owner) goes out, the peers' request lists come in -> local send indices
        int64_t* d_req = nullptr;
        hipError_t e = hipMalloc(&d_req, std::max<int64_t>(n_ghost, 1) * sizeof(int64_t));
        if (e != hipSuccess) return fail(kk_hip_fail(e, "hipMalloc", __FILE__, __LINE__));
        (void)hipMemcpyAsync(d_req, needed.data(), n_ghost * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
        s = kk_comm_exchange_i64(c, d_req, recv_counts.data(), plan->d_send_idx, send_counts.data());
        std::vector<int64_t> idx((size_t)plan->total_send);
        if (s == KK_OK && plan->total_send &&
            hipMemcpyAsync(idx.data(), plan->d_send_idx, idx.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess) s = KK_ERR_HIP;
        if (s == KK_OK && hipStreamSynchronize(c->stream) != hipSuccess) s = KK_ERR_HIP;
        (void)hipFree(d_req);
        if (s != KK_OK) return fail(s);
        for (int64_t& g : idx) {
            if (g < lo || g >= hi) { kk_set_error("kk_csr_create_sharded: a peer requested row %lld which rank %d does not own", (long long)g, rank); return fail(KK_ERR_DIM); }
            g -= lo;
        }
        if (plan->total_send) (void)hipMemcpyAsync(plan->d_send_idx, idx.data(), idx.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(KK_ERR_HIP);
    }
    if (world == 1 && loop_from >= 0 && plan->total_send) {   // loop-back: the request list is my own `needed`
        std::vector<int64_t> idx(needed);
        for (int64_t& g : idx) g -= lo;
        if (hipMemcpy(plan->d_send_idx, idx.data(), idx.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) return fail(KK_ERR_HIP);
    }
    op->A.n_local = nrows_local;
    op->A.n_ghost = n_ghost;
    op->A.ghost = plan->d_ghost;
    op->n_local_cols = nrows_local; op->n_ghost = n_ghost;
    *out = op;
    return KK_OK;
}

// Row block of a rectangular map A (nrows_global x ncols_global) for the sharded GKL.  The
// long vectors (rows of A) are sharded by this rank's rows, the short ones (columns of A) evenly with stride
// shard = ceil(ncols_global / world): rank r owns columns [r*shard, min((r+1)*shard, ncols_global)).
//   A x : all-gather of the short vector, local SpMV on the gathered buffer
//   A'x : local transposed SpMV (full-length partial), reduce-scatter (sum) onto the shards
KK_API int kk_csr_create_sharded_rect(kk_ctx c, int64_t nrows_local, int64_t ncols_global, int64_t nnz, const int64_t* rowptr,
                                      const int64_t* colind, const double* val, int index_base, kk_op* out,
                                      int64_t* ncols_local) {
    KK_CHECK(c && out && rowptr && (nnz == 0 || (colind && val)), KK_ERR_INVALID, "kk_csr_create_sharded_rect: null arg");
    KK_CHECK(nrows_local > 0 && ncols_global > 0 && nnz >= 0, KK_ERR_INVALID, "kk_csr_create_sharded_rect: bad dimensions");
    KK_CHECK(index_base == 0 || index_base == 1, KK_ERR_INVALID, "index_base must be 0 or 1");
    KK_TRY(check_ptr_array("kk_csr_create_sharded_rect", rowptr, nrows_local, nnz, index_base));
    int rank, world;
    comm_shape(c, &rank, &world);
    const int64_t shard = (ncols_global + world - 1) / world;
    const int64_t full = shard * world;
    const int64_t n_loc = std::max<int64_t>(0, std::min(shard, ncols_global - rank * shard));
    KK_CHECK(n_loc > 0, KK_ERR_DIM, "kk_csr_create_sharded_rect: rank %d owns no columns (%lld columns over %d ranks)", rank,
             (long long)ncols_global, world);
    KK_CHECK(full < (int64_t)1 << 31 && nrows_local < (int64_t)1 << 31, KK_ERR_UNSUPPORTED,
             "kk_csr_create_sharded_rect: local block too large for int32 indices");
    KK_HIP(hipSetDevice(c->device));
    kk_op op = new kk_op_s();
    op->ctx = c; op->nrows = nrows_local; op->ncols = full; op->nnz = nnz; op->flags = 0;
    kk_host_csr& h = op->hA;   // kept: the transposed image is built from it on first use
    h.nrows = nrows_local; h.ncols = full;
    h.rowptr.resize(nrows_local + 1);
    for (int64_t i = 0; i <= nrows_local; ++i) h.rowptr[i] = rowptr[i] - index_base;
    h.col.resize(nnz); h.val.assign(val, val + nnz);
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t g = colind[p] - index_base;
        if (g < 0 || g >= ncols_global) {
            delete op;
            kk_set_error("kk_csr_create_sharded_rect: column index %lld out of range at entry %lld", (long long)g, (long long)p);
            return KK_ERR_DIM;
        }
        h.col[p] = (int32_t)g;
    }
    int s = upload_sparse(c, h, op->A);
    if (s != KK_OK) { free_sparse(op->A); delete op; return s; }
    kk_gather_plan* g = new kk_gather_plan();
    op->gather = g;
    g->ncols_global = ncols_global; g->shard = shard; g->n_local = n_loc;
    const int64_t pad = (full + KK_SUB - 1) / KK_SUB * KK_SUB + KK_SUB;   // the SpMV kernels may store whole row tiles
    hipError_t e = hipMalloc(&g->vfull, pad * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&g->zfull, pad * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&g->stage, (shard + KK_SUB) * sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(g->vfull, 0, pad * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(g->zfull, 0, pad * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(g->stage, 0, (shard + KK_SUB) * sizeof(double), c->stream);
    if (e != hipSuccess) { kk_op_free(op); return kk_hip_fail(e, "hipMalloc (gather plan)", __FILE__, __LINE__); }
    // ghost-only operator: every column is read from the gathered buffer
    op->A.n_local = 0;
    op->A.n_ghost = full;
    op->A.ghost = g->vfull;
    op->n_local_cols = 0; op->n_ghost = full;
    if (ncols_local) *ncols_local = n_loc;
    *out = op;
    return KK_OK;
}

// y = A x (x: this rank's shard of a short vector, y: this rank's rows) or y = A' x (x: this rank's rows, y: shard)
int rect_apply(kk_op op, int transpose, const double* x, double* y) {
    kk_ctx c = op->ctx;
    kk_gather_plan* g = op->gather;
    kk_spmv_fuse f;
    if (!transpose) {
        KK_HIP(hipMemcpyAsync(g->stage, x, g->n_local * sizeof(double), hipMemcpyDeviceToDevice, c->stream));  // tail stays zero
        KK_TRY(kk_comm_allgather_f64(c, g->stage, g->vfull, g->shard));
        return kk_launch_spmv(c, op->A, x, y, 0, f);
    }
    const kk_sparse_dev* At;
    KK_TRY(get_matrix(op, 1, &At));
    KK_TRY(kk_launch_spmv(c, *At, x, g->zfull, 0, f));
    KK_TRY(kk_comm_reducescatter_f64(c, g->zfull, g->stage, g->shard));
    KK_HIP(hipMemcpyAsync(y, g->stage, g->n_local * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return KK_OK;
}

KK_API int kk_op_info(kk_op op, int64_t* nrows, int64_t* ncols, int64_t* nnz, int* format, int64_t* bytes) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    if (nrows) *nrows = op->nrows;
    if (ncols) *ncols = op->ncols;
    if (nnz) *nnz = op->nnz;
    if (format) *format = (op->A.format == 0 && op->A.dia_D > 0) ? (op->A.dia_const ? 5 : 4) : op->A.format;
    if (bytes) *bytes = op->A.bytes + op->At.bytes;
    return KK_OK;
}

KK_API int kk_op_set_ghost(kk_op op, int64_t n_local_cols, int64_t n_ghost, void* device_ghost) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    KK_CHECK(n_local_cols >= 0 && n_ghost >= 0 && n_local_cols + n_ghost == op->ncols, KK_ERR_DIM,
             "kk_op_set_ghost: n_local (%lld) + n_ghost (%lld) != ncols (%lld)", (long long)n_local_cols,
             (long long)n_ghost, (long long)op->ncols);
    KK_CHECK(n_ghost == 0 || device_ghost, KK_ERR_INVALID, "kk_op_set_ghost: null ghost buffer");
    op->A.n_local = n_local_cols;
    op->A.n_ghost = n_ghost;
    op->A.ghost = (double*)device_ghost;  // caller-owned
    return KK_OK;
}

KK_API int kk_op_set_halo_hook(kk_op op, kk_halo_fn fn, void* user) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    op->A.halo = fn;
    op->A.halo_user = user;
    return KK_OK;
}
KK_API int kk_gather_ptr(kk_ctx c, const void* x_device, const int64_t* device_idx, int64_t count, void* device_out) {
    KK_CHECK(c && x_device && (count == 0 || (device_idx && device_out)), KK_ERR_INVALID, "kk_gather_ptr: null arg");
    return kk_launch_gather(c, (const double*)x_device, device_idx, count, (double*)device_out);
}

int get_matrix(kk_op op, int transpose, const kk_sparse_dev** M) {
    if (!transpose || (op->flags & KK_OP_SYMMETRIC)) {
        *M = &op->A;
        return KK_OK;
    }
    if (!op->have_At) {
        KK_CHECK(!op->hA.rowptr.empty(), KK_ERR_INVALID, "transpose requested but host copy is gone");
        kk_host_csr ht;
        transpose_csr(op->hA, ht);
        KK_TRY(upload_sparse(op->ctx, ht, op->At));
        op->have_At = true;
        kk_host_csr().rowptr.swap(op->hA.rowptr); op->hA.col.clear(); op->hA.col.shrink_to_fit(); op->hA.val.clear(); op->hA.val.shrink_to_fit();
    }
    *M = &op->At;
    return KK_OK;
}

// dimension check of y = op(A) x ; with ghosts the x-vector holds the local columns only
int check_apply(kk_op op, int transpose, kk_basis bx, kk_basis by) {
    if (op->gather) {
        const int64_t inr = transpose ? op->nrows : op->gather->n_local, outr = transpose ? op->gather->n_local : op->nrows;
        KK_CHECK(bx->n == inr && by->n == outr, KK_ERR_DIM, "apply: sharded map has %lld local rows / %lld local columns%s, x has %lld rows, y has %lld rows",
                 (long long)op->nrows, (long long)op->gather->n_local, transpose ? " (adjoint)" : "", (long long)bx->n, (long long)by->n);
        KK_CHECK(bx->ctx == op->ctx && by->ctx == op->ctx, KK_ERR_INVALID, "apply: objects belong to different contexts");
        return KK_OK;
    }
    const int64_t in = transpose ? op->nrows : (op->A.n_ghost > 0 ? op->A.n_local : op->ncols);
    const int64_t outn = transpose ? op->ncols : op->nrows;
    // ghost-only operator (n_local == 0): every column comes from the caller's gathered buffer, x is unused
    const bool ghost_only = !transpose && op->A.n_ghost > 0 && op->A.n_local == 0;
    KK_CHECK((ghost_only || bx->n == in) && by->n == outn, KK_ERR_DIM, "apply: operator is %lldx%lld%s, x has %lld rows, y has %lld rows",
             (long long)op->nrows, (long long)op->ncols, transpose ? " (adjoint)" : "", (long long)bx->n, (long long)by->n);
    KK_CHECK(bx->ctx == op->ctx && by->ctx == op->ctx, KK_ERR_INVALID, "apply: objects belong to different contexts");
    return KK_OK;
}

KK_API int kk_spmv(kk_op op, int transpose, kk_basis bx, int cx, kk_basis by, int cy) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    CHECK_COL(bx, cx); CHECK_COL(by, cy);
    KK_TRY(check_apply(op, transpose, bx, by));
    KK_CHECK(!(bx == by && cx == cy), KK_ERR_INVALID, "kk_spmv: x and y must differ");
    if (op->gather) {
        gram_touch(by, cy);
        return rect_apply(op, transpose, bx->col(cx), by->col(cy));
    }
    const kk_sparse_dev* M;
    KK_TRY(get_matrix(op, transpose, &M));
    gram_touch(by, cy);
    kk_spmv_fuse f;
    return kk_launch_spmv(op->ctx, *M, bx->col(cx), by->col(cy), by->ld, f);
}

KK_API int kk_spmv_affine(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, double a0, double a1) {
    KK_CHECK(op, KK_ERR_INVALID, "null op");
    CHECK_COL(bx, cx); CHECK_COL(by, cy);
    KK_TRY(check_apply(op, 0, bx, by));
    KK_CHECK(!op->gather && (op->nrows == op->ncols || op->A.n_ghost > 0), KK_ERR_DIM, "affine apply needs a square operator");
    KK_CHECK(!(bx == by && cx == cy), KK_ERR_INVALID, "kk_spmv_affine: x and y must differ");
    gram_touch(by, cy);
    kk_spmv_fuse f;
    f.a0 = a0; f.a1 = a1;
    return kk_launch_spmv(op->ctx, op->A, bx->col(cx), by->col(cy), by->ld, f);
}

// q = a0 p + a1 A p with the fused <p, q> (the CG / short-recurrence apply, linsolve/cg.jl:35-36,61-62)
KK_API int kk_spmv_affine_dot(kk_op op, kk_basis bx, int cx, kk_basis by, int cy, double a0, double a1, double* dot) {
    KK_CHECK(op && dot, KK_ERR_INVALID, "null arg");
    CHECK_COL(bx, cx); CHECK_COL(by, cy);
    KK_TRY(check_apply(op, 0, bx, by));
    KK_CHECK(!(bx == by && cx == cy), KK_ERR_INVALID, "kk_spmv_affine_dot: x and y must differ");
    kk_ctx c = op->ctx;
    gram_touch(by, cy);
    kk_spmv_fuse f;
    f.a0 = a0; f.a1 = a1;
    f.dot_mode = 2;
    f.dot_out = SCP(c, SC_DOT);
    KK_TRY(kk_launch_spmv(c, op->A, bx->col(cx), by->col(cy), by->ld, f));
    KK_TRY(ws_fetch_async(c, WS_SCAL + SC_DOT, 1, 0));
    KK_TRY(stream_sync(c));
    *dot = pin(c, WS_SCAL + SC_DOT)[0];
    return KK_OK;
}
KK_API int kk_gather(kk_basis bx, int cx, const int64_t* device_idx, int64_t count, void* device_out) {
    CHECK_COL_RO(bx, cx);
    return kk_launch_gather(bx->ctx, bx->col(cx), device_idx, count, (double*)device_out);
}

