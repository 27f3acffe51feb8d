"""CPU emulation of k_spmm_dia_al (csrc/kk_kernels_spmv.hip): the aligned form of the sweeping multi-column apply of a value-free
5-point grid stencil (apply(f, ::Block) of blocklanczos.jl:242-263 for the operator of BASELINE config 5).

What is emulated, wave by wave and lane by lane, in the kernel's own terms:
  * a wave = (strip of 128 positions of a grid line) x (sweep of `lines` grid lines); lane l owns positions 2l, 2l + 1;
  * window loads through a bounds-checked descriptor of `nrows` doubles: a pair whose 32-bit byte offset is out of range reads as
    zero, offsets of lanes that own nothing / of lines beyond the sweep's halo are replaced by an out-of-range constant, a NEGATIVE
    row wraps to a huge unsigned offset (= out of range);
  * the strip's edge elements: lane 0 loads position p0 - 1 (strips > 0), lane 63 position p0 + 128 (if inside the line);
  * +-1 neighbours = wave shifts of the centre pair with the edge element as the `old` operand (lane 0 / lane 63 keep it);
  * coefficient masks: no -1 entry at position 0, no +1 entry at position D - 1;
  * the four-line window rotation (A, B, C, E) with the pair fetch two lines ahead and the edge fetch one line ahead;
  * stores switched off outside [row_lo, row_hi) and for lanes beyond the line.
The emulation must reproduce SciPy's A @ x for line lengths around the strip width, every sweep length and a row window."""
import numpy as np
import pytest
import scipy.sparse as sp


def stencil(D, T, c):
    """rows = T lines of length D; offsets -D, -1, 0, +1, +D with coefficients c[0..4] (no wrap inside a line)"""
    n = D * T
    i = np.arange(n)
    ix = i % D
    rows, cols, vals = [], [], []
    for q, off in enumerate((-D, -1, 0, 1, D)):
        ok = (i + off >= 0) & (i + off < n)
        if off == -1:
            ok &= ix > 0
        if off == 1:
            ok &= ix < D - 1
        rows.append(i[ok]); cols.append(i[ok] + off); vals.append(np.full(ok.sum(), c[q]))
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


def emulate(x, D, T, c, lines, row_lo, row_hi):
    nrows = D * T
    nbytes = nrows * 8
    OFF_NONE = 0xFFFFFFF0
    lane = np.arange(64)

    def load_pair(off):            # raw_buffer_load_b128: 16 bytes at byte offset `off` (uint32 per lane), zero when out of range
        off = off.astype(np.uint64) & np.uint64(0xFFFFFFFF)
        ok = off + np.uint64(16) <= np.uint64(nbytes)
        idx = np.where(ok, off // np.uint64(8), 0).astype(np.int64)
        return np.where(ok, x[idx], 0.0), np.where(ok, x[np.minimum(idx + 1, nrows - 1)], 0.0)

    def load_one(off):             # raw_buffer_load_b64
        off = off.astype(np.uint64) & np.uint64(0xFFFFFFFF)
        ok = off + np.uint64(8) <= np.uint64(nbytes)
        idx = np.where(ok, off // np.uint64(8), 0).astype(np.int64)
        return np.where(ok, x[idx], 0.0)

    y = np.full(nrows, np.nan)
    stored = np.zeros(nrows, dtype=int)
    strips = (D + 127) // 128
    Tlo, Thi = row_lo // D, (row_hi + D - 1) // D
    nwaves = strips * ((Thi - Tlo + lines - 1) // lines)
    for wv in range(nwaves):
        strip = wv % strips
        t0 = Tlo + (wv // strips) * lines
        if t0 >= Thi:
            continue
        t1 = min(t0 + lines, Thi)
        p0 = strip * 128
        p = p0 + 2 * lane
        own = p < D
        pe = np.where(lane == 0, p0 - 1, p0 + 128)
        eown = ((lane == 0) & (strip > 0)) | ((lane == 63) & (pe < D))

        def fetch_pairs(t):
            off = np.where((t <= t1) & own, ((t * D + p) * 8) % (1 << 32), OFF_NONE)
            return load_pair(off)

        def fetch_edges(t):
            off = np.where((t < t1) & eown, ((t * D + pe) * 8) % (1 << 32), OFF_NONE)
            return load_one(off)

        cW0 = np.where(p == 0, 0.0, c[1]); cE1 = np.where(p + 2 == D, 0.0, c[3])

        def line(Lm, L0, Lp, t):
            (mx, my), (x0, y0, e0), (px, py) = Lm[:2], L0, Lp[:2]
            left = np.concatenate([[e0[0]], y0[:-1]])        # wave_shr:1, lane 0 keeps `old` = its edge element
            right = np.concatenate([x0[1:], [e0[63]]])       # wave_shl:1, lane 63 keeps `old`
            ax = c[0] * mx + cW0 * left + c[2] * x0 + c[3] * y0 + c[4] * px
            ay = c[0] * my + c[1] * x0 + c[2] * y0 + cE1 * right + c[4] * py
            r = t * D + p
            st = own & (r >= row_lo) & (r < row_hi)
            y[r[st]] = ax[st]; y[r[st] + 1] = ay[st]
            stored[r[st]] += 1; stored[r[st] + 1] += 1

        # window slots hold (x, y, edge); the edge of a slot is fetched one line later than its pairs
        W = {}
        W["A"] = (*fetch_pairs(t0 - 1), None)
        W["B"] = (*fetch_pairs(t0), fetch_edges(t0))
        W["C"] = (*fetch_pairs(t0 + 1), None)
        order = ["A", "B", "C", "E"]
        t = t0
        while True:
            m, z, pl, nx = order
            W[nx] = (*fetch_pairs(t + 2), None)
            W[pl] = (W[pl][0], W[pl][1], fetch_edges(t + 1))
            line(W[m], W[z], W[pl], t)
            t += 1
            if t >= t1:
                break
            order = [z, pl, nx, m]
    return y, stored


@pytest.mark.parametrize("D,T", [(64, 9), (128, 7), (130, 6), (254, 5), (256, 5), (300, 4), (2, 40), (126, 11)])
def test_aligned_sweep_reproduces_the_stencil(D, T):
    rng = np.random.default_rng(D * 31 + T)
    c = np.array([-1.25, -1.5, 4.0, -0.5, -0.75])
    x = rng.standard_normal(D * T)
    A = stencil(D, T, c)
    ref = A @ x
    for lines in (1, 2, 3, 4, 5, 16):
        y, stored = emulate(x, D, T, c, lines, 0, D * T)
        assert np.all(stored == 1), (D, T, lines, "every row must be stored exactly once")
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-13, err_msg=f"D={D} T={T} lines={lines}")


def test_row_window_of_a_sharded_interior():
    D, T = 130, 9
    rng = np.random.default_rng(5)
    c = np.array([-1.0, -2.0, 4.0, -3.0, -0.5])
    x = rng.standard_normal(D * T)
    ref = stencil(D, T, c) @ x
    lo, hi = 2 * D, 7 * D - 4           # even bounds (the launcher sends odd ones to the 8-byte form)
    for lines in (1, 4, 6):
        y, stored = emulate(x, D, T, c, lines, lo, hi)
        assert np.all(stored[lo:hi] == 1) and stored[:lo].sum() == 0 and stored[hi:].sum() == 0
        np.testing.assert_allclose(y[lo:hi], ref[lo:hi], rtol=0, atol=1e-13)
