"""CPU emulation of the lane map of k_block_gram2p's whole-line loads (csrc/kk_kernels_block.hip: g2_load_tile / g2_line_fix).

Lane l = c8 + 8 b3 + 16 kq of a wave loads, for a 16-row group of a 16-column stream,
    a = 16 B of column c8     at row pair h = l >> 3 (rows 2h, 2h + 1)         [instruction covers 8 columns x 128 B]
    b = 16 B of column c8 + 8 at the same row pair                             [the next instruction: the other 8 columns]
and must end up, for v_mfma_f64_16x16x4 (operand layout: lane l supplies column l & 15 at contraction slot l >> 4), with column
(l & 15) at the four rows 4 kq .. 4 kq + 3.  The exchange: lanes l and l ^ 8 (same 16-lane row) swap one value,
    first = b3 ? partner's b : own a,   second = b3 ? own b : partner's a      (row_ror:8 DPP moves under a bank mask)."""
import numpy as np


def test_whole_line_loads_restore_the_mfma_operand_layout():
    lanes = np.arange(64)
    c8, b3, kq = lanes & 7, (lanes >> 3) & 1, lanes >> 4
    h = lanes >> 3
    # each loaded value is tagged (column, first row of its pair)
    a = np.stack([c8, 2 * h], 1)
    b = np.stack([c8 + 8, 2 * h], 1)
    partner = lanes ^ 8
    assert np.all((partner >> 4) == (lanes >> 4))                 # row_ror:8 stays inside a 16-lane row
    first = np.where(b3[:, None] == 1, b[partner], a)
    second = np.where(b3[:, None] == 1, b, a[partner])
    col = lanes & 15
    assert np.all(first[:, 0] == col) and np.all(second[:, 0] == col)
    assert np.all(first[:, 1] == 4 * kq) and np.all(second[:, 1] == 4 * kq + 2)   # rows 4kq, 4kq+1 | 4kq+2, 4kq+3
    # every (column, row) of the 16 x 16 tile is held exactly once
    held = {(int(c), int(r0) + e) for (c, r0) in np.vstack([first, second]) for e in (0, 1)}
    assert held == {(c, r) for c in range(16) for r in range(16)}
    # and one load instruction touches whole 128-byte lines: 8 lanes of a column cover 16 consecutive rows
    for c in range(8):
        rows = sorted(int(2 * hh) + e for hh in h[c8 == c] for e in (0, 1))
        assert rows == list(range(16))
