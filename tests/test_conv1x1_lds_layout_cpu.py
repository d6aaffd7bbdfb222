"""Host replay of the LDS layout of csrc/conv1x1_split3.hip (round 6): the float32 token stage (rows of 128 B, eight 16-byte slots), written by LDS-DMA
pieces of 8 rows x 128 B (lane-linear, so the swizzle is applied on the SOURCE side) and read as B fragments of v_mfma_f32_16x16x32_bf16 -- lane (r, g)
needs k = 8 g .. 8 g + 7 of token row r as two float4.  Checked with the lane-group rule of MI355X_MICROARCH.md (a ds_read_b128 is serviced in the four
16-lane groups below over 64 four-byte banks): (1) the DMA's lane -> (row, slot) map and the fragment read's address expression agree on where every
float4 of the tile lives, (2) both reads of every group touch 16 distinct 16-byte positions of the 256-byte bank window.  The first layout of the kernel
(natural slot u = 2 g + h at u ^ ((r >> 1) & 7)) fails (2): the lane groups pair even-g lanes of rows {0-3, 12-15} with odd-g lanes of rows {4-11}, whose
slot indices differ by XOR 2 -- exactly the XOR distance of their row terms.  Test infrastructure: mirrors xoff / x_off of the kernel."""
import pytest

READ_GROUPS = ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
               [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63])


def sigma(u, fixed):
    """stored (logical) slot of the float4 holding k = 4 u .. 4 u + 3"""
    return ((u >> 1) + 4 * (u & 1)) if fixed else u


def sigma_inv(s, fixed):
    return (((s & 3) << 1) | (s >> 2)) if fixed else s


def dma_piece(pc, lane, fixed):
    """-> (stage byte address written by the lane, token row, source byte offset inside the row's 128-byte chunk) -- xrow / xoff of the kernel"""
    row = pc * 8 + (lane >> 3)
    phys = lane & 7
    u = sigma_inv(phys ^ ((row >> 1) & 7), fixed)
    return pc * 1024 + lane * 16, row, u * 16


def frag_addr(wm, fm, lane, half, fixed):
    """byte address of the float4 with k = 8 g + 4 half .. of row wm * 32 + fm * 16 + r -- x_off of the kernel"""
    r, g = lane & 15, lane >> 4
    row = wm * 32 + fm * 16 + r
    return row * 128 + ((sigma(2 * g + half, fixed) ^ ((r >> 1) & 7)) << 4), row, (2 * g + half) * 16


@pytest.mark.parametrize("bm", [64, 128])
def test_dma_pieces_and_fragment_reads_agree(bm):
    where = {}
    for pc in range(bm // 8):
        for lane in range(64):
            a, row, src = dma_piece(pc, lane, True)
            assert a not in where
            where[a] = (row, src)
    assert len(where) == bm * 8                                   # every 16-byte slot of the stage written exactly once
    for wm in range(bm // 32):
        for fm in range(2):
            for lane in range(64):
                for half in range(2):
                    a, row, src = frag_addr(wm, fm, lane, half, True)
                    assert where[a] == (row, src), (wm, fm, lane, half)


def _worst(fixed):
    worst = 0
    for fm in range(2):
        for half in range(2):
            for grp in READ_GROUPS:
                seen = {}
                for lane in grp:
                    a, _, _ = frag_addr(0, fm, lane, half, fixed)
                    seen.setdefault((a % 256) // 16, set()).add(a)
                worst = max(worst, max(len(v) for v in seen.values()))
    return worst


def test_fragment_reads_are_conflict_free():
    assert _worst(fixed=False) == 2      # the first layout: two lanes of every group on the same 16-byte position
    assert _worst(fixed=True) == 1


def test_weight_fragment_reads_are_conflict_free():
    """weight stage = gemm_split3.hip's: 64-byte rows, slot g of row r at g ^ ((r >> 1) & 3)"""
    for grp in READ_GROUPS:
        pos = set()
        for lane in grp:
            r, g = lane & 15, lane >> 4
            a = r * 64 + ((g ^ ((r >> 1) & 3)) << 4)
            pos.add((a % 256) // 16)
        assert len(pos) == 16
