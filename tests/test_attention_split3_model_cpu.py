"""Lane-level numpy model of csrc/vit.hip vit_attention_split3_kernel's index arithmetic (test infrastructure; the kernel itself is checked
on the GPU by tests/op_checks.py vit_attention_split3): the LDS images of one 64-key tile -- K rows with the slot swizzle, V^T rows with the
key permutation inside each 32-key block and the slot swizzle -- the fragment reads of S^T = K.Q^T and O^T = V^T.P^T, the lane layout of the
v_mfma_f32_16x16x32_bf16 operands and accumulators, and the packing of P's accumulator quads into the B operand.  Planes and the six-term
sum are not modelled (they are a sum over identical index patterns)."""
import numpy as np


def mfma_16x16x32(a, b, c):
    """a, b: [64 lanes, 8]; c: [64, 4].  A[i][k]: lane i + 16 g holds k = 8 g + t; B[k][j]: lane j + 16 g holds k = 8 g + t;
    D[i][j] in lane j + 16 (i // 4), register i % 4."""
    A = np.zeros((16, 32))
    Bm = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
        Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
    D = A @ Bm
    out = c.copy()
    for i in range(16):
        out[np.arange(16) + 16 * (i // 4), i % 4] += D[i]
    return out


def test_split_attention_tile_layout_and_fragments():
    rng = np.random.default_rng(0)
    Q = rng.standard_normal((16, 64))            # one wave: 16 queries
    K = rng.standard_normal((64, 64))            # one tile: 64 keys x 64 d
    V = rng.standard_normal((64, 64))
    # ---- staging (lstore): byte-addressed LDS images, 2 bytes per element -> model in elements
    Ks = np.full((64 * 64,), np.nan)             # [key][64]: 16-byte slot s (8 elements) of row key at slot s ^ (key & 7)
    for key in range(64):
        for slot in range(8):
            dst = key * 64 + ((slot ^ (key & 7)) << 3)
            Ks[dst:dst + 8] = K[key, 8 * slot:8 * slot + 8]
    Vs = np.full((64 * 64,), np.nan)             # [d][64 positions]
    for kp in range(32):                         # a thread stores the key pair (2 kp, 2 kp + 1) of eight d values as 4-byte words
        key = 2 * kp
        kk, a, gg, t = key >> 5, (key >> 4) & 1, (key >> 2) & 3, key & 3
        pos = 32 * kk + 8 * gg + 4 * a + t
        assert pos % 2 == 0
        for d in range(64):
            dst = d * 64 + (((pos >> 3) ^ (d & 7)) << 3) + (pos & 7)
            Vs[dst], Vs[dst + 1] = V[key, d], V[key + 1, d]
    assert not np.isnan(Ks).any() and not np.isnan(Vs).any()
    lane = np.arange(64)
    r, g = lane & 15, lane >> 4
    # ---- S^T = K . Q^T: fragment kf, halves kh
    S = np.zeros((4, 64, 4))
    for kf in range(4):
        for kh in range(2):
            key = 16 * kf + r
            src = key * 64 + (((4 * kh + g) ^ (key & 7)) << 3)
            a = np.stack([Ks[s:s + 8] for s in src])                          # lane (key r, g): d = 32 kh + 8 g ..
            b = np.stack([Q[r[l], 32 * kh + 8 * g[l]:32 * kh + 8 * g[l] + 8] for l in range(64)])
            S[kf] = mfma_16x16x32(a, b, S[kf])
    want_S = K @ Q.T                                                           # [key, query]
    for kf in range(4):
        for l in range(64):
            for e in range(4):
                assert abs(S[kf, l, e] - want_S[16 * kf + 4 * g[l] + e, r[l]]) < 1e-9   # lane (query r, g) holds keys 16 kf + 4 g + e
    # ---- P (any function of S, lane-local) packed as the B operand of key block kk: t < 4 -> fragment 2 kk, t >= 4 -> fragment 2 kk + 1
    P = np.tanh(S)
    O = np.zeros((4, 64, 4))
    for fd in range(4):
        for kk in range(2):
            d = 16 * fd + r
            src = d * 64 + (((4 * kk + g) ^ (d & 7)) << 3)
            a = np.stack([Vs[s:s + 8] for s in src])                          # lane (d r, g): the eight permuted keys of block kk
            b = np.stack([np.concatenate([P[2 * kk, l], P[2 * kk + 1, l]]) for l in range(64)])
            O[fd] = mfma_16x16x32(a, b, O[fd])
    want_O = V.T @ np.tanh(want_S)                                             # [d, query]
    for fd in range(4):
        for l in range(64):
            for e in range(4):
                assert abs(O[fd, l, e] - want_O[16 * fd + 4 * g[l] + e, r[l]]) < 1e-9   # lane (query r, g) holds d = 16 fd + 4 g + e
    # ---- bank behaviour of the 16-byte fragment reads: the four 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md) hit 16 distinct slots
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for kh in range(2):
        for kf in range(4):
            key = 16 * kf + r
            byte = key * 128 + (((4 * kh + g) ^ (key & 7)) << 4)
            for grp in groups:
                pos = [(byte[l] % 256) // 16 for l in grp]
                assert max(pos.count(x) for x in set(pos)) == 1        # conflict-free
