"""Host replay of the lane / register algebra of csrc/swin.hip swin_window_attention_mfma_kernel (round 6) with the documented operand layout of
v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md / cdna_hip_programming.md: A -- lane l holds A[m = l % 16][k = l / 16]; B -- lane l holds B[k = l / 16][n = l % 16];
C / D -- register i of lane l holds D[m = 4 (l / 16) + i][n = l % 16]):
  * S^T = K Q^T puts the scores of ONE query (n = l % 16) into a lane: keys 16 kt + 4 (l / 16) + i over the nine key tiles;
  * the relative-position bias is read at byte offset  [(yi + 11) 23 + xi + 11] 4 - 4 (23 yj + xj)  of the head's 529-entry row (koff table of the kernel);
  * O^T = V^T P^T: MFMA number (kt, i) takes the lane's probability register [kt][i] as its B operand, so its contraction index kk = l / 16 must stand for
    key 16 kt + 4 kk + i -- which is the V row the kernel hands it as A operand.
The replay evaluates one (window, head) with numpy exactly in that dataflow and compares with softmax(q k^T / sqrt(d) + bias + mask) v.  Test
infrastructure: mirrors the index expressions of the kernel; the kernel itself is checked on the GPU by tests/op_checks.py swin_ops."""
import numpy as np
import pytest

WIN, WTOK = 12, 144


def mfma_16x16x4(a_lane, b_lane, acc):
    """a_lane, b_lane: [64] per-lane operands; acc: [64, 4] per-lane accumulator registers -> new acc"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for l in range(64):
        A[l % 16, l // 16] = a_lane[l]
        B[l // 16, l % 16] = b_lane[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for i in range(4):
            out[l, i] += D[4 * (l // 16) + i, l % 16]
    return out


def relative_position_index():
    c = np.stack(np.meshgrid(np.arange(WIN), np.arange(WIN), indexing="ij")).reshape(2, -1)
    rel = c[:, :, None] - c[:, None, :]
    return (rel[0] + WIN - 1) * (2 * WIN - 1) + rel[1] + WIN - 1          # [144, 144], swin_layers.py:104-113


@pytest.mark.parametrize("hd,masked", [(4, False), (16, True), (32, True)])
def test_one_item_of_the_kernel_matches_plain_attention(hd, masked):
    rng = np.random.default_rng(hd)
    q, k, v = (rng.standard_normal((WTOK, hd)) for _ in range(3))
    table = rng.standard_normal(529) * 0.5
    region = (rng.integers(0, 3, WTOK) if masked else np.zeros(WTOK, int))
    scale = hd ** -0.5
    s = (q * scale) @ k.T + table[relative_position_index()]
    s = s + np.where(region[:, None] != region[None, :], -100.0, 0.0)
    p = np.exp(s - s.max(1, keepdims=True))
    ref = (p / p.sum(1, keepdims=True)) @ v
    koff = np.array([-4 * ((t // WIN) * (2 * WIN - 1) + t % WIN) for t in range(WTOK)])        # the kernel's byte steps
    out = np.zeros((WTOK, hd))
    for qt in range(9):
        # ---- S^T tiles: acc[kt] = sum_j mfma(A = K[16 kt + r][4 j + c], B = Q[16 qt + r][4 j + c])
        acc = np.zeros((9, 64, 4))
        for j in range((hd + 3) // 4):
            b = np.array([q[qt * 16 + (l % 16), 4 * j + l // 16] * scale if 4 * j + l // 16 < hd else 0.0 for l in range(64)])
            for kt in range(9):
                a = np.array([k[kt * 16 + (l % 16), 4 * j + l // 16] if 4 * j + l // 16 < hd else 0.0 for l in range(64)])
                acc[kt] = mfma_16x16x4(a, b, acc[kt])
        # ---- bias, mask, softmax in the lane's registers: query r = l % 16, keys 16 kt + 4 c + i
        P = np.zeros_like(acc)
        inv = np.zeros(64)
        for l in range(64):
            r, c = l % 16, l // 16
            tq = qt * 16 + r
            base = 4 * ((WIN - 1) * (2 * WIN - 1) + WIN - 1) - koff[tq]                           # byte address of the query's bias origin
            sc = np.zeros((9, 4))
            for kt in range(9):
                for i in range(4):
                    key = kt * 16 + 4 * c + i
                    sc[kt, i] = acc[kt, l, i] + table[(base + koff[key]) // 4] + (-100.0 if region[key] != region[tq] else 0.0)
            # maximum / sum over the lane's 36 registers and the four lanes l, l ^ 16, l ^ 32, l ^ 48 -- same query, the other key quarters
            P[:, l, :] = sc
        mx = np.array([max(P[:, l % 16 + 16 * cc, :].max() for cc in range(4)) for l in range(64)])
        for l in range(64):
            P[:, l, :] = np.exp2((P[:, l, :] - mx[l]) * 1.4426950408889634)
        for l in range(64):
            inv[l] = 1.0 / sum(P[:, l % 16 + 16 * cc, :].sum() for cc in range(4))
        # ---- O^T = V^T P^T: MFMA (kt, i) with A = V[16 kt + 4 c + i][16 dt + r], B = the lane's P register [kt][i]
        for dt in range(max(1, hd // 16)):
            o = np.zeros((64, 4))
            for kt in range(9):
                for i in range(4):
                    a = np.array([v[kt * 16 + 4 * (l // 16) + i, dt * 16 + l % 16] if dt * 16 + l % 16 < hd else 0.0 for l in range(64)])
                    o = mfma_16x16x4(a, P[kt, :, i], o)
            for l in range(64):
                r, c = l % 16, l // 16
                for i in range(4):
                    d = dt * 16 + 4 * c + i                                                         # register i of lane l = O^T[d][query r]
                    if d < hd:
                        out[qt * 16 + r, d] = o[l, i] * inv[l]
    assert np.abs(out - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()) * 1e3
