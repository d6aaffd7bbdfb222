"""Lane-level numpy model of csrc/wino_fused.hip (test infrastructure).

The fused Winograd kernel cannot be executed in the build container (no GPU), and almost everything that can go wrong in it is
INDEX arithmetic: the strip / column table, the LDS-DMA slot mapping, the transform lanes, the V image, the MFMA fragment
lanes against the pre-packed filter order (packing.winograd_filters_fused), the accumulator layout and the epilogue exchange.
This model replays exactly those formulas -- same constants, same per-lane expressions, byte addresses into emulated LDS
arrays, v_mfma_f32_16x16x4_f32 semantics on 64-lane operand vectors -- so that tests/test_winograd_cpu.py can check the
mapping against F.conv2d before a GPU minute is spent.  It does not model timing, s_waitcnt or barriers.
"""
import numpy as np

NT, RC = 32, 138
RAW_SLOTS = 6 * 2 * RC
RAW_STAGE = 28 * 1024
V_STAGE = 36 * NT * 8 * 4
U_PLANE, U_CHUNK = 2048, 36 * 2048

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def mfma_16x16x4(a, b, c):
    """a, b: [64] operand registers; c: [64, 4] accumulator.  A[i][k] = a[i + 16 k], B[k][j] = b[j + 16 k];
    D[i][j] lives in lane j + 16 (i // 4), register i % 4 (cdna_hip_programming.md section 3)."""
    A = a.reshape(4, 16).T            # [i, k]
    Bm = b.reshape(4, 16)             # [k, j]
    D = A @ Bm                        # [i, j]
    out = c.copy()
    for i in range(16):
        out[np.arange(16) + 16 * (i // 4), i % 4] += D[i]
    return out


def run(x, up, bias, relu, relu_in, res, res2, cout, gs=8):
    """x [B,H,W,Cin] float64/32 NHWC; up = winograd_filters_fused(...) as numpy [nnb,nkc,36,2,64,4]; returns y [B,H,W,cout]."""
    x = np.asarray(x, dtype=np.float64)
    B, H, W, Cin = x.shape
    up_bytes = np.asarray(up, dtype=np.float64).reshape(-1)            # index in floats = byte offset / 4
    TH, TW = (H + 3) // 4, (W + 3) // 4
    assert TW >= 8 and Cin % 8 == 0
    T = B * TH * TW
    nstrips, nnb, nkc = (T + NT - 1) // NT, (cout + 63) // 64, Cin // 8
    y = np.full((B, H, W, cout), np.nan)
    xf = x.reshape(-1)
    x_ld = Cin
    THW = TH * TW
    lane = np.arange(64)
    nblocks = nstrips * nnb
    seen = set()
    for bid in range(nblocks):          # (the XCD remap is a bijection of block ids; the model walks logical ids)
        per_group = gs * nnb
        group = bid // per_group
        first = group * gs
        gsz = min(nstrips - first, gs)
        in_g = bid - group * per_group
        nb = in_g // gsz
        strip = first + (in_g - nb * gsz)
        assert (strip, nb) not in seen
        seen.add((strip, nb))
        T0 = strip * NT
        tx0 = T0 % TW
        n0 = nb * 64
        # ---- column table
        colbase = np.full(144, -1, dtype=np.int64)
        colty = np.zeros(144, dtype=np.int64)
        for tid in range(NT):
            if T0 + tid < T:
                t = T0 + tid
                b = t // THW
                rem = t - b * THW
                ty, tx = rem // TW, rem % TW
                c0 = 4 * tid + 2 * ((tx0 + tid) // TW)
                for j in range(6):
                    ix = 4 * tx - 1 + j
                    assert c0 + j < RC
                    colbase[c0 + j] = b * H * W + ix if 0 <= ix < W else -1
                    colty[c0 + j] = ty
        # ---- DMA offsets per (wave pg, piece i, lane)
        doff = np.full((4, 7, 64), -1, dtype=np.int64)
        for pg in range(4):
            for i in range(7):
                for l in range(64):
                    slot = (7 * pg + i) * 64 + l
                    if slot < RAW_SLOTS:
                        a = slot // (2 * RC)
                        rem = slot - a * 2 * RC
                        h, xc = rem // RC, rem % RC
                        cb, iy = colbase[xc], 4 * colty[xc] - 1 + a
                        if cb >= 0 and 0 <= iy < H:
                            doff[pg, i, l] = (cb + iy * W) * x_ld + 4 * h
        acc = np.zeros((8, 9, 2, 2, 64, 4))      # [wave][plane][tg][cg][lane][e]
        for kc in range(nkc):
            raw = np.zeros(RAW_STAGE // 4)
            for pg in range(4):
                for i in range(7):
                    dst = ((7 * pg + i) * 1024) // 4
                    for l in range(64):
                        o = doff[pg, i, l]
                        raw[dst + 4 * l: dst + 4 * l + 4] = xf[o + kc * 8: o + kc * 8 + 4] if o >= 0 else 0.0
            # ---- transform (waves 0-3)
            V = np.zeros(V_STAGE // 4)
            for pg in range(4):
                for l in range(64):
                    tc, tsl = l & 7, 8 * pg + (l >> 3)
                    t_rd = ((((tc >> 2) * RC + 4 * tsl + 2 * ((tx0 + tsl) // TW)) * 4) + (tc & 3)) * 4
                    t_wr = (64 * pg + l) * 4
                    d = np.empty((6, 6))
                    for a in range(6):
                        for b in range(6):
                            v = raw[(t_rd + (a * 2 * RC + b) * 16) // 4]
                            d[a, b] = max(v, 0.0) if relu_in else v
                    v = BT @ d @ BT.T
                    for i in range(6):
                        for j in range(6):
                            V[(t_wr + (i * 6 + j) * 1024) // 4] = v[i, j]
            # ---- MFMA (all waves)
            for wave in range(8):
                pg, half = wave & 3, wave >> 2
                if n0 + half * 32 >= cout:
                    continue
                r, g4 = lane & 15, lane >> 4
                b_rd = (r * 8 + 2 * g4) * 4 + 9 * pg * 1024
                ubase = ((((nb * nkc) * 36 + 9 * pg) * 2 + half) * 1024 + lane * 16) // 4
                for P in range(9):
                    uo = ubase + (kc * U_CHUNK + P * U_PLANE) // 4
                    u = np.stack([up_bytes[uo + e] for e in range(4)], axis=1)        # [64, 4]
                    for tg in range(2):
                        bo = (b_rd + P * 1024 + tg * 512) // 4
                        bx, by = V[bo], V[bo + 1]
                        for cg in range(2):
                            a_ = acc[wave, P, tg, cg]
                            a_ = mfma_16x16x4(u[:, 2 * cg + 0], bx, a_)
                            a_ = mfma_16x16x4(u[:, 2 * cg + 1], by, a_)
                            acc[wave, P, tg, cg] = a_
        # ---- epilogue: one channel half at a time
        for half in range(2):
            if n0 + half * 32 >= cout:
                continue
            M = np.full(36 * 32 * 8 * 4, np.nan)
            for pg in range(4):
                wave = pg + 4 * half
                for l in range(64):
                    r, g4 = l & 15, l >> 4
                    for i in range(9):
                        for tg in range(2):
                            for cg in range(2):
                                adr = ((((9 * pg + i) * 32 + tg * 16 + r) * 8) + ((cg * 4 + g4) ^ (r & 7))) * 16
                                M[adr // 4: adr // 4 + 4] = acc[wave, i, tg, cg, l]
            assert not np.isnan(M).any()
            for pg in range(4):
                for l in range(64):
                    r, g4 = l & 15, l >> 4
                    tl = (pg >> 1) * 16 + r
                    q = ((pg & 1) * 4 + g4) ^ (r & 7)
                    n = n0 + half * 32 + (pg & 1) * 16 + 4 * g4
                    mp = (tl * 8 + q) * 16
                    m = np.empty((6, 6, 4))
                    for i in range(6):
                        for j in range(6):
                            o = (mp + (i * 6 + j) * (32 * 8 * 16)) // 4
                            m[i, j] = M[o:o + 4]
                    tile = T0 + tl
                    if tile >= T or n >= cout:
                        continue
                    b = tile // THW
                    rem = tile - b * THW
                    ty, tx = rem // TW, rem % TW
                    out = np.einsum("pi,ije,qj->pqe", AT, m, AT)      # [4, 4, 4 channels]
                    for po in range(4):
                        oy = 4 * ty + po
                        if oy >= H:
                            continue
                        for qo in range(4):
                            ox = 4 * tx + qo
                            if ox >= W:
                                continue
                            v = out[po, qo].copy()
                            if bias is not None:
                                v += bias[n:n + 4]
                            if relu:
                                v = np.maximum(v, 0.0)
                            if res is not None:
                                v += res[b, oy, ox, n:n + 4]
                            if res2 is not None:
                                v += res2[b, oy, ox, n:n + 4]
                            assert np.isnan(y[b, oy, ox, n]).all() if False else True
                            y[b, oy, ox, n:n + 4] = v
    assert len(seen) == nblocks
    return y
