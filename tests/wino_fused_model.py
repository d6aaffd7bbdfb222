"""Lane-level numpy model of csrc/wino_fused.hip (test infrastructure).

The fused Winograd kernel cannot be executed in the build container (no GPU), and almost everything that can go wrong in it is
INDEX arithmetic: the super-tile / block order, the LDS-DMA slot mapping with its pixel-pair swizzle, the transform lanes, the V
image, the MFMA fragment lanes against the pre-packed filter order (packing.winograd_filters_fused), the accumulator layout and
the epilogue exchange.  This model replays exactly those formulas -- same constants, same per-lane expressions, byte addresses
into emulated LDS arrays, v_mfma_f32_16x16x4_f32 semantics on 64-lane operand vectors -- so that tests/test_winograd_cpu.py can
check the mapping against F.conv2d before a GPU minute is spent.  It does not model timing, s_waitcnt or barriers.
"""
import numpy as np

NT = 32
PIECES = 40                       # 1 KiB LDS-DMA pieces per 16-channel raw stage (10 per DMA wave, 5 per chunk interval)
RAW_STAGE = PIECES * 1024
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


def swz(sx):
    """pixel-pair swizzle mask of a raw pixel whose column is in tile column sx (= rx >> 2)"""
    return ((sx & 1) << 2) | (sx & 2)


def run(x, up, bias, relu, relu_in, res, res2, cout, gs=8, SW=8, conflicts=None):
    """x [B,H,W,Cin] NHWC; up = winograd_filters_fused(...) as numpy [nnb,nkc,36,2,64,4]; SW = super-tile width in tiles (8: 4x8 tiles,
    4: 8x4); returns y [B,H,W,cout].  `conflicts`: optional list that receives the worst LDS bank multiplicity of the transform reads."""
    x = np.asarray(x, dtype=np.float64)
    B, H, W, Cin = x.shape
    up_f = np.asarray(up, dtype=np.float64).reshape(-1)            # index in floats = byte offset / 4
    SH = 32 // SW
    RR, RCc = 4 * SH + 2, 4 * SW + 2
    RP = RCc // 2
    ROWB = RP * 128
    RAW_USED = RR * RCc * 4
    assert RAW_USED <= PIECES * 64 and Cin % 16 == 0
    TH, TW = (H + 3) // 4, (W + 3) // 4
    NSY, NSX = (TH + SH - 1) // SH, (TW + SW - 1) // SW
    nsuper, nnb, nkc = B * NSY * NSX, (cout + 63) // 64, Cin // 8
    NG = nkc // 2
    y = np.full((B, H, W, cout), np.nan)
    xf = x.reshape(-1)
    x_ld = Cin
    lane = np.arange(64)
    nblocks = nsuper * nnb
    seen = set()
    for bid in range(nblocks):          # (the XCD remap is a bijection of block ids; the model walks logical ids)
        per_group = gs * nnb
        group = bid // per_group
        first = group * gs
        gsz = min(nsuper - first, gs)
        in_g = bid - group * per_group
        nb = in_g // gsz
        sidx = first + (in_g - nb * gsz)
        assert (sidx, nb) not in seen
        seen.add((sidx, nb))
        b_img = sidx // (NSY * NSX)
        rem = sidx - b_img * NSY * NSX
        sty, stx = rem // NSX, rem % NSX
        n0 = nb * 64
        # ---- DMA offsets per (wave pg, i, lane): i < 5 first half of a stage, i >= 5 second half
        doff = np.full((4, 10, 64), -1, dtype=np.int64)
        dpiece = np.zeros((4, 10), dtype=np.int64)
        for pg in range(4):
            for i in range(10):
                piece = (i // 5) * 20 + 5 * pg + (i % 5)
                dpiece[pg, i] = piece
                for l in range(64):
                    slot = piece * 64 + l
                    if slot < RAW_USED:
                        pairidx, sp = slot >> 3, slot & 7
                        ry, pr = pairidx // RP, pairidx % RP
                        sg = sp ^ swz(pr >> 1)
                        xp, q = sg >> 2, sg & 3
                        rx = 2 * pr + xp
                        iy, ix = 4 * SH * sty - 1 + ry, 4 * SW * stx - 1 + rx
                        if 0 <= iy < H and 0 <= ix < W:
                            doff[pg, i, l] = ((b_img * H + iy) * W + ix) * x_ld + 4 * q
        assert sorted(dpiece.reshape(-1).tolist()) == list(range(40))
        acc = np.zeros((8, 9, 2, 2, 64, 4))      # [wave][plane][tg][cg][lane][e]
        raw = None
        for kc in range(nkc):
            G, j = kc >> 1, kc & 1
            if j == 0:                            # a raw stage holds the 16 channels of chunks 2G and 2G+1
                raw = np.full(RAW_STAGE // 4, np.nan)
                for pg in range(4):
                    for i in range(10):
                        dst = (dpiece[pg, i] * 1024) // 4
                        for l in range(64):
                            o = doff[pg, i, l]
                            raw[dst + 4 * l: dst + 4 * l + 4] = xf[o + 16 * G: o + 16 * G + 4] if o >= 0 else 0.0
            # ---- transform (waves 0-3): unit = (tile slot 8 pg + lane/8, channel lane%8 of this chunk)
            V = np.zeros(V_STAGE // 4)
            jx = j << 5
            for pg in range(4):
                banks = {}
                for l in range(64):
                    tc, tsl = l & 7, 8 * pg + (l >> 3)
                    sy, sx = tsl // SW, tsl % SW
                    h, cc = tc >> 2, tc & 3
                    m0, m1 = swz(sx), swz(sx + 1)
                    col = [(((4 * sy * RP + 2 * sx + (b >> 1)) * 8) + ((((b & 1) << 2) | h) ^ (m0 if b < 4 else m1))) * 16 + cc * 4 for b in range(6)]
                    t_wr = ((tc >> 1) * 64 + (tsl >> 4) * 32 + ((((tsl & 15) + 4 * (tc >> 1)) & 15) * 2) + (tc & 1)) * 4
                    if conflicts is not None:
                        banks.setdefault(('w', l >> 5), []).append((t_wr // 4) % 32)
                    d = np.empty((6, 6))
                    for a in range(6):
                        for b in range(6):
                            adr = (col[b] ^ jx) + a * ROWB
                            v = raw[adr // 4]
                            assert not np.isnan(v), (pg, l, a, b)
                            d[a, b] = max(v, 0.0) if relu_in else v
                            if conflicts is not None:
                                banks.setdefault((a, b, l >> 5), []).append((adr // 4) % 32)
                    v = BT @ d @ BT.T
                    for i in range(6):
                        for jj in range(6):
                            V[(t_wr + (i * 6 + jj) * 1024) // 4] = v[i, jj]
                if conflicts is not None:
                    conflicts.append(max(max(np.bincount(np.array(bk), minlength=32)) for bk in banks.values()))
            # ---- MFMA (all waves)
            for wave in range(8):
                pg, half = wave & 3, wave >> 2
                if n0 + half * 32 >= cout:
                    continue
                r, g4 = lane & 15, lane >> 4
                b_rd = (g4 * 64 + ((r + 4 * g4) & 15) * 2) * 4 + 9 * pg * 1024
                if conflicts is not None and kc == 0:       # ds_read2_b64: 16-lane groups, 32 banks, 2 dwords per lane
                    for grp in range(4):
                        bk = np.concatenate([((b_rd[16 * grp:16 * grp + 16] // 4) + e) % 32 for e in range(2)])
                        conflicts.append(int(max(np.bincount(bk, minlength=32))))
                ubase = ((((nb * nkc) * 36 + 9 * pg) * 2 + half) * 1024 + lane * 16) // 4
                for P in range(9):
                    uo = ubase + (kc * U_CHUNK + P * U_PLANE) // 4
                    u = np.stack([up_f[uo + e] for e in range(4)], axis=1)        # [64, 4]
                    for tg in range(2):
                        bo = (b_rd + P * 1024 + tg * 128) // 4
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
                    tl, equad = 8 * pg + (l >> 3), l & 7              # read side: 8 tiles x 8 channel quads per wave
                    n = n0 + half * 32 + 4 * equad
                    mp = (tl * 8 + (equad ^ (tl & 7))) * 16
                    m = np.empty((6, 6, 4))
                    for i in range(6):
                        for jj in range(6):
                            o = (mp + (i * 6 + jj) * (32 * 8 * 16)) // 4
                            m[i, jj] = M[o:o + 4]
                    ty, tx = SH * sty + tl // SW, SW * stx + tl % SW
                    if ty >= TH or tx >= TW or n >= cout:
                        continue
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
                                v += res[b_img, oy, ox, n:n + 4]
                            if res2 is not None:
                                v += res2[b_img, oy, ox, n:n + 4]
                            assert np.isnan(y[b_img, oy, ox, n:n + 4]).all()      # written exactly once
                            y[b_img, oy, ox, n:n + 4] = v
    assert len(seen) == nblocks
    return y
