"""GPU probe (round 4): the persistent tile walk of the split-precision GEMM (csrc/gemm_split3.hip gemm_split3_persist_kernel) against the
one-tile-per-block ping-pong kernel -- the batched transform-domain GEMMs of the three-step Winograd layers and the ViT-L block linears.
usage: python tools/persist_probe.py [wino] [vit]      (PF_S3_PERSIST=0 / 1 is set per measurement; times are HIP-event averages)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk       # noqa: E402
from patchfusion_amd.hip_ops import ops        # noqa: E402

DEV = "cuda"


def timed(fn, modes=("0", "1")):
    out = []
    for m in modes:
        os.environ["PF_S3_PERSIST"] = m
        out.append(fn())
    os.environ.pop("PF_S3_PERSIST", None)
    return out


def wino():
    print("| batched transform-domain GEMM (36 planes) | T | one-tile ms | persistent ms | speed-up | useful TF/s | executed bf16 TF/s (incl. padding) | of 2500/6 |")
    print("|---|---|---|---|---|---|---|---|")
    g = torch.Generator(device=DEV).manual_seed(0)
    for (cin, cout, B, H, W) in ((544, 544, 8, 392, 518), (768, 768, 8, 224, 296), (768, 768, 8, 112, 148), (768, 256, 8, 224, 296),
                                 (512, 256, 8, 224, 296), (256, 256, 8, 224, 296), (768, 768, 8, 56, 74), (544, 544, 4, 392, 518)):
        T = B * -(-H // 4) * -(-W // 4)
        V3 = torch.randn(3, 36, T, cin, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
        rows = -(-cout // 16) * 16
        U3 = (torch.randn(3, 36, rows, cin, device=DEV, generator=g) / cin ** 0.5).to(torch.bfloat16)
        Mw = torch.empty(36, T, cout, device=DEV)
        t0, t1 = timed(lambda: ops.gemm_planes_split3(V3, U3, Mw, T, cin, cout, 5))
        V3k, U3k = V3.view(3, 36, cin // 32, T, 32), U3.view(3, 36, cin // 32, rows, 32)       # (random data: a reinterpretation is as good as a permute)
        k0, k1 = timed(lambda: ops.gemm_planes_split3(V3k, U3k, Mw, T, cin, cout, 5))
        fl = 36 * 2.0 * T * cin * cout
        pad = (-(-cout // 128) * 128) / cout
        print(f"| {cin}->{cout} @ {B}x{H}x{W} | {T} | {t0:.3f} | {t1:.3f} | {t0 / t1:.2f}x | {fl / t1 / 1e9:.1f} | {6 * fl * pad / t1 / 1e9:.0f} | {fl / t1 / 1e9 / (2500 / 6):.3f} |"
              f" chunk-major: {k0:.3f} | {k1:.3f} | {fl / k1 / 1e9:.1f} TF/s | {fl / k1 / 1e9 / (2500 / 6):.3f} |")
        del V3, U3, Mw
        torch.cuda.empty_cache()


def vit():
    print("\n| ViT-L linear | M | one-tile ms | persistent ms | speed-up | useful TF/s | of 2500/6 |")
    print("|---|---|---|---|---|---|---|")
    g = torch.Generator().manual_seed(0)
    for M in (8 * 1037, 1037):
        for name, K, N, act, res, scale, split_out in (("qkv", 1024, 3072, None, False, False, True), ("proj", 1024, 1024, None, True, True, False),
                                                       ("fc1", 1024, 4096, "gelu", False, False, True), ("fc2", 4096, 1024, None, True, True, False)):
            w = torch.randn(N, K, generator=g) / K ** 0.5
            b = torch.randn(N, generator=g)
            sc = (0.5 + torch.rand(N, generator=g)) if scale else None
            pw3 = pk.pack_conv_split3(w, b, scale=sc).to(DEV)
            x3 = torch.randn(3, M, K, generator=g).to(torch.bfloat16).to(DEV)
            r = torch.randn(M, N, generator=g).to(DEV) if res else None
            y = torch.empty(3, M, N, dtype=torch.bfloat16, device=DEV) if split_out else torch.empty(M, N, device=DEV)
            t0, t1 = timed(lambda: ops.conv_split3(x3, pw3, y, act=act, res=r, _timed=20), ("0", "2"))
            fl = 2.0 * M * K * N
            print(f"| {name} {K}->{N}{' gelu' if act else ''}{' +res*ls' if res else ''}{' planes out' if split_out else ''} | {M} | {t0:.3f} | {t1:.3f} | {t0 / t1:.2f}x | {fl / t1 / 1e9:.1f} | {fl / t1 / 1e9 / (2500 / 6):.3f} |")


def sustain():
    """is the dominant launch power-limited in steady state?  the same launch timed over 5 / 40 / 160 back-to-back iterations, one-tile vs
    persistent kernel, random vs all-zero operands (zeros toggle no multiplier inputs: the clock stays high)"""
    cin = cout = 544
    T = 8 * 98 * 130
    g = torch.Generator(device=DEV).manual_seed(0)
    rows = -(-cout // 16) * 16
    Mw = torch.empty(36, T, cout, device=DEV)
    print("\n| operands | kernel | ms/launch over 5 | over 40 | over 160 |")
    print("|---|---|---|---|---|")
    for data in ("random", "zeros"):
        if data == "random":
            V3 = torch.randn(3, 36, T, cin, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
            U3 = (torch.randn(3, 36, rows, cin, device=DEV, generator=g) / cin ** 0.5).to(torch.bfloat16)
        else:
            V3 = torch.zeros(3, 36, T, cin, device=DEV, dtype=torch.bfloat16)
            U3 = torch.zeros(3, 36, rows, cin, device=DEV, dtype=torch.bfloat16)
        for mode, name in (("0", "one tile per block"), ("1", "persistent")):
            os.environ["PF_S3_PERSIST"] = mode
            ts = [ops.gemm_planes_split3(V3, U3, Mw, T, cin, cout, n) for n in (5, 40, 160)]
            print(f"| {data} | {name} | {ts[0]:.3f} | {ts[1]:.3f} | {ts[2]:.3f} |")
        del V3, U3
    os.environ.pop("PF_S3_PERSIST", None)


def layers():
    """whole 3x3 layers below 512 output channels: the fused f32-MFMA Winograd kernel (csrc/wino_fused.hip) against the three-step form with the
    split-precision GEMM (round 4: persistent, chunk-major) -- the data behind hip_ops._fused_wanted"""
    from patchfusion_amd import hip_ops
    print("\n| 3x3 layer @ B=8 | fused f32-MFMA kernel ms | three-step, split GEMM ms | ratio |")
    print("|---|---|---|---|")
    g = torch.Generator().manual_seed(0)
    for (cin, cout, H, W) in ((768, 256, 224, 296), (512, 256, 224, 296), (256, 256, 224, 296), (256, 128, 224, 296), (256, 256, 196, 259),
                              (768, 256, 112, 148), (512, 256, 112, 148), (256, 256, 112, 148), (256, 256, 98, 129), (544, 32, 392, 518),
                              (128, 32, 392, 518), (512, 256, 56, 74), (768, 256, 56, 74), (256, 256, 56, 74)):
        w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
        pw = pk.pack_conv(w, torch.zeros(cout), dtype=torch.float32).to(DEV)
        x = torch.randn(8, H, W, cin, device=DEV)
        y = torch.empty(8, H, W, cout, device=DEV)
        t = {}
        for mode in ("2", "0"):
            os.environ["PF_WINO_FUSED"] = mode
            hip_ops.refresh_env()
            t[mode] = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
        os.environ.pop("PF_WINO_FUSED")
        hip_ops.refresh_env()
        print(f"| {cin}->{cout} @ {H}x{W} | {t['2']:.3f} | {t['0']:.3f} | {t['2'] / t['0']:.2f} |")
        del x, y
        torch.cuda.empty_cache()


def decomp():
    """timing decomposition of the persistent kernel on the dominant launch (debug build: PF_LIB_PATH=patchfusion_amd/libpf_wfdbg.so; results are
    wrong by construction): PF_S3_DBG bit 0 = no DMA after the ring fill, bit 1 = no MFMA, bit 2 = no fragment reads, bit 3 = no stores"""
    cin = cout = 544
    T = 8 * 98 * 130
    g = torch.Generator(device=DEV).manual_seed(0)
    rows = -(-cout // 16) * 16
    V3 = torch.randn(3, 36, T, cin, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    U3 = (torch.randn(3, 36, rows, cin, device=DEV, generator=g) / cin ** 0.5).to(torch.bfloat16)
    Mw = torch.empty(36, T, cout, device=DEV)
    V3, U3 = V3.view(3, 36, cin // 32, T, 32), U3.view(3, 36, cin // 32, rows, 32)          # chunk-major (the layout the layer uses)
    print("\n| PF_S3_DBG | what is removed | ms |")
    print("|---|---|---|")
    names = {0: "nothing (full kernel)", 1: "DMA", 2: "MFMA", 4: "fragment reads", 8: "stores", 3: "DMA + MFMA", 5: "DMA + reads", 6: "MFMA + reads",
             7: "DMA + MFMA + reads", 9: "DMA + stores", 13: "DMA + reads + stores (MFMA stream alone)", 15: "everything (tile walk + barriers only)",
             14: "MFMA + reads + stores (DMA stream alone)", 11: "DMA + MFMA + stores (fragment reads alone)"}
    for d, nm in names.items():
        os.environ["PF_S3_DBG"] = str(d)
        t = ops.gemm_planes_split3(V3, U3, Mw, T, cin, cout, 5)
        print(f"| {d} | {nm} | {t:.3f} |")
    os.environ.pop("PF_S3_DBG", None)


def timeline192():
    """the same stamps in the 192 x 192 kernel (round 5: group B issues its W pieces from its own load phase)"""
    os.environ["PF_S3_T192"] = "2"
    timeline(names_a=["tile cursor / stores", "DMA issue (9 pieces)", "27 fragment reads issued", "lgkmcnt(0) + barrier [phase ends]", "108 MFMAs",
                      "vmcnt(0)", "closing barrier [phase ends]"],
             names_b=["tile cursor / stores", "DMA issue (9 pieces)", "27 fragment reads issued", "vmcnt(0)", "lgkmcnt(0) + barrier [phase ends]", "108 MFMAs",
                      "closing barrier [phase ends]"])
    os.environ.pop("PF_S3_T192", None)


def timeline(names_a=None, names_b=None):
    """s_memtime stamps of the persistent kernel's phases (debug build: PF_LIB_PATH=patchfusion_amd/libpf_wfdbg.so), dominant launch, block 0,
    waves 0 (group A) and 4 (group B), stream chunks 64 .. 95: where a chunk's ~2600 cycles go"""
    import ctypes as C
    from patchfusion_amd import _lib
    from patchfusion_amd.hip_ops import _L, _stream
    cin = cout = 544
    T = 8 * 98 * 130
    g = torch.Generator(device=DEV).manual_seed(0)
    rows = -(-cout // 16) * 16
    V3 = torch.randn(3, 36, T, cin, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    U3 = (torch.randn(3, 36, rows, cin, device=DEV, generator=g) / cin ** 0.5).to(torch.bfloat16)
    Mw = torch.empty(36, T, cout, device=DEV)
    tl = torch.zeros(2 * 32 * 8, dtype=torch.int64, device=DEV)
    p = _lib.ConvParams()
    V3, U3 = V3.view(3, 36, cin // 32, T, 32), U3.view(3, 36, cin // 32, rows, 32)          # chunk-major (the layout the layer uses)
    p.x, p.x_ld, p.B, p.H, p.W, p.Cin = V3.data_ptr(), cin, 1, 1, T, cin
    p.w, p.w_rows, p.Kpad = U3.data_ptr(), rows, cin
    p.y, p.y_ld, p.OH, p.OW, p.Cout = Mw.data_ptr(), cout, 1, T, cout
    p.KH = p.KW = p.stride = 1
    p.act, p.shuffle, p.dtype, p.out_f32, p.batch = 0, 1, 1, 1, 36
    p.korder = 6
    p.x_bstride, p.w_bstride = V3.stride(0), U3.stride(0)
    p.res2 = tl.data_ptr()
    for _ in range(2):
        _lib.check(_L.pf_gemm_split3(C.byref(p), _stream()), "pf_gemm_split3")
    torch.cuda.synchronize()
    t = tl.cpu().view(2, 32, 8)
    names = ["epilogue/decode", "DMA issue (6 pieces)", "fragment reads issued", "vmcnt wait", "lgkmcnt(0) + barrier", "48 MFMAs", "closing barrier"]
    for grp in range(2):
        names_g = (names_a, names_b)[grp] or names
        print("\n| wave group | " + " | ".join(names_g) + " | chunk total |")
        print("|---|" + "---|" * (len(names_g) + 1))
        d = (t[grp, :, 1:] - t[grp, :, :-1]).double()
        tot = (t[grp, 1:, 0] - t[grp, :-1, 0]).double()
        print(f"| {'AB'[grp]} mean of 32 chunks | " + " | ".join(f"{float(d[:, i].mean()):.0f}" for i in range(7)) + f" | {float(tot.mean()):.0f} |")
        print(f"| {'AB'[grp]} median | " + " | ".join(f"{float(d[:, i].median()):.0f}" for i in range(7)) + f" | {float(tot.median()):.0f} |")
        print(f"| {'AB'[grp]} max | " + " | ".join(f"{float(d[:, i].max()):.0f}" for i in range(7)) + f" | {float(tot.max()):.0f} |")
    print("(s_memtime ticks = shader clock cycles; the instrumentation itself costs ~10 % of a chunk)")


def ordersweep():
    """tile order of the persistent kernels for nt = 3 / 5 / 6 channel tiles: channel tile fastest (default) against the 30-tile patches (PF_S3_ORDER=0)"""
    _sweep((("PF_S3_ORDER",), ("0", "1")), only_wino=True)
    os.environ["PF_S3_T192"] = "0"
    _sweep((("PF_S3_ORDER",), ("0", "1")), only_wino=True)
    os.environ.pop("PF_S3_T192")


def t192sweep():
    """the 192 x 192 persistent kernel (two-slot ring, PF_S3_T192=2) against the 128 x 128 one (PF_S3_T192=0) on the same launches"""
    _sweep((("PF_S3_T192",), ("0", "2")))


def envsweep(spec):
    """generic interleaved per-launch A/B of one library switch: `envsweep:NAME=a,b,c` (Winograd-domain GEMMs + the ViT-L linears, PF_S3_T192=2 for the
    latter so that every shape runs the kernel under test)"""
    name, vals = spec.split("=", 1)
    _sweep(((name,), tuple(vals.split(","))))


def _sweep(knob, only_wino=False):
    names, p2s = knob
    print("| launch | " + " | ".join(f"{names[0]}={v} ms" for v in p2s) + " | best | useful TF/s at best | of 2500/6 |")
    print("|---|" + "---|" * (len(p2s) + 3))
    g = torch.Generator(device=DEV).manual_seed(0)

    def row(name, fl, fn):
        # (round 5: an untimed pass first, then THREE interleaved rounds over the variants, mean per variant -- the first timing after the tensors are made runs on
        #  a cold clock: a single A, B pass credited the second arm with up to 14 % it had not earned, profiles/r5_sweep_order_bias.md)
        for nm in names:
            os.environ[nm] = p2s[0]
        fn()
        acc = [0.0] * len(p2s)
        for _ in range(3):
            for i, v in enumerate(p2s):
                for nm in names:
                    os.environ[nm] = v
                acc[i] += fn()
        ts = [a / 3 for a in acc]
        for nm in names:
            os.environ.pop(nm, None)
        b = min(range(len(ts)), key=lambda i: ts[i])
        print(f"| {name} | " + " | ".join(f"{t:.3f}" for t in ts) + f" | {names[0]}={p2s[b]} ({ts[0] / ts[b]:.2f}x) | {fl / ts[b] / 1e9:.1f} | {fl / ts[b] / 1e9 / (2500 / 6):.3f} |", flush=True)

    for (cin, cout, B, H, W) in ((544, 544, 8, 392, 518), (768, 768, 8, 224, 296), (768, 256, 8, 224, 296), (512, 256, 8, 224, 296), (256, 256, 8, 224, 296),
                                 (768, 768, 8, 112, 148), (768, 768, 8, 56, 74)):
        T = B * -(-H // 4) * -(-W // 4)
        rows = -(-cout // 16) * 16
        V3k = torch.randn(3, 36, cin // 32, T, 32, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
        U3k = (torch.randn(3, 36, cin // 32, rows, 32, device=DEV, generator=g) / cin ** 0.5).to(torch.bfloat16)
        Mw = torch.empty(36, T, cout, device=DEV)
        row(f"wino GEMM {cin}->{cout} @ {B}x{H}x{W}", 36 * 2.0 * T * cin * cout, lambda: ops.gemm_planes_split3(V3k, U3k, Mw, T, cin, cout, 5))
        del V3k, U3k, Mw
        torch.cuda.empty_cache()
    if only_wino:
        return
    gc = torch.Generator().manual_seed(0)
    os.environ["PF_S3_PERSIST"] = "2"
    for M in (8 * 1037, 1037):
        for name, K, N, act, res, scale, split_out in (("qkv", 1024, 3072, None, False, False, True), ("proj", 1024, 1024, None, True, True, False),
                                                       ("fc1", 1024, 4096, "gelu", False, False, True), ("fc2", 4096, 1024, None, True, True, False)):
            w = torch.randn(N, K, generator=gc) / K ** 0.5
            b = torch.randn(N, generator=gc)
            sc = (0.5 + torch.rand(N, generator=gc)) if scale else None
            pw3 = pk.pack_conv_split3(w, b, scale=sc).to(DEV)
            x3 = torch.randn(3, K // 32, M, 32, generator=gc).to(torch.bfloat16).to(DEV)
            r = torch.randn(M, N, generator=gc).to(DEV) if res else None
            y = torch.empty(3, N // 32, M, 32, dtype=torch.bfloat16, device=DEV) if split_out else torch.empty(M, N, device=DEV)
            row(f"{name} {K}->{N} M={M}", 2.0 * M * K * N, lambda: ops.conv_split3(x3, pw3, y, act=act, res=r, _timed=20))
    os.environ.pop("PF_S3_PERSIST", None)


def traffic():
    """is the dominant launch bounded by its operand traffic?  (debug build: PF_LIB_PATH=patchfusion_amd/libpf_wfdbg.so; PF_S3_DBG=16 makes every
    tile of the 192 x 192 kernel read the X rows of ONE token tile -- L2 hits -- while the MFMA / DMA / LDS work and the M stores stay the same)"""
    cin = cout = 544
    T = 8 * 98 * 130
    g = torch.Generator(device=DEV).manual_seed(0)
    rows = -(-cout // 16) * 16
    V3 = torch.randn(3, 36, cin // 32, T, 32, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    U3 = (torch.randn(3, 36, cin // 32, rows, 32, device=DEV, generator=g) / cin ** 0.5).to(torch.bfloat16)
    Mw = torch.empty(36, T, cout, device=DEV)
    print("\n| PF_S3_DBG | X operand | ms |")
    print("|---|---|---|")
    for d, nm in ((0, "every tile its own token rows (full kernel)"), (16, "every tile the rows of token tile 0 (L2-resident)"), (0, "full kernel again")):
        os.environ["PF_S3_DBG"] = str(d)
        t = ops.gemm_planes_split3(V3, U3, Mw, T, cin, cout, 5)
        print(f"| {d} | {nm} | {t:.3f} |")
    os.environ.pop("PF_S3_DBG", None)


if __name__ == "__main__":
    what = sys.argv[1:] or ["wino", "vit"]
    if "wino" in what:
        wino()
    if "vit" in what:
        vit()
    if "sustain" in what:
        sustain()
    if "timeline" in what:
        timeline()
    if "timeline192" in what:
        timeline192()
    if "decomp" in what:
        decomp()
    if "layers" in what:
        layers()
    if "traffic" in what:
        traffic()
    if "ordersweep" in what:
        ordersweep()
    if "t192sweep" in what:
        t192sweep()
    for w in what:
        if w.startswith("envsweep:"):
            envsweep(w.split(":", 1)[1])
