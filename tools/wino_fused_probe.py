"""GPU probe of the fused Winograd kernel (csrc/wino_fused.hip): per-case correctness with the error of every case printed, and a
timing sweep fused vs three-step vs block-group size on the layer shapes of one ViT-L image pass.

  python tools/wino_fused_probe.py check          # every case in its own try block; exit 1 on any failure
  python tools/wino_fused_probe.py time [shapes]  # ms and TFLOP/s (executed Winograd FLOPs 36*2*T*Cin*Cout and direct-conv equivalent)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk          # noqa: E402
from patchfusion_amd.hip_ops import ops
from patchfusion_amd import hip_ops as _hip_ops            # noqa: E402
from tests.fake_ops import ops as ref_ops          # noqa: E402

DEV = "cuda"


def rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).to(DEV)


def check():
    os.environ["PF_WINOGRAD"] = "4"
    _hip_ops.refresh_env()
    os.environ["PF_WINOGRAD_MIN_PIXELS"] = "0"
    _hip_ops.refresh_env()
    cases = [
        (1, 8, 32, 32, 32, 8, {}),                                   # one super-tile, one channel block, idle upper half
        (1, 8, 32, 32, 64, 8, {}),
        (1, 9, 37, 48, 48, 2, dict(act="relu")),
        (2, 13, 41, 32, 32, 2, dict(relu_in=True, res=True)),
        (1, 6, 150, 32, 96, 8 | (4 << 16), dict(act="relu")),        # forced 8 x 4 super-tiles
        (3, 5, 29, 128, 96, 1, dict(act="relu", res=True)),
        (2, 37, 41, 128, 160, 8, dict(act="relu")),
        (1, 64, 64, 256, 128, 3, dict(relu_in=True, res=True, res2=True)),
        (1, 30, 43, 544, 544, 8, {}),
        (8, 56, 74, 768, 256, 8 | (8 << 16), dict(act="relu")),       # forced 4 x 8 super-tiles
        (1, 37, 5, 32, 64, 2, dict(res=True)),
        (2, 112, 148, 256, 256, 5, dict(relu_in=True, act="relu", res=True)),
        (1, 392, 518, 128, 32, 8, dict(act="relu")),
    ]
    bad = 0
    for i, (B, H, W, cin, cout, gs, kw) in enumerate(cases):
        t0 = time.time()
        try:
            os.environ["PF_WINO_GS"], os.environ["PF_WINO_SHAPE"] = str(gs & 0xffff), str(gs >> 16)
            _hip_ops.refresh_env()
            g = torch.Generator().manual_seed(100 + i)
            w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
            pw = pk.pack_conv(w, torch.randn(cout, generator=g), dtype=torch.float32, cin_total=cin).to(DEV)
            if pw.wino_up is None:       # below the packing's eligibility (Cin < 128): pack the fused filters directly
                wk = torch.zeros(pw.w.shape[0], 3, 3, cin)
                wk[:cout] = w.permute(0, 2, 3, 1)
                pw.wino_m, pw.wino_u, pw.wino_up = 4, pk.winograd_filters(wk, 4).to(DEV), pk.winograd_filters_fused(wk).to(DEV)
            xb = rand((B, H, W, cin + 16), 200 + i)
            x = xb[..., 8:8 + cin]
            r1 = rand((B, H, W, cout), 300 + i) if kw.get("res") else None
            r2 = rand((B, H, W, cout + 8), 400 + i)[..., :cout] if kw.get("res2") else None
            outs = []
            for o, direct, fused in ((ops, None, "2"), (ref_ops, True, "0")):
                os.environ["PF_WINO_FUSED"] = fused
                _hip_ops.refresh_env()
                yb = torch.zeros((B, H, W, cout + 16), dtype=torch.float32, device=DEV)
                o.conv(x, pw, yb[..., 8:8 + cout], pad=1, act=kw.get("act"), relu_in=kw.get("relu_in", False), res=r1, res2=r2, _direct=direct)
                torch.cuda.synchronize()
                outs.append(yb)
            a, b = outs
            fin = bool(torch.isfinite(a).all())
            err = float((a - b).abs().max() / max(1.0, float(b.abs().max())))
            pad_ok = float(a[..., :8].abs().max()) == 0.0 and float(a[..., 8 + cout:].abs().max()) == 0.0
            ok = fin and err <= 3.2e-5 and pad_ok
            # where is the error? per-channel-block and per-row summary helps to localise an indexing bug
            info = ""
            if not ok:
                d = (a - b)[..., 8:8 + cout].abs()
                info = (f" | per 16-ch group max {[round(float(d[..., c:c + 16].max()), 4) for c in range(0, cout, 16)][:12]}"
                        f" | per image max {[round(float(d[bb].max()), 4) for bb in range(B)]}"
                        f" | rows with err {[int(v) for v in torch.nonzero(d.amax(dim=(0, 2, 3)) > 1e-3).flatten()[:12]]}"
                        f" | cols with err {[int(v) for v in torch.nonzero(d.amax(dim=(0, 1, 3)) > 1e-3).flatten()[:12]]}")
            print(f"case {i}: B{B} {H}x{W} {cin}->{cout} gs{gs} {sorted(kw)}: err {err:.3e} finite {fin} pad {pad_ok} "
                  f"{'OK' if ok else 'FAIL'} ({time.time() - t0:.1f}s){info}", flush=True)
            bad += 0 if ok else 1
        except Exception as e:          # noqa: BLE001
            print(f"case {i}: EXCEPTION {type(e).__name__}: {e}", flush=True)
            bad += 1
    print(f"{bad} failing cases", flush=True)
    return bad


SHAPES = {
    "c544_544": (8, 392, 518, 544, 544), "c768_768_L4": (8, 224, 296, 768, 768), "c768_256_L4": (8, 224, 296, 768, 256),
    "c544_32": (8, 392, 518, 544, 32), "c512_256_L4": (8, 224, 296, 512, 256), "c768_768_L3": (8, 112, 148, 768, 768),
    "c256_256_L4": (8, 224, 296, 256, 256), "c768_768_L2": (8, 56, 74, 768, 768), "c128_32": (8, 392, 518, 128, 32),
    "c256_256_L3": (8, 112, 148, 256, 256), "c256_128_L4": (8, 224, 296, 256, 128), "c512_256_L2": (8, 56, 74, 512, 256),
    "c256_256_B1": (1, 224, 296, 256, 256), "c768_768_L1": (8, 28, 37, 768, 768),
    # below the three-step form's channel threshold: fused kernel against the DIRECT kernel (the "three-step" column is the direct kernel there)
    "c64_32": (8, 392, 518, 64, 32), "c32_32": (8, 392, 518, 32, 32), "c32_256": (8, 196, 259, 32, 256), "c64_64_L4": (8, 224, 296, 64, 64),
}


def timing(only):
    os.environ["PF_WINOGRAD"] = "4"
    _hip_ops.refresh_env()
    os.environ["PF_WINOGRAD_MIN_PIXELS"] = "0"
    _hip_ops.refresh_env()
    for name, (B, H, W, cin, cout) in SHAPES.items():
        if only and name not in only:
            continue
        w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
        pw = pk.pack_conv(w, torch.zeros(cout), dtype=torch.float32).to(DEV)
        x = torch.randn(B, H, W, cin, device=DEV)
        y = torch.empty(B, H, W, cout, device=DEV)
        T = B * -(-H // 4) * -(-W // 4)
        fl_w, fl_d = 36 * 2.0 * T * cin * cout, 2.0 * B * H * W * 9 * cin * cout
        os.environ["PF_WINO_FUSED"] = "0"
        _hip_ops.refresh_env()
        os.environ["PF_WINO_SPLIT3"] = "0"
        _hip_ops.refresh_env()
        ms3 = ops.conv(x, pw, y, pad=1, act="relu", _timed=3)
        os.environ["PF_WINO_SPLIT3"] = "1"
        _hip_ops.refresh_env()
        ms3s = ops.conv(x, pw, y, pad=1, act="relu", _timed=3)
        line = (f"{name:14s} B{B} {H}x{W} {cin}->{cout}: three-step {ms3:8.3f} ms ({fl_d / ms3 / 1e9:6.1f} TF/s direct-eq), with the split-precision "
                f"GEMM {ms3s:8.3f} ms | fused")
        os.environ["PF_WINO_FUSED"] = "2"
        _hip_ops.refresh_env()
        best = None
        for gs, shp in ((1, 0), (4, 0), (8, 0), (16, 0), (64, 0), (8, 8), (8, 4)):
            os.environ["PF_WINO_GS"], os.environ["PF_WINO_SHAPE"] = str(gs), str(shp)
            _hip_ops.refresh_env()
            ms = ops.conv(x, pw, y, pad=1, act="relu", _timed=3)
            line += f" gs{gs}{'/sw' + str(shp) if shp else ''}: {ms:.3f}"
            if best is None or ms < best[0]:
                best = (ms, f"{gs}/sw{shp}")
        line += (f" | best gs{best[1]} {best[0]:.3f} ms = {fl_w / best[0] / 1e9:6.1f} TF/s executed ({fl_w / best[0] / 1e9 / 157.3:.3f} of f32 peak), "
                 f"{fl_d / best[0] / 1e9:6.1f} TF/s direct-eq, {ms3 / best[0]:.2f}x three-step")
        print(line, flush=True)
        del x, y, pw
        torch.cuda.empty_cache()


def decomp(names):
    """timing decomposition with the kernel's debug switches (results are wrong by construction)"""
    os.environ["PF_WINOGRAD"], os.environ["PF_WINOGRAD_MIN_PIXELS"], os.environ["PF_WINO_FUSED"] = "4", "0", "2"
    _hip_ops.refresh_env()
    os.environ["PF_WINO_GS"], os.environ["PF_WINO_SHAPE"] = "8", "0"
    _hip_ops.refresh_env()
    for name in names:
        B, H, W, cin, cout = SHAPES[name]
        w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
        pw = pk.pack_conv(w, torch.zeros(cout), dtype=torch.float32).to(DEV)
        x = torch.randn(B, H, W, cin, device=DEV)
        y = torch.empty(B, H, W, cout, device=DEV)
        line = f"{name}:"
        for dbg, what in ((0, "full"), (1, "U hot"), (2, "no transform"), (4, "no DMA"), (6, "no transform, no DMA"), (7, "U hot, no transform, no DMA"),
                          (8, "no MFMA in T waves"), (16, "no MFMA in D waves"), (24, "no MFMA at all"), (30, "barriers only"), (9, "U hot + no T MFMA"),
                          (3, "U hot, no transform")):
            os.environ["PF_WINO_DBG"] = str(dbg)
            _hip_ops.refresh_env()
            ms = ops.conv(x, pw, y, pad=1, act="relu", _timed=3)
            line += f"\n    dbg {dbg:2d} ({what}): {ms:.3f} ms"
        os.environ["PF_WINO_DBG"] = "0"
        _hip_ops.refresh_env()
        print(line, flush=True)


def timeline(name, blocks=(1, 5)):
    """s_memtime stamps of one block (debug build): per wave entry / prologue / per chunk (T waves: planes 0-3 done, transform done, planes
    done; DMA waves: DMA issued, planes done, DMA landed) / epilogue phases"""
    os.environ["PF_WINOGRAD"], os.environ["PF_WINOGRAD_MIN_PIXELS"], os.environ["PF_WINO_FUSED"] = "4", "0", "2"
    _hip_ops.refresh_env()
    os.environ["PF_WINO_GS"], os.environ["PF_WINO_SHAPE"] = "8", "0"
    _hip_ops.refresh_env()
    B, H, W, cin, cout = SHAPES[name]
    pw = pk.pack_conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.zeros(cout), dtype=torch.float32).to(DEV)
    x, y = torch.randn(B, H, W, cin, device=DEV), torch.empty(B, H, W, cout, device=DEV)
    nkc = cin // 8
    for blk in blocks:
        tb = torch.zeros(8 * 256 * 2, dtype=torch.float32, device=DEV)
        os.environ["PF_WINO_DBG"] = str(128 | (blk << 8))
        _hip_ops.refresh_env()
        ops.conv(x, pw, y, pad=1, act="relu", res2=tb.view(1, 1, 1, -1))
        ops.conv(x, pw, y, pad=1, act="relu", res2=tb.view(1, 1, 1, -1))
        torch.cuda.synchronize()
        t = tb.view(torch.int64).view(8, 256).cpu()
        t0 = int(t[:, 0][t[:, 0] > 0].min())
        rel = lambda v: int(v) - t0 if int(v) > 0 else -1
        print(f"{name} block {blk * 1000}: cycles relative to the first wave's entry (s_memtime ticks)")
        for c in (30, 31):
            print(f"  chunk {c}: absolute stamps (barrier exit, T: planes0-3 done / D: DMA issued, T: transform done / D: planes done, end before barrier)")
            base = int(t[0, 2 + 4 * c])
            for w in range(8):
                print(f"    wave {w}: " + "  ".join(f"{int(t[w, 2 + 4 * c + k]) - base:6d}" for k in range(4)))
        for w in (0, 4):
            iv = [int(t[w, 2 + 4 * (c + 1)]) - int(t[w, 2 + 4 * c]) for c in range(2, 58)]
            print(f"  wave {w}: mean interval chunks 2..57: {sum(iv) / len(iv):.0f} cycles")
            print(f"    prologue done {rel(t[w, 1])}, last chunk done {rel(t[w, 249])}, barrier {rel(t[w, 250])}, after write+barrier {rel(t[w, 251])}, {rel(t[w, 252])}, end {rel(t[w, 253])}")
    os.environ["PF_WINO_DBG"] = "0"
    _hip_ops.refresh_env()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(1 if check() else 0)
    if mode == "timeline":
        timeline(sys.argv[2] if len(sys.argv) > 2 else "c544_544")
        sys.exit(0)
    if mode == "decomp":
        decomp(sys.argv[2].split(",") if len(sys.argv) > 2 else ["c544_544", "c544_32"])
        sys.exit(0)
    if mode == "one":            # N launches of one shape (for rocprofv3 --pmc passes)
        os.environ["PF_WINOGRAD"], os.environ["PF_WINOGRAD_MIN_PIXELS"], os.environ["PF_WINO_FUSED"] = "4", "0", os.environ.get("PF_WINO_FUSED", "2")
        _hip_ops.refresh_env()
        B, H, W, cin, cout = SHAPES[sys.argv[2]]
        pw = pk.pack_conv(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.zeros(cout), dtype=torch.float32).to(DEV)
        x, y = torch.randn(B, H, W, cin, device=DEV), torch.empty(B, H, W, cout, device=DEV)
        print(sys.argv[2], ops.conv(x, pw, y, pad=1, act="relu", _timed=5), "ms")
        sys.exit(0)
    timing(sys.argv[2].split(",") if len(sys.argv) > 2 else None)
