"""Typed Python wrappers over the C ABI (include/pf_hip.h).

Every function takes torch tensors that live on the GPU, extracts raw device pointers / strides and
launches the hand-written HIP kernel on torch's *current* stream.  PyTorch is used for memory and
streams only.  There is no CPU path here: CPU tensors are rejected and a missing shared library
makes this module fail to import (see _lib.load).

Tensor conventions: activations are NHWC views ``[B,H,W,C]`` (or token matrices ``[M,C]``) with
unit channel stride; the pixel stride (``ld``) may exceed C when the tensor is a channel slice of a
concat buffer.  dtype float32 = exact mode, bfloat16 = fast mode; a few head tensors are always f32.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvParams, check
from .packing import PackedConv, winograd_applies

_L = _lib.load()

ACT = {"none": 0, None: 0, "relu": 1, "gelu": 2, "softplus": 3}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.PfError("patchfusion_amd.hip_ops: tensor is not on the GPU (there is no CPU path)")
    if t.device.index != torch.cuda.current_device():
        # kernels are launched on torch's CURRENT device/stream; a tensor of another GPU would be accessed through the
        # wrong context (and the >64 KiB LDS opt-in is per device): make the mismatch loud instead
        raise _lib.PfError(f"patchfusion_amd.hip_ops: tensor lives on cuda:{t.device.index} but the current device is "
                           f"cuda:{torch.cuda.current_device()} (wrap the call in torch.cuda.device(...))")
    return C.c_void_p(t.data_ptr())


def _dt(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise _lib.PfError(f"unsupported dtype {t.dtype}")


def _as4(t):
    if t.dim() == 2:
        return t.unsqueeze(0).unsqueeze(0)
    if t.dim() == 3:
        return t.unsqueeze(0)
    return t


def _ld(t):
    """pixel stride of a dense-in-(B,H,W) NHWC view"""
    t = _as4(t)
    B, H, W, Cc = t.shape
    assert t.stride(3) == 1, "channel stride must be 1"
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else max(Cc, t.stride(2)))
    if W > 1 and H > 1:
        assert t.stride(1) == W * ld, "rows must be dense"
    if B > 1:
        assert t.stride(0) == H * W * ld, "batches must be dense"
    return ld


_ENV = {}


def _env(name, default):
    """PF_* switch, resolved ONCE (first use after import / refresh_env) instead of on every library call: round 3 read 4-6 environment
    variables per convolution, ~20 % of the 22 us the Python host spent per call (tools/host_overhead.py)."""
    v = _ENV.get(name)
    if v is None:
        import os
        v = _ENV[name] = os.environ.get(name, default)
    return v


def refresh_env():
    """forget every cached PF_* switch and every cached dispatch decision (called by the engine build and by tests that flip switches)"""
    _ENV.clear()
    _CONV_CACHE.clear()


_CONV_CACHE = {}


def _bf16_pp_enabled():
    """plain bf16 linears through the split GEMM's ping-pong pipeline (pf_gemm_bf16_pp, 256 x 128 tiles): measured and NOT kept -- 0.095 /
    0.047 / 0.118 / 0.126 ms on the ViT-L qkv / proj / fc1 / fc2 shapes against 0.086 / 0.038 / 0.108 / 0.105 for the implicit-GEMM kernels
    (profiles/r3_bf16_pp_sweep.log): one product per fragment pair does not amortise the pipeline the way the split GEMM's six do.
    PF_BF16_PP=1 turns it on (A/B, tests/op_checks.py conv_bf16_pp)."""
    return _env("PF_BF16_PP", "0") == "1"


def _conv1x1_split3_wanted(M, pw):
    """float32 1x1 layer through csrc/conv1x1_split3.hip (persistent walk of 64 x 128 tiles, two blocks per CU) instead of the f32-MFMA implicit GEMM?
    PF_CONV1X1_SPLIT3: 0 = never, 2 = wherever the planes exist, 1 (default) = the measured rule (profiles/r6_conv1x1_split3.md: 1.23-1.55x on every layer
    of the pass with at least 64 output channels and two tiles per CU, incl. the 32 -> 96 / 128 linears of the first G2L level; 0.78x on 128 -> 32 and
    1.06x on 32 -> 32, where a 128-wide channel tile multiplies zeros)."""
    mode = _env("PF_CONV1X1_SPLIT3", "1")
    if mode == "0":
        return False
    if mode == "2":
        return True
    tiles = -(-M // 64) * -(-pw.cout // 128)
    return tiles >= 512 and pw.cout >= 64


def _split3_three_step(pw):
    """does the three-step form of this layer run its GEMM in split precision? (filters packed as three planes and PF_WINO_SPLIT3 != 0)"""
    return pw.wino_u3 is not None and _env("PF_WINO_SPLIT3", "1") != "0"


def _fused_wanted(B, H, W, pw):
    """Fused Winograd kernel (csrc/wino_fused.hip) or the three-step form for this call?  PF_WINO_FUSED (read per call): 0 = never fused,
    2 = fused wherever supported, 1 (default) = the measured rule (profiles/r3_wino_fused_layers.log, r3_three_step_split.log; 1x MI355X):
    the fused kernel needs at least two rounds of blocks, and it loses to the three-step form on layers with MANY OUTPUT CHANNELS -- each of
    the ceil(Cout/64) channel blocks repeats the input transform, while the batched GEMM of the three-step form runs near its peak there:
    with the split-precision GEMM from 256 output channels on (round 3: 512; 544->544 @ 8x392x518: 16.3 vs 22.5 ms; 768->768 @ 8x224x296: 8.4 vs 13.5),
    with the f32 GEMM only at 768+ -> 768+ (13.0 vs 13.5).  Below that the fused kernel wins 1.1x (768->256) ... 2.4x (-> 32 channels)."""
    mode = _env("PF_WINO_FUSED", "1")
    if mode == "0":
        return False
    if mode == "2":
        return True
    th, tw = -(-H // 4), -(-W // 4)
    nsuper = B * min(-(-th // 4) * -(-tw // 8), -(-th // 8) * -(-tw // 4))
    if pw.wino_u is None:
        many_out = False                                    # fused-only layer: the alternative is the direct kernel
    elif _split3_three_step(pw):
        # round 4 (persistent split GEMM on chunk-major planes, profiles/r4_fused_vs_three_step.md): three steps win 1.04 .. 1.9x from 256 output
        # channels on (768->256 @ 8x224x296: 3.87 vs 4.84 ms), the fused kernel keeps 256->128 (0.91x) and the 32-channel layers (0.4x)
        many_out = pw.cout >= 256
    else:
        many_out = pw.cin >= 768 and pw.cout >= 768
    return nsuper * -(-pw.cout // 64) >= 512 and not many_out


_WS = {}


def _workspace(device, nV, nM):
    """V / M workspaces of the three-step Winograd path, one growing pair per (device, stream): launches of one stream are ordered, so
    the pair can be reused by every layer of that stream instead of 2 x 8 GB of fresh allocations per call (round-2 advisor finding).
    Growth drops the old buffer BEFORE allocating the larger one (round-3 advisor finding: `v = None` only cleared the local name, the dict
    still held the old buffer, so old and new coexisted); release_workspaces() frees everything (model teardown / `_apply`)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    v, m = _WS.get(key, (None, None))
    if v is None or v.numel() < nV or m is None or m.numel() < nM:
        _WS.pop(key, None)
        if v is None or v.numel() < nV:
            v = None
            v = torch.empty(nV, dtype=torch.float32, device=device)
        if m is None or m.numel() < nM:
            m = None
            m = torch.empty(nM, dtype=torch.float32, device=device)
        _WS[key] = (v, m)
    return v, m


def release_workspaces():
    """free the per-(device, stream) Winograd arenas (at the headline layer 12 GB + 8 GB per stream); they are re-grown on demand.  Also drops the
    cached dispatch plans: they hold strong references to the packed layers (device weights + Winograd filter planes, several GB for ViT-L), so
    `del model; release_workspaces(); torch.cuda.empty_cache()` really returns the memory (round-4 advisor finding)"""
    _WS.clear()
    _CONV_CACHE.clear()


def workspace_bytes():
    return sum(v.numel() * 4 + m.numel() * 4 for v, m in _WS.values())


def _fused_group():
    """PF_WINO_GS: super-tiles per block group (L2 locality knob); PF_WINO_SHAPE: 0 = automatic, 8 | 4 = forced super-tile width (tuning)"""
    return ((int(_env("PF_WINO_GS", "8")) & 0xffff) | (int(_env("PF_WINO_SHAPE", "0")) << 16)
            | (int(_env("PF_WINO_DBG", "0")) << 20))     # PF_WINO_DBG: timing decomposition only (wrong results), see wino_fused.hip


def wino3_window(B, H, W, pw):
    """(window, nwin, nV, nM): a three-step split-precision Winograd layer over T = B ceil(H/4) ceil(W/4) tiles runs `window` tiles at a time (nwin
    windows, the last one ragged) through ONE arena pair of nV + nM float32 words (V: three bf16 planes; M: float32) -- csrc/winograd.hip run_split3.
    Winograd tiles are independent, so the numbers are identical for every window.  PF_WS_CAP_GB (default 2.5) is the largest V + M pair a layer may ask
    for; windows are multiples of 192 tiles (the GEMM's token tile).  The headline layer (544->544 @ 8 x 392 x 518: 11.9 + 8.0 GB per stream in one piece)
    runs in 8 windows of 2.3 GB.  Measured (profiles/r5_ws_cap.md, r5_subbatch_probe.md): the batched GEMM keeps its rate (0.555 of 2500/6 at 8 tiles of
    392 x 518 per launch, 0.551 at 2, 0.538 at 1), and the IMAGE pass gets faster -- 188.6 ms in one piece, 187.3 at 5 GB, 186.4 at 2.5 GB: the two tile
    streams interleave at a finer grain -- while the peak allocation falls from 62.8 to 31 GiB."""
    T = B * -(-H // 4) * -(-W // 4)
    per_tile = 36 * (pw.cin * 6 + pw.cout * 4)
    cap = float(_env("PF_WS_CAP_GB", "2.5")) * 2 ** 30
    window = max(int(cap // per_tile) // 192 * 192, 192)
    if window >= T:
        window = -(-T // 8) * 8
    nwin = -(-T // window)
    return window, nwin, (36 * window * pw.cin * 3 + 1) // 2, 36 * window * pw.cout


class HipOps:
    name = "hip"

    # ---------------- allocation ----------------
    @staticmethod
    def empty(shape, dtype, device):
        return torch.empty(shape, dtype=dtype, device=device)

    @staticmethod
    def zeros(shape, dtype, device):
        return torch.zeros(shape, dtype=dtype, device=device)

    # ---------------- conv / linear ----------------
    @staticmethod
    def _conv_plan(x, pw, y, stride, pad, act, relu_in, res, res2, direct):
        """everything about a conv call that does not change while shapes / strides / the packed layer stay the same: the argument checks, the filled
        pf_conv_params and the dispatch decision ('pp' | 'fused' | 'wino3' | 'wino' | 'direct').  Cached by HipOps.conv per (layer, layout)."""
        x4, y4 = _as4(x), _as4(y)
        B, H, W, _ = x4.shape
        OH = (H + 2 * pad - pw.KH) // stride + 1
        OW = (W + 2 * pad - pw.KW) // stride + 1
        s = pw.shuffle
        assert y4.shape[0] == B and y4.shape[1] == OH * s and y4.shape[2] == OW * s, (x4.shape, y4.shape, OH, OW, s)
        assert x4.shape[3] >= pw.cin and y4.shape[3] >= pw.cout // (s * s), (x4.shape, y4.shape, pw.cin, pw.cout)
        p = ConvParams()
        p.x, p.x_ld, p.B, p.H, p.W, p.Cin = x4.data_ptr(), _ld(x4), B, H, W, pw.cin
        p.w, p.w_rows, p.Kpad = pw.w.data_ptr(), pw.w.shape[0], pw.w.shape[1]
        p.bias = pw.bias.data_ptr() if pw.bias is not None else None
        p.scale = pw.scale.data_ptr() if pw.scale is not None else None
        p.res = res.data_ptr() if res is not None else None
        p.res_ld = _ld(res) if res is not None else 0
        p.res2 = res2.data_ptr() if res2 is not None else None
        p.res2_ld = _ld(res2) if res2 is not None else 0
        p.y, p.y_ld, p.OH, p.OW, p.Cout = y4.data_ptr(), _ld(y4), OH, OW, pw.cout
        p.KH, p.KW, p.stride, p.pad = pw.KH, pw.KW, stride, pad
        p.act, p.relu_in = ACT[act], int(bool(relu_in))
        p.dtype = _dt(x4)
        assert pw.w.dtype == x4.dtype, "weights must be packed in the activation dtype"
        p.out_f32 = int(y4.dtype == torch.float32 and x4.dtype != torch.float32)
        p.shuffle = s
        p.korder = pw.korder
        for t in (x4, y4, pw.w, res, res2):
            _p(t)
        f32_io = y4.dtype == torch.float32 and (res is None or res.dtype == torch.float32) and (res2 is None or res2.dtype == torch.float32)
        if (x4.dtype == torch.bfloat16 and pw.KH == 1 and pw.KW == 1 and stride == 1 and pad == 0 and s == 1 and not relu_in and not direct and
                B * H * W >= 2048 and pw.cin % 64 == 0 and pw.cin >= 512 and pw.cout >= 512 and _bf16_pp_enabled()):
            # bf16 linear layers with many token rows (ViT blocks, DPT projections): 256 x 128 ping-pong tiles (csrc/gemm_split3.hip, PLAIN)
            return "pp", p, None
        if (pw.w3 is not None and x4.dtype == torch.float32 and f32_io and stride == 1 and pad == 0 and s == 1 and not direct and
                _conv1x1_split3_wanted(B * H * W, pw)):
            # float32 1x1 layers with enough tokens: split-precision product on the bf16 matrix cores, x split in the kernel's loader
            return "s3_1x1", p, (_p(pw.w3), pw.w3.shape[2])
        wino = winograd_applies(pw, B * H * W, stride, pad, act) and not direct
        if wino and pw.wino_up is not None and _fused_wanted(B, H, W, pw) and _L.pf_conv_winograd_fused_supported(C.byref(p)):
            # fused F(4x4,3x3): one kernel, no V / M workspaces (csrc/wino_fused.hip)
            assert f32_io
            return "fused", p, (_p(pw.wino_up), pw.wino_up.shape[0], _fused_group())
        if wino and pw.wino_u is not None:
            # float32 3x3 layers: input transform -> (m+2)^2 GEMMs -> output transform (csrc/winograd.hip)
            m = pw.wino_m
            T = B * -(-H // m) * -(-W // m)
            a2 = (m + 2) ** 2
            assert f32_io
            if m == 4 and _split3_three_step(pw):
                # (split: V holds three bf16 planes = 6 bytes per element instead of 4; whole tile octets, csrc/winograd.hip)
                window, _, nV, nM = wino3_window(B, H, W, pw)
                return "wino3", p, (_p(pw.wino_u3), pw.wino_u3.shape[3], pw.wino_u3.shape[2] * 32, nV, nM, window)
            return "wino", p, (m, _p(pw.wino_u), pw.wino_u.shape[1], pw.wino_u.shape[2], a2 * T * pw.cin, a2 * T * pw.cout)
        return "direct", p, None          # (incl. fused-only layers below the block threshold)

    @staticmethod
    def _conv_exec(route, p, extra, device):
        if route == "direct":
            check(_L.pf_conv(C.byref(p), _stream()), "pf_conv")
        elif route == "s3_1x1":
            check(_L.pf_conv1x1_split3(C.byref(p), extra[0], extra[1], _stream()), "pf_conv1x1_split3")
        elif route == "fused":
            check(_L.pf_conv_winograd_fused(C.byref(p), extra[0], extra[1], extra[2], _stream()), "pf_conv_winograd_fused")
        elif route == "wino3":
            V, Mw = _workspace(device, extra[3], extra[4])
            check(_L.pf_conv_winograd_split3_windowed(C.byref(p), extra[0], extra[1], extra[2], C.c_void_p(V.data_ptr()), C.c_void_p(Mw.data_ptr()),
                                                      extra[5], _stream()), "pf_conv_winograd_split3_windowed")
        elif route == "wino":
            V, Mw = _workspace(device, extra[4], extra[5])
            check(_L.pf_conv_winograd(C.byref(p), extra[0], extra[1], extra[2], extra[3], C.c_void_p(V.data_ptr()), C.c_void_p(Mw.data_ptr()), _stream()),
                  "pf_conv_winograd")
        else:
            check(_L.pf_gemm_bf16_pp(C.byref(p), _stream()), "pf_gemm_bf16_pp")

    @staticmethod
    def conv(x, pw: PackedConv, y, stride=1, pad=0, act=None, relu_in=False, res=None, res2=None, _timed=None, _direct=None):
        """y = epi(conv(x)) through the kernel the dispatch rules pick (pf_conv / fused or three-step Winograd / bf16 ping-pong GEMM).
        The plan of a call -- checks, filled pf_conv_params, route -- is cached per (packed layer, tensor layouts, epilogue); a repeat call only
        refreshes the four data pointers (round-3 review: 22 us of Python per library call, most of it here)."""
        if _timed is None and _direct is None:
            key = (id(pw), x.shape, x.stride(), y.shape, y.stride(), stride, pad, act, relu_in, x.dtype, y.dtype, x.device,
                   None if res is None else (res.shape, res.stride(), res.dtype), None if res2 is None else (res2.shape, res2.stride(), res2.dtype))
            ent = _CONV_CACHE.get(key)
            if ent is None or ent[0] is not pw:
                if len(_CONV_CACHE) > 4096:
                    _CONV_CACHE.clear()
                ent = (pw,) + HipOps._conv_plan(x, pw, y, stride, pad, act, relu_in, res, res2, None)
                _CONV_CACHE[key] = ent
            else:
                p = ent[2]
                p.x, p.y = x.data_ptr(), y.data_ptr()
                if res is not None:
                    p.res = res.data_ptr()
                if res2 is not None:
                    p.res2 = res2.data_ptr()
            HipOps._conv_exec(ent[1], ent[2], ent[3], x.device)
            return y
        route, p, extra = HipOps._conv_plan(x, pw, y, stride, pad, act, relu_in, res, res2, _direct)
        if _timed is None:
            HipOps._conv_exec(route, p, extra, x.device)
            return y
        if route == "direct":
            ms = C.c_float(0)
            check(_L.pf_conv_timed(C.byref(p), int(_timed), C.byref(ms), _stream()), "pf_conv_timed")
            return ms.value
        if route == "fused":
            ms = C.c_float(0)
            check(_L.pf_conv_winograd_fused_timed(C.byref(p), extra[0], extra[1], extra[2], int(_timed), C.byref(ms), _stream()),
                  "pf_conv_winograd_fused_timed")
            return ms.value
        HipOps._conv_exec(route, p, extra, x.device)          # whole three-step layer / ping-pong GEMM: events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(int(_timed)):
            HipOps._conv_exec(route, p, extra, x.device)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / int(_timed)

    @staticmethod
    def gemm_planes_split3(V3, U3, Mw, T, cin, cout, iters=None):
        """the batched split-precision GEMM launch of a three-step Winograd layer alone, exactly as csrc/winograd.hip run_split3 issues it:
        CHUNK-MAJOR operands V3 bf16 [3, P, cin/32, T, 32], U3 bf16 [3, P, cin/32, rows, 32] (PackedConv.wino_u3), Mw float32 [P, T, cout]
        (P = 36 transform points there) -- or both ROW-major, V3 [3, P, T, cin], U3 [3, P, rows, cin] (4-D; tests); iters = None runs it once,
        else returns the average milliseconds of `iters` launches (HIP events on the launch stream; bench.py roofline)"""
        P = V3.shape[1]
        kmaj = V3.dim() == 5
        assert V3.dim() == U3.dim() and V3.dtype == U3.dtype == torch.bfloat16 and Mw.dtype == torch.float32 and U3.shape[:2] == (3, P)
        assert tuple(V3.shape) == ((3, P, cin // 32, T, 32) if kmaj else (3, P, T, cin)) and (U3.shape[2] * 32 if kmaj else U3.shape[3]) == cin
        assert V3.is_contiguous() and U3.is_contiguous() and Mw.is_contiguous() and Mw.numel() >= P * T * cout
        p = ConvParams()
        p.x, p.x_ld, p.B, p.H, p.W, p.Cin = V3.data_ptr(), cin, 1, 1, T, cin
        p.w, p.w_rows, p.Kpad = U3.data_ptr(), U3.shape[3] if kmaj else U3.shape[2], cin
        p.korder = 6 if kmaj else 0
        p.y, p.y_ld, p.OH, p.OW, p.Cout = Mw.data_ptr(), cout, 1, T, cout
        p.KH = p.KW = p.stride = 1
        p.act, p.shuffle, p.dtype, p.out_f32, p.batch = 0, 1, 1, 1, P
        p.x_bstride, p.w_bstride = V3.stride(0), U3.stride(0)
        for t in (V3, U3, Mw):
            _p(t)
        if iters is None:
            check(_L.pf_gemm_split3(C.byref(p), _stream()), "pf_gemm_split3")
            return Mw
        ms = C.c_float(0)
        check(_L.pf_gemm_split3_timed(C.byref(p), int(iters), C.byref(ms), _stream()), "pf_gemm_split3_timed")
        return ms.value

    @staticmethod
    def gemm_planes_split3_timed(V3, U3, Mw, T, cin, cout, iters):
        return HipOps.gemm_planes_split3(V3, U3, Mw, T, cin, cout, iters)

    @staticmethod
    def gemm_planes_timed(V, U, Mw, planes, T, cin, cout, iters):
        """time the batched GEMM launch of a Winograd layer alone (bench.py roofline): `planes` x ([T, cin] . [cout, cin]^T) through
        pf_conv with pf_conv_params.batch, exactly as csrc/winograd.hip issues it; V / Mw float32 workspaces, U [planes, rows, Kpad]"""
        assert V.dtype == U.dtype == Mw.dtype == torch.float32 and V.numel() >= planes * T * cin and Mw.numel() >= planes * T * cout
        p = ConvParams()
        p.x, p.x_ld, p.B, p.H, p.W, p.Cin = V.data_ptr(), cin, 1, 1, T, cin
        p.w, p.w_rows, p.Kpad = U.data_ptr(), U.shape[1], U.shape[2]
        p.y, p.y_ld, p.OH, p.OW, p.Cout = Mw.data_ptr(), cout, 1, T, cout
        p.KH = p.KW = p.stride = 1
        p.shuffle, p.dtype = 1, 0
        p.batch, p.x_bstride, p.w_bstride, p.y_bstride = planes, T * cin, U.shape[1] * U.shape[2], T * cout
        for t in (V, U, Mw):
            _p(t)
        ms = C.c_float(0)
        check(_L.pf_conv_timed(C.byref(p), int(iters), C.byref(ms), _stream()), "pf_conv_timed")
        return ms.value

    # ---------------- split-precision GEMM (exploratory "f32x3" mode) ----------------
    @staticmethod
    def split3(x, y3):
        """x float32 [M, K] (row stride >= K) -> y3 bfloat16 [3, M, K] planes"""
        assert x.dtype == torch.float32 and y3.dtype == torch.bfloat16 and y3.dim() == 3 and y3.shape[0] == 3 and y3.is_contiguous()
        M, K = x.shape
        check(_L.pf_split3(_p(x), x.stride(0), _p(y3), y3.stride(1), y3.stride(0), M, K, _stream()), "pf_split3")
        return y3

    @staticmethod
    def _conv_split3_plan(x3, pw, y, act, res, res2):
        """x3 bfloat16 [3, M, K] planes; pw from packing.pack_conv_split3; y float32 [M, N] or bfloat16 [3, M, N] planes (split output);
        res / res2 float32 [M, N].  float32-grade linear layer on the bf16 matrix cores (csrc/gemm_split3.hip)."""
        assert x3.dtype == torch.bfloat16 and x3.shape[0] == 3 and pw.w.dtype == torch.bfloat16
        split_out = y.dtype == torch.bfloat16
        p = ConvParams()
        korder = 0
        if x3.dim() == 4:                                     # chunk-major planes [3, K/32, M, 32]
            assert x3.is_contiguous() and x3.shape[3] == 32 and x3.shape[1] * 32 == pw.cin
            M = x3.shape[2]
            p.x, p.x_ld = x3.data_ptr(), pw.cin
            korder |= 2
        else:
            assert x3.dim() == 3 and x3.stride(2) == 1 and x3.shape[2] >= pw.cin
            M = x3.shape[1]
            p.x, p.x_ld = x3.data_ptr(), x3.stride(1)
        p.B, p.H, p.W, p.Cin = 1, 1, M, pw.cin
        p.x_bstride = x3.stride(0)
        if pw.w.dim() == 4:                                   # chunk-major weight planes [3, K/32, rows, 32] (packing.pack_conv_split3)
            assert pw.w.is_contiguous() and pw.w.shape[1] * 32 == pw.cin
            p.w, p.w_rows, p.Kpad, p.w_bstride = pw.w.data_ptr(), pw.w.shape[2], pw.cin, pw.w.stride(0)
            korder |= 4
        else:
            p.w, p.w_rows, p.Kpad, p.w_bstride = pw.w.data_ptr(), pw.w.shape[1], pw.w.shape[2], pw.w.stride(0)
        p.bias = pw.bias.data_ptr() if pw.bias is not None else None
        p.scale = pw.scale.data_ptr() if pw.scale is not None else None
        p.res = res.data_ptr() if res is not None else None
        p.res_ld = res.stride(-2) if res is not None else 0
        p.res2 = res2.data_ptr() if res2 is not None else None
        p.res2_ld = res2.stride(-2) if res2 is not None else 0
        if split_out and y.dim() == 4:                        # chunk-major output planes [3, N/32, M, 32]
            assert y.is_contiguous() and tuple(y.shape) == (3, pw.cout // 32, M, 32) and pw.cout % 32 == 0
            p.y, p.y_ld, p.y_bstride, p.out_f32 = y.data_ptr(), pw.cout, y.stride(0), 0
            korder |= 8
        elif split_out:
            assert y.dim() == 3 and y.shape[0] == 3 and y.shape[1] == M and y.stride(2) == 1
            p.y, p.y_ld, p.y_bstride, p.out_f32 = y.data_ptr(), y.stride(1), y.stride(0), 0
        else:
            assert y.dtype == torch.float32 and y.stride(-1) == 1
            p.y, p.y_ld, p.out_f32 = y.data_ptr(), y.stride(-2), 1
        p.OH, p.OW, p.Cout = 1, M, pw.cout
        p.KH = p.KW = p.stride = 1
        p.act, p.shuffle, p.dtype, p.korder = ACT[act], 1, 1, korder
        for t in (x3, y, pw.w, res, res2):
            _p(t)                                             # (device / layout checks of every tensor handed to the library)
        for t in (res, res2):
            assert t is None or (t.dtype == torch.float32 and t.shape[-1] >= pw.cout and t.shape[-2] == M), "residuals are float32 [M, >= N]"
        assert y.dim() == 4 or (y.shape[-1] >= pw.cout and y.shape[-2] == M)
        return p

    @staticmethod
    def conv_split3(x3, pw: PackedConv, y, act=None, res=None, res2=None, _timed=None):
        """x3 bfloat16 planes, row-major [3, M, K] or chunk-major [3, K/32, M, 32]; pw from packing.pack_conv_split3; y float32 [M, N] or bfloat16
        planes ([3, M, N] / chunk-major [3, N/32, M, 32]: split output for a following split GEMM); res / res2 float32 [M, N].  float32-grade linear
        layer on the bf16 matrix cores (csrc/gemm_split3.hip).  Like HipOps.conv, the checked and filled pf_conv_params of a call is cached per
        (layer, layouts); a repeat call refreshes the data pointers only."""
        key = (id(pw), x3.shape, x3.stride(), y.shape, y.stride(), y.dtype, act, x3.device,
               None if res is None else (res.shape, res.stride()), None if res2 is None else (res2.shape, res2.stride()))
        ent = _CONV_CACHE.get(key)
        if ent is None or ent[0] is not pw:
            if len(_CONV_CACHE) > 4096:
                _CONV_CACHE.clear()
            ent = (pw, HipOps._conv_split3_plan(x3, pw, y, act, res, res2))
            _CONV_CACHE[key] = ent
        else:
            p = ent[1]
            p.x, p.y = x3.data_ptr(), y.data_ptr()
            if res is not None:
                p.res = res.data_ptr()
            if res2 is not None:
                p.res2 = res2.data_ptr()
        p = ent[1]
        if _timed is not None:
            ms = C.c_float(0)
            check(_L.pf_gemm_split3_timed(C.byref(p), int(_timed), C.byref(ms), _stream()), "pf_gemm_split3_timed")
            return ms.value
        check(_L.pf_gemm_split3(C.byref(p), _stream()), "pf_gemm_split3")
        return y

    # ---------------- ViT ----------------
    @staticmethod
    def patch_im2col(img, out):
        B, _, H, W = img.shape
        assert img.dtype == torch.float32 and img.is_contiguous()
        check(_L.pf_patch_im2col(_p(img), B, H, W, _p(out), out.stride(0), _dt(out), _stream()), "pf_patch_im2col")

    @staticmethod
    def assemble_tokens(emb, tokens, cls, pos):
        B, S, D = tokens.shape
        assert emb.is_contiguous() and tokens.is_contiguous()
        check(_L.pf_assemble_tokens(_p(emb), _p(tokens), _p(cls), _p(pos), B, S, D, _dt(tokens), _stream()), "pf_assemble_tokens")

    @staticmethod
    def layernorm(x, y, g, b, eps, batches=1, in_rows_per_batch=None, in_row_offset=0, out_rows_per_batch=None):
        D = x.shape[-1]
        rows = x.numel() // D
        if in_rows_per_batch is None:
            in_rows_per_batch = out_rows_per_batch = rows
            batches = 1
        check(_L.pf_layernorm(_p(x), x.stride(-2), _p(y), y.stride(-2), _p(g), _p(b), float(eps), batches, in_rows_per_batch,
                              in_row_offset, out_rows_per_batch, D, _dt(x), _stream()), "pf_layernorm")

    @staticmethod
    def layernorm_split3(x, y3, g, b, eps):
        """LayerNorm of float32 rows x [M, D], output as three bfloat16 planes (input of ops.conv_split3): y3 [3, M, D] row-major or
        [3, D/32, M, 32] chunk-major"""
        assert x.dtype == torch.float32 and y3.dtype == torch.bfloat16 and y3.shape[0] == 3
        M, D = x.shape
        if y3.dim() == 4:
            assert y3.is_contiguous() and tuple(y3.shape) == (3, D // 32, M, 32)
            check(_L.pf_layernorm_split3(_p(x), x.stride(0), _p(y3), D, y3.stride(0), 1, _p(g), _p(b), float(eps), M, D, _stream()),
                  "pf_layernorm_split3")
            return y3
        assert y3.dim() == 3 and y3.stride(2) == 1 and y3.shape[1] == M and y3.shape[2] == D
        check(_L.pf_layernorm_split3(_p(x), x.stride(0), _p(y3), y3.stride(1), y3.stride(0), 0, _p(g), _p(b), float(eps), M, D, _stream()),
              "pf_layernorm_split3")
        return y3

    @staticmethod
    def vit_attention(qkv, out, B, S, heads):
        """qkv [B*S, 3*D] -> out [B*S, D]; head_dim must be 64.  float32 qkv with a bfloat16 out [3, B*S, D]: the output is written as
        the three split planes of the projection GEMM (ops.conv_split3)."""
        kmaj = int(out.dim() == 4)                            # output planes chunk-major [3, D/32, B*S, 32] (input of the projection GEMM)
        if qkv.dim() == 3:
            # the QKV GEMM's output as three bf16 planes [3, B*S, 3*D]: attention entirely in split precision (csrc/vit.hip)
            D = qkv.shape[2] // 3
            assert qkv.dtype == out.dtype == torch.bfloat16 and tuple(qkv.shape) == (3, B * S, 3 * D)
            assert tuple(out.shape) == ((3, D // 32, B * S, 32) if kmaj else (3, B * S, D))
            assert D == heads * 64 and qkv.is_contiguous() and out.is_contiguous()
            # PF_ATTN_V2 (default 1): csrc/attn_split3.hip (LDS-DMA tiles, transposing V reads); PF_ATTN_QW = 16 / 32 forces the queries per wave
            # (0: by launch size), PF_ATTN_SCHED = 1 the two-phase kernel (bit-identical to version 1) / 2 the software-pipelined kernel (0: default)
            if _env("PF_ATTN_V2", "1") != "0":
                check(_L.pf_vit_attention_split3_v2(_p(qkv), qkv.stride(0), _p(out), out.stride(0), kmaj, B, S, heads, int(_env("PF_ATTN_QW", "0")),
                                                    int(_env("PF_ATTN_SCHED", "0")), _stream()), "pf_vit_attention_split3_v2")
                return
            check(_L.pf_vit_attention_split3(_p(qkv), qkv.stride(0), _p(out), out.stride(0), kmaj, B, S, heads, _stream()), "pf_vit_attention_split3")
            return
        D = qkv.shape[1] // 3
        assert D == heads * 64 and qkv.is_contiguous() and out.is_contiguous()
        if qkv.dtype == torch.float32 and out.dtype == torch.bfloat16:
            assert tuple(out.shape) == ((3, D // 32, B * S, 32) if kmaj else (3, B * S, D))
            check(_L.pf_vit_attention_qkv_split3(_p(qkv), _p(out), out.stride(0), kmaj, B, S, heads, _stream()), "pf_vit_attention_qkv_split3")
            return
        if qkv.dtype == torch.float32 and _env("PF_ATTN_QKV", "1") != "0":
            # f32: the attention kernel reads q / k / v rows straight out of the QKV GEMM's output (csrc/vit.hip, version 2)
            check(_L.pf_vit_attention_qkv(_p(qkv), _p(out), B, S, heads, 0, _stream()), "pf_vit_attention_qkv")
            return
        Sp = (S + 63) // 64 * 64
        q = torch.empty((B, heads, S, 64), dtype=qkv.dtype, device=qkv.device)
        k = torch.empty_like(q)
        vt = torch.empty((B, heads, 64, Sp), dtype=qkv.dtype, device=qkv.device)
        check(_L.pf_qkv_split(_p(qkv), B, S, heads, _p(q), _p(k), _p(vt), Sp, 0.125, _dt(qkv), _stream()), "pf_qkv_split")
        check(_L.pf_vit_attention(_p(q), _p(k), _p(vt), _p(out), B, S, Sp, heads, _dt(qkv), _stream()), "pf_vit_attention")

    # ---------------- Swin / G2L ----------------
    @staticmethod
    def swin_ln_partition(x, xw, g, b, eps, shift):
        B, H, W, Cc = x.shape
        assert xw.is_contiguous()
        check(_L.pf_swin_ln_partition(_p(x), _ld(x), _p(xw), _p(g), _p(b), float(eps), B, H, W, Cc, shift, _dt(x), _stream()),
              "pf_swin_ln_partition")

    @staticmethod
    def swin_window_attention(qkv, out, bias_table, B, Hp, Wp, Cc, heads, shift):
        assert qkv.is_contiguous() and out.is_contiguous() and bias_table.is_contiguous()
        check(_L.pf_swin_window_attention(_p(qkv), _p(out), _p(bias_table), B, Hp, Wp, Cc, heads, shift, _dt(qkv), _stream()),
              "pf_swin_window_attention")

    @staticmethod
    def swin_unpartition_add(proj, shortcut, y, shift):
        B, H, W, Cc = shortcut.shape
        check(_L.pf_swin_unpartition_add(_p(proj), _p(shortcut), _ld(shortcut), _p(y), _ld(y), B, H, W, Cc, shift, _dt(y), _stream()),
              "pf_swin_unpartition_add")

    @staticmethod
    def add_rowwise(x, pos):
        """x [B,T,C] += pos [T,C] (float32)"""
        B, T, Cc = x.shape
        check(_L.pf_add_rowwise(_p(x), x.stride(1), _p(pos), B, T, Cc, _dt(x), _stream()), "pf_add_rowwise")

    # ---------------- image ops ----------------
    @staticmethod
    def resize(x, y, add=None, dtype=None):
        x4, y4 = _as4(x), _as4(y)
        B, H, W, Cc = x4.shape
        _, OH, OW, _ = y4.shape
        cd = dtype if dtype is not None else (x4.dtype if x4.dtype != torch.float32 else y4.dtype)
        code = 1 if cd == torch.bfloat16 else 0
        in_f32 = int(x4.dtype == torch.float32)
        out_f32 = int(y4.dtype == torch.float32)
        if add is not None:
            assert add.dtype == y4.dtype
        check(_L.pf_resize_bilinear(_p(x4), _ld(x4), B, H, W, Cc, _p(y4), _ld(y4), OH, OW, _p(add), _ld(add) if add is not None else 0,
                                    in_f32, out_f32, code, _stream()), "pf_resize_bilinear")

    @staticmethod
    def resize_concat(xs, y):
        """y[..., c0:c0+C_i] = bilinear(xs[i]) for 2 or 3 NHWC sources of the compute dtype (one launch, whole rows of y)"""
        n = len(xs)
        y4 = _as4(y)
        B, OH, OW, Ct = y4.shape
        assert 2 <= n <= 3 and sum(x.shape[-1] for x in xs) <= Ct and all(x.dtype == y4.dtype and x.shape[0] == B for x in xs)
        ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        arr = lambda v: (C.c_int * n)(*v)
        for t in list(xs) + [y4]:
            _p(t)
        check(_L.pf_resize_concat(ptrs, arr([_ld(x) for x in xs]), arr([x.shape[1] for x in xs]), arr([x.shape[2] for x in xs]),
                                  arr([x.shape[3] for x in xs]), n, B, _p(y4), _ld(y4), OH, OW, _dt(y4), _stream()), "pf_resize_concat")

    @staticmethod
    def crop_resize(img, boxes, out):
        """img [3,H,W] f32; boxes int32 [P,4] (x0,y0,x1,y1) on device; out [P,3,oh,ow] f32"""
        Cc, H, W = img.shape
        P, _, oh, ow = out.shape
        assert img.is_contiguous() and out.is_contiguous() and boxes.dtype == torch.int32
        check(_L.pf_crop_resize_planar(_p(img), Cc, H, W, _p(boxes), P, _p(out), oh, ow, _stream()), "pf_crop_resize_planar")

    @staticmethod
    def roi_align_depth(feat, rois, y, spatial_scale):
        """single-channel planar float map: feat [Bf,1,H,W] f32 -> y [K,1,oh,ow] f32"""
        Bf, _, H, W = feat.shape
        K, _, oh, ow = y.shape
        assert feat.is_contiguous() and y.is_contiguous() and feat.dtype == y.dtype == torch.float32
        check(_L.pf_roi_align(_p(feat), 1, Bf, H, W, 1, _p(rois), K, _p(y), 1, oh, ow, float(spatial_scale), 1, 1, 0, _stream()),
              "pf_roi_align")

    @staticmethod
    def roi_align(feat, rois, y, spatial_scale, dtype=None):
        """feat [Bf,H,W,C] NHWC; rois f32 [K,5]; y [K,oh,ow,C]"""
        Bf, H, W, Cc = feat.shape
        K, oh, ow, _ = y.shape
        code = 1 if (dtype or feat.dtype) == torch.bfloat16 else 0
        check(_L.pf_roi_align(_p(feat), _ld(feat), Bf, H, W, Cc, _p(rois), K, _p(y), _ld(y), oh, ow, float(spatial_scale),
                              int(feat.dtype == torch.float32), int(y.dtype == torch.float32), code, _stream()), "pf_roi_align")

    @staticmethod
    def maxpool2(x, y):
        B, H, W, Cc = x.shape
        check(_L.pf_maxpool2(_p(x), _ld(x), B, H, W, Cc, _p(y), _ld(y), _dt(x), _stream()), "pf_maxpool2")

    @staticmethod
    def copy_channels(x, y):
        x4, y4 = _as4(x), _as4(y)
        B, H, W, Cc = x4.shape
        code = 1 if torch.bfloat16 in (x4.dtype, y4.dtype) else 0
        check(_L.pf_copy_channels(_p(x4), _ld(x4), _p(y4), _ld(y4), B * H * W, Cc, int(x4.dtype == torch.float32),
                                  int(y4.dtype == torch.float32), code, _stream()), "pf_copy_channels")

    @staticmethod
    def pack_fusion_input(cdepth, fdepth, crops, y):
        B, h, w, c8 = y.shape
        assert c8 == 8 and y.is_contiguous() and cdepth.is_contiguous() and fdepth.is_contiguous() and crops.is_contiguous()
        check(_L.pf_pack_fusion_input(_p(cdepth), _p(fdepth), _p(crops), _p(y), B, h, w, _dt(y), _stream()), "pf_pack_fusion_input")

    @staticmethod
    def copy_plane(src, dst):
        """dst[...] = src[:, 0] for float32 depth planes (device-to-device, async on the current stream)"""
        dst.copy_(src[:, 0], non_blocking=True)

    @staticmethod
    def nhwc_to_nchw(x, Cc=None):
        B, H, W, Ct = x.shape
        Cc = Cc or Ct
        y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
        code = 1 if x.dtype == torch.bfloat16 else 0
        check(_L.pf_nhwc_to_nchw_f32(_p(x), _ld(x), _p(y), B, H, W, Cc, int(x.dtype == torch.float32), code, _stream()), "pf_nhwc_to_nchw_f32")
        return y

    # ---------------- metric-bins head ----------------
    @staticmethod
    def attractor(A, n_attr, b_prev, out, a_stride=1, a_eps=0.0, attractor_type="inv", kind="mean"):
        """A [B,h,w,>=n_attr*a_stride] f32 ; b_prev [B,hp,wp,n_bins] f32 ; out [B,h,w,n_bins] f32 (include/pf_hip.h: pf_attractor)"""
        B, h, w, nb = out.shape
        _, hp, wp, _ = b_prev.shape
        assert A.dtype == b_prev.dtype == out.dtype == torch.float32 and b_prev.is_contiguous() and out.is_contiguous()
        assert A.shape[-1] >= (n_attr - 1) * a_stride + 1 and kind in ("mean", "sum")
        check(_L.pf_attractor(_p(A), _ld(A), n_attr, int(a_stride), float(a_eps), int(attractor_type == "exp"), int(kind == "sum"),
                              _p(b_prev), hp, wp, _p(out), B, h, w, nb, _stream()), "pf_attractor")

    @staticmethod
    def seed_bin_centers(x, out, min_depth, max_depth, bounded, normalize):
        """x [B,h,w,>=n_bins] f32 (relu / softplus MLP output) -> out [B,h,w,n_bins] f32 (include/pf_hip.h: pf_seed_bin_centers)"""
        assert x.dtype == out.dtype == torch.float32 and out.is_contiguous() and x.shape[:3] == out.shape[:3]
        check(_L.pf_seed_bin_centers(_p(x), _ld(x), _p(out), out.numel() // out.shape[-1], out.shape[-1], float(min_depth), float(max_depth),
                                     int(bool(bounded)), int(bool(normalize)), _stream()), "pf_seed_bin_centers")
        return out

    @staticmethod
    def bounded_bin_centers(b, out, min_depth, max_depth):
        """attractor.py:132-135: out = clip(sort((max - min) * b + min)) along the bins; b, out [B,h,w,n_bins] f32 contiguous"""
        assert b.dtype == out.dtype == torch.float32 and b.is_contiguous() and out.is_contiguous() and b.shape == out.shape
        check(_L.pf_bounded_bin_centers(_p(b), _p(out), out.numel() // out.shape[-1], out.shape[-1], float(min_depth), float(max_depth),
                                        _stream()), "pf_bounded_bin_centers")
        return out

    @staticmethod
    def logbinom_depth(pt, centers, depth, min_temp, max_temp):
        """pt [B,h,w,>=4] f32 (softplus applied); centers [B,hc,wc,n_bins] f32; depth [B,h,w] f32"""
        B, h, w = depth.shape
        _, hc, wc, nb = centers.shape
        assert pt.dtype == centers.dtype == depth.dtype == torch.float32 and centers.is_contiguous() and depth.is_contiguous()
        check(_L.pf_logbinom_depth(_p(pt), _ld(pt), _p(centers), hc, wc, _p(depth), B, h, w, nb, float(min_temp), float(max_temp), _stream()),
              "pf_logbinom_depth")

    @staticmethod
    def bins_tail(clb, emb, tw, centers, depth, min_temp, max_temp):
        """clb [B,H,W,ctot] f32 (channels [0,32) = last, [160,168) = rel when tw.nq == 11; the embedding slice is NOT read); emb [B,he,we,128]
        f32; tw packing.BinsTail; centers [B,hc,wc,64] f32; depth [B,H,W] f32"""
        B, H, W, _ = clb.shape
        _, he, we, ce = emb.shape
        _, hc, wc, nb = centers.shape
        assert clb.dtype == emb.dtype == centers.dtype == depth.dtype == torch.float32 and ce == 128 and nb == 64
        assert emb.is_contiguous() and centers.is_contiguous() and depth.is_contiguous() and clb.stride(3) == 1
        check(_L.pf_bins_tail(_p(clb), _ld(clb), tw.rel_off, _p(emb), he, we, _p(tw.w0f), _p(tw.b0), _p(tw.w2), _p(tw.b2), _p(centers), hc, wc,
                              _p(depth), B, H, W, tw.nq, float(min_temp), float(max_temp), _stream()), "pf_bins_tail")

    # ---------------- stitching ----------------
    @staticmethod
    def stitch_init(pred, count, depth, mask, yx):
        MH, MW = pred.shape
        P, ph, pw = depth.shape
        assert depth.is_contiguous() and mask.is_contiguous() and yx.dtype == torch.int32
        check(_L.pf_stitch_init(_p(pred), _p(count), MH, MW, _p(depth), _p(mask), _p(yx), P, ph, pw, _stream()), "pf_stitch_init")

    @staticmethod
    def stitch_finish_init(avg, pred, count):
        check(_L.pf_stitch_finish_init(_p(avg), _p(pred), _p(count), avg.numel(), _stream()), "pf_stitch_finish_init")

    @staticmethod
    def stitch_update(avg, count, depth, mask, y0, x0):
        MH, MW = avg.shape
        dh, dw = depth.shape
        ph, pw = mask.shape
        assert depth.is_contiguous() and mask.is_contiguous()
        check(_L.pf_stitch_update(_p(avg), _p(count), MH, MW, _p(depth), dh, dw, _p(mask), int(y0), int(x0), ph, pw, _stream()), "pf_stitch_update")

    @staticmethod
    def resize_nearest_f32(x, y):
        check(_L.pf_resize_nearest_f32(_p(x), x.shape[0], x.shape[1], _p(y), y.shape[0], y.shape[1], _stream()), "pf_resize_nearest_f32")

    @staticmethod
    def resize_bilinear_f32(x, y):
        check(_L.pf_resize_bilinear_f32(_p(x), x.shape[0], x.shape[1], _p(y), y.shape[0], y.shape[1], _stream()), "pf_resize_bilinear_f32")

    # ---------------- input / output side (io.hip) ----------------
    @staticmethod
    def u8_bicubic_to_f32(img_u8, out, reverse_channels=False):
        """img_u8 [H,W,3] uint8 -> out [3,OH,OW] float32 (value/255, bicubic align_corners=True evaluated in double)."""
        assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_contiguous()
        assert out.dtype == torch.float32 and out.dim() == 3 and out.shape[0] == 3 and out.is_contiguous()
        check(_L.pf_u8_bicubic_to_f32(_p(img_u8), img_u8.shape[0], img_u8.shape[1], int(bool(reverse_channels)), _p(out), out.shape[1],
                                      out.shape[2], _stream()), "pf_u8_bicubic_to_f32")
        return out

    @staticmethod
    def percentiles(x, q0, q1, invalid_val=None, out=None, invalid_mask=None):
        """Exact np.percentile(x[x != invalid_val], [q0, q1]) (linear) of a float32 tensor -> device float32 [2];
        invalid_mask (uint8, same numel, non-zero = excluded) replaces the value test when given (color.py:121-122)."""
        assert x.dtype == torch.float32 and x.is_contiguous()
        if invalid_mask is not None:
            assert invalid_mask.dtype == torch.uint8 and invalid_mask.numel() == x.numel() and invalid_mask.is_contiguous()
        ws = torch.empty(_L.pf_percentile_workspace_bytes(), dtype=torch.uint8, device=x.device)
        out = torch.empty(2, dtype=torch.float32, device=x.device) if out is None else out
        check(_L.pf_percentiles_f32(_p(x), x.numel(), float(invalid_val if invalid_val is not None else 0.0), int(invalid_val is not None),
                                    _p(invalid_mask), float(q0), float(q1), _p(out), _p(ws), _stream()), "pf_percentiles_f32")
        return out

    @staticmethod
    def colorize(depth, vmin_vmax, lut_rgba, N, invalid_val, background_rgba, out, invalid_mask=None):
        assert depth.dtype == torch.float32 and depth.is_contiguous() and vmin_vmax.dtype == torch.float32 and vmin_vmax.numel() == 2
        assert lut_rgba.dtype == torch.uint8 and lut_rgba.shape == (N + 3, 4) and lut_rgba.is_contiguous()
        assert out.dtype == torch.uint8 and out.numel() == depth.numel() * 4 and out.is_contiguous()
        if invalid_mask is not None:
            assert invalid_mask.dtype == torch.uint8 and invalid_mask.numel() == depth.numel() and invalid_mask.is_contiguous()
        r, g, b, a = (int(v) & 255 for v in background_rgba)
        check(_L.pf_colorize_f32(_p(depth), depth.numel(), _p(vmin_vmax), _p(lut_rgba), int(N), float(invalid_val if invalid_val is not None else 0.0),
                                 int(invalid_val is not None), _p(invalid_mask), r | (g << 8) | (b << 16) | (a << 24), _p(out), _stream()),
              "pf_colorize_f32")
        return out

    @staticmethod
    def depth_to_u16(depth, out, scale=256.0):
        assert depth.dtype == torch.float32 and depth.is_contiguous() and out.dtype == torch.uint16 and out.numel() == depth.numel()
        check(_L.pf_depth_to_u16(_p(depth), depth.numel(), float(scale), _p(out), _stream()), "pf_depth_to_u16")
        return out

    @staticmethod
    def silog_loss(pred, target, min_depth, max_depth, beta=0.15):
        """SILogLoss forward (losses.py:15-62): pred, target float32 of equal shape -> device float32 scalar tensor"""
        assert pred.dtype == target.dtype == torch.float32 and pred.shape == target.shape and pred.is_contiguous() and target.is_contiguous()
        ws = torch.empty(3, dtype=torch.float64, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        check(_L.pf_silog_loss(_p(pred), _p(target), pred.numel(), float(min_depth), float(max_depth), float(beta), _p(ws), _p(loss),
                               _stream()), "pf_silog_loss")
        return loss

    @staticmethod
    def depth_metrics(gt, pred, edges, min_depth, max_depth, crop, out13, additional_mask=None):
        """gt [H,W], pred [h,w] float32, edges [H,W] float32 or None, crop = (y0, y1, x0, x1), additional_mask [H,W] uint8 or None
        (zero = excluded) -> out13 (device float64 [13])."""
        assert gt.dtype == torch.float32 and pred.dtype == torch.float32 and gt.dim() == 2 and pred.dim() == 2
        assert gt.is_contiguous() and pred.is_contiguous() and out13.dtype == torch.float64 and out13.numel() == 13
        if edges is not None:
            assert edges.dtype == torch.float32 and edges.shape == gt.shape and edges.is_contiguous()
        if additional_mask is not None:
            assert additional_mask.dtype == torch.uint8 and additional_mask.shape == gt.shape and additional_mask.is_contiguous()
        y0, y1, x0, x1 = (int(v) for v in crop)
        check(_L.pf_depth_metrics(_p(gt), gt.shape[0], gt.shape[1], _p(pred), pred.shape[0], pred.shape[1], _p(edges), _p(additional_mask),
                                  float(min_depth), float(max_depth), y0, y1, x0, x1, _p(out13), _stream()), "pf_depth_metrics")
        return out13


ops = HipOps()
