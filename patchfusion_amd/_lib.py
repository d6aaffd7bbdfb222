"""ctypes binding of libpf_hip.so (the C ABI declared in include/pf_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent the
import of :mod:`patchfusion_amd.hip_ops` raises.  Build with ``python __graft_entry__.py`` or
``make -C patchfusion_amd/csrc``.
"""
import ctypes as C
import os

# Load order matters: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 and this library needs the SAME HIP runtime that
# owns torch's streams and device pointers.  With torch imported first the dynamic linker resolves libpf_hip.so's libamdhip64.so.7
# to the copy torch already loaded; dlopen-ing libpf_hip.so first would pull /opt/rocm's runtime into the process next to torch's,
# and every launch on a torch stream then fails (seen as "pf_patch_im2col failed (status 2)" when build() ran before smoke()).
import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PF_LIB_PATH") or os.path.join(_HERE, "libpf_hip.so")   # PF_LIB_PATH: kernel-tuning builds only

vp, ci, cf, cl = C.c_void_p, C.c_int, C.c_float, C.c_long


class ConvParams(C.Structure):
    _fields_ = [
        ("x", vp), ("x_ld", ci), ("B", ci), ("H", ci), ("W", ci), ("Cin", ci),
        ("w", vp), ("w_rows", ci), ("Kpad", ci),
        ("bias", vp), ("scale", vp),
        ("res", vp), ("res_ld", ci), ("res2", vp), ("res2_ld", ci),
        ("y", vp), ("y_ld", ci), ("OH", ci), ("OW", ci), ("Cout", ci),
        ("KH", ci), ("KW", ci), ("stride", ci), ("pad", ci),
        ("act", ci), ("relu_in", ci), ("out_f32", ci), ("shuffle", ci), ("dtype", ci), ("korder", ci),
        ("batch", ci), ("x_bstride", C.c_long), ("w_bstride", C.c_long), ("y_bstride", C.c_long),
    ]


# name -> argtypes (every function returns int status except pf_last_error / pf_version)
SIGNATURES = {
    "pf_conv": [C.POINTER(ConvParams), vp],
    "pf_conv_timed": [C.POINTER(ConvParams), ci, C.POINTER(cf), vp],
    "pf_patch_im2col": [vp, ci, ci, ci, vp, ci, ci, vp],
    "pf_assemble_tokens": [vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "pf_layernorm": [vp, ci, vp, ci, vp, vp, cf, ci, ci, ci, ci, ci, ci, vp],
    "pf_qkv_split": [vp, ci, ci, ci, vp, vp, vp, ci, cf, ci, vp],
    "pf_vit_attention": [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
    "pf_vit_attention_qkv": [vp, vp, ci, ci, ci, ci, vp],
    "pf_swin_ln_partition": [vp, ci, vp, vp, vp, cf, ci, ci, ci, ci, ci, ci, vp],
    "pf_swin_window_attention": [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    "pf_swin_unpartition_add": [vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    "pf_add_rowwise": [vp, ci, vp, ci, ci, ci, ci, vp],
    "pf_resize_bilinear": [vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, vp, ci, ci, ci, ci, vp],
    "pf_resize_concat": [C.POINTER(vp), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), ci, ci, vp, ci, ci, ci, ci, vp],
    "pf_crop_resize_planar": [vp, ci, ci, ci, vp, ci, vp, ci, ci, vp],
    "pf_roi_align": [vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, ci, cf, ci, ci, ci, vp],
    "pf_maxpool2": [vp, ci, ci, ci, ci, ci, vp, ci, ci, vp],
    "pf_copy_channels": [vp, ci, vp, ci, cl, ci, ci, ci, ci, vp],
    "pf_pack_fusion_input": [vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "pf_nhwc_to_nchw_f32": [vp, ci, vp, ci, ci, ci, ci, ci, ci, vp],
    "pf_conv_winograd": [C.POINTER(ConvParams), ci, vp, ci, ci, vp, vp, vp],
    "pf_conv_winograd_split3": [C.POINTER(ConvParams), vp, ci, ci, vp, vp, vp],
    "pf_conv_winograd_split3_windowed": [C.POINTER(ConvParams), vp, ci, ci, vp, vp, C.c_long, vp],
    "pf_gemm_split3": [C.POINTER(ConvParams), vp],
    "pf_gemm_split3_ex": [C.POINTER(ConvParams), ci, vp],
    "pf_conv1x1_split3": [C.POINTER(ConvParams), vp, ci, vp],
    "pf_gemm_split3_timed": [C.POINTER(ConvParams), ci, C.POINTER(cf), vp],
    "pf_gemm_bf16_pp": [C.POINTER(ConvParams), vp],
    "pf_split3": [vp, ci, vp, ci, cl, cl, ci, vp],
    "pf_layernorm_split3": [vp, ci, vp, ci, cl, ci, vp, vp, cf, cl, ci, vp],
    "pf_vit_attention_qkv_split3": [vp, vp, cl, ci, ci, ci, ci, vp],
    "pf_vit_attention_split3": [vp, cl, vp, cl, ci, ci, ci, ci, vp],
    "pf_vit_attention_split3_v2": [vp, cl, vp, cl, ci, ci, ci, ci, ci, ci, vp],
    "pf_conv_winograd_fused": [C.POINTER(ConvParams), vp, ci, ci, vp],
    "pf_conv_winograd_fused_timed": [C.POINTER(ConvParams), vp, ci, ci, ci, C.POINTER(cf), vp],
    "pf_attractor": [vp, ci, ci, ci, cf, ci, ci, vp, ci, ci, vp, ci, ci, ci, ci, vp],
    "pf_seed_bin_centers": [vp, ci, vp, cl, ci, cf, cf, ci, ci, vp],
    "pf_bounded_bin_centers": [vp, vp, cl, ci, cf, cf, vp],
    "pf_logbinom_depth": [vp, ci, vp, ci, ci, vp, ci, ci, ci, ci, cf, cf, vp],
    "pf_bins_tail": [vp, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, ci, ci, ci, ci, cf, cf, vp],
    "pf_stitch_init": [vp, vp, ci, ci, vp, vp, vp, ci, ci, ci, vp],
    "pf_stitch_finish_init": [vp, vp, vp, cl, vp],
    "pf_stitch_update": [vp, vp, ci, ci, vp, ci, ci, vp, ci, ci, ci, ci, vp],
    "pf_resize_nearest_f32": [vp, ci, ci, vp, ci, ci, vp],
    "pf_resize_bilinear_f32": [vp, ci, ci, vp, ci, ci, vp],
    # input / output side (io.hip)
    "pf_u8_bicubic_to_f32": [vp, ci, ci, ci, vp, ci, ci, vp],
    "pf_percentiles_f32": [vp, cl, cf, ci, vp, C.c_double, C.c_double, vp, vp, vp],
    "pf_colorize_f32": [vp, cl, vp, vp, ci, cf, ci, vp, C.c_uint32, vp, vp],
    "pf_depth_to_u16": [vp, cl, cf, vp, vp],
    "pf_silog_loss": [vp, vp, cl, cf, cf, cf, vp, vp, vp],
    "pf_depth_metrics": [vp, ci, ci, vp, ci, ci, vp, vp, cf, cf, ci, ci, ci, ci, vp, vp],
}
NON_STATUS = ("pf_last_error", "pf_version", "pf_percentile_workspace_bytes", "pf_conv_winograd_fused_supported", "pf_gemm_split3_route")   # entry points that do not return a status

_lib = None


def load():
    """Load the shared library and bind every symbol of the C ABI; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(or `make -C patchfusion_amd/csrc`). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = ci
    lib.pf_last_error.restype = C.c_char_p
    lib.pf_last_error.argtypes = []
    lib.pf_version.restype = ci
    lib.pf_version.argtypes = []
    lib.pf_percentile_workspace_bytes.restype = ci
    lib.pf_percentile_workspace_bytes.argtypes = []
    lib.pf_conv_winograd_fused_supported.restype = ci          # 1 / 0, not a status
    lib.pf_conv_winograd_fused_supported.argtypes = [C.POINTER(ConvParams)]
    lib.pf_gemm_split3_route.restype = ci                      # PF_S3_ROUTE_* (or -1), not a status
    lib.pf_gemm_split3_route.argtypes = [C.POINTER(ConvParams), ci]
    _lib = lib
    return lib


class PfError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().pf_last_error()
        raise PfError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
