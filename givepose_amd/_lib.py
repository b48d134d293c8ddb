"""ctypes binding of libgivepose_hip.so (the C ABI declared in include/givepose_hip.h).

There is NO fallback: if the HIP library is missing or does not load, importing the product ops
raises.  Nothing here imports oracle/.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# GP_LIB_PATH: A/B runs of two builds on one box (scripts/race_probe.py); the default is the in-tree library
LIB_PATH = os.environ.get("GP_LIB_PATH") or os.path.join(_HERE, "libgivepose_hip.so")

ABI_VERSION = 323        # include/givepose_hip.h GP_ABI_VERSION: gp_gemm_desc layout (checked against gp_version() at load)
GP_F32, GP_F16, GP_F64 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU, ACT_LRELU = 0, 1, 2, 3
EPI_NONE, EPI_GELU, EPI_RELU, EPI_LRELU, EPI_SCALE_RES, EPI_RES_RELU, EPI_LNFOLD_GELU = 0, 1, 2, 3, 4, 5, 6
KC_GEMM, KC_DCNV3, KC_DWCONV_LN, KC_NORM, KC_ELEMENTWISE, KC_SMALL, KC_COUNT = 0, 1, 2, 3, 4, 5, 6
KC_NAMES = ["gemm", "dcnv3", "dwconv_ln", "norm", "elementwise", "small"]


class GemmDesc(Structure):
    _fields_ = [("X", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("gamma", c_void_p), ("residual", c_void_p),
                ("C", c_void_p), ("workspace", c_void_p),
                ("M", c_int), ("N", c_int), ("K", c_int), ("ldx", c_int), ("ldc", c_int), ("ldres", c_int),
                ("epilogue", c_int), ("out_f32", c_int), ("splitk", c_int),
                ("B", c_int), ("H", c_int), ("Win", c_int), ("Cin", c_int), ("KH", c_int), ("KW", c_int),
                ("stride", c_int), ("pad", c_int), ("Ho", c_int), ("Wo", c_int), ("dtype", c_int),
                ("gn_partial", c_void_p), ("gn_groups", c_int), ("gn_hw", c_int), ("variant", c_int),
                ("ln_stats", c_void_p), ("ln_colsum", c_void_p), ("ln_nslab", c_int), ("ln_eps", c_float),
                ("co_scheduled", c_int), ("prefetch", c_void_p), ("prefetch_bytes", c_long),
                ("split_shift", c_int), ("x_plane_stride", c_long), ("w_plane_stride", c_long),
                ("out_planes", c_int), ("c_plane_stride", c_long),
                ("residual_f32", c_int), ("c16", c_void_p), ("ldc16", c_int), ("gn_rows", c_int)]


# name -> argtypes; every symbol include/givepose_hip.h declares (tests/test_abi.py checks both ways)
_P = c_void_p
PROTOTYPES = {
    "gp_last_error": ([], c_char_p),
    "gp_version": ([], c_int),
    "gp_device_info": ([POINTER(c_int), c_char_p, c_int], c_int),
    "gp_dcnv3_forward": ([_P, _P, _P, _P] + [c_int] * 9 + [c_float] + [c_int] * 7 + [_P], c_int),
    "gp_dcnv3_forward_any": ([_P] * 4 + [c_int] * 13 + [c_float] + [c_int] * 3 + [_P], c_int),
    "gp_dcnv3_backward": ([_P] * 7 + [c_long, c_long] + [c_int] * 13 + [c_float] + [c_int] * 3 + [_P], c_int),
    "gp_gemm": ([POINTER(GemmDesc), _P], c_int),
    "gp_gemm_gn_rows": ([c_int] * 4, c_int),
    "gp_split_planes": ([_P, _P, c_long, c_int, c_long, c_long, c_int, _P], c_int),
    "gp_convnext_mlp_pack_w2": ([_P, _P, c_int, _P], c_int),
    "gp_convnext_mlp_pack_w2_s32": ([_P, _P, c_int, _P], c_int),
    "gp_convnext_mlp": ([_P] * 8 + [c_long, c_int, c_int, _P], c_int),
    "gp_convnext_stem": ([_P] * 6 + [c_int] * 4 + [c_float, c_int, _P], c_int),
    "gp_dwconv_ln": ([_P] * 6 + [c_int] * 5 + [c_float, c_int, c_long, c_int, _P], c_int),
    "gp_dwconv_ln_groups": ([_P] * 6 + [c_int] * 5 + [c_float, c_int, _P, c_int, _P], c_int),
    "gp_dwconv7_raw_stats": ([_P] * 5 + [c_int] * 5 + [_P], c_int),
    "gp_layernorm": ([_P] * 4 + [c_long, c_int, c_float, c_int, c_int, _P], c_int),
    "gp_groupnorm_chunks": ([c_int, c_int], c_int),
    "gp_groupnorm_stats": ([_P] * 2 + [c_int] * 5 + [_P], c_int),
    "gp_groupnorm_apply": ([_P] * 5 + [c_int] * 4 + [c_float] + [c_int] * 4 + [_P], c_int),
    "gp_groupnorm_upsample2x": ([_P] * 5 + [c_int] * 5 + [c_float] + [c_int] * 3 + [_P], c_int),
    "gp_groupnorm_apply_xyz": ([_P] * 8 + [c_int] * 4 + [c_float] + [c_int] * 3 + [_P], c_int),
    "gp_upsample_bilinear2x": ([_P, _P] + [c_int] * 5 + [_P], c_int),
    "gp_deconv_col2im": ([_P, _P] + [c_int] * 5 + [_P], c_int),
    "gp_xyz_out_layer": ([_P] * 5 + [c_int] * 4 + [_P], c_int),
    "gp_pointwise_k3": ([_P] * 4 + [c_long, c_int, c_int, _P], c_int),
    "gp_pnp_conv1": ([_P] * 4 + [c_int] * 4 + [_P], c_int),
    "gp_xyz_conv3x3_s2": ([_P] * 3 + [c_int] * 4 + [_P], c_int),
    "gp_size_head": ([_P] * 8 + [c_int] * 5 + [_P], c_int),
    "gp_pose_tail": ([_P, _P, c_int] + [_P] * 10 + [c_int, c_int] + [_P] * 5 + [c_int, _P], c_int),
    "gp_patchify_xyz": ([_P, _P, c_int, c_int, c_int, c_int, _P], c_int),
    "gp_attention64": ([_P, _P, c_int, c_int, c_int, _P], c_int),
    "gp_resnet_stem": ([_P] * 4 + [c_int] * 4 + [_P], c_int),
    "gp_maxpool3x3s2": ([_P, _P] + [c_int] * 5 + [_P], c_int),
    "gp_mask_resize_nearest": ([_P, _P, c_int, c_int, c_int, _P], c_int),
    "gp_crop_rois": ([_P] * 12 + [c_int] * 7 + [_P], c_int),
    "gp_pred_rt": ([_P] * 6 + [c_int, _P], c_int),
    "gp_pack_poses": ([_P] * 4 + [c_int, _P], c_int),
    "gp_sn_stem": ([_P] * 4 + [c_int] * 3 + [_P], c_int),
    "gp_sn_pointwise": ([_P] * 6 + [c_long] + [c_int] * 4 + [_P], c_int),
    "gp_sn_depthwise": ([_P] * 4 + [c_int] * 7 + [_P], c_int),
    "gp_sn_avgpool": ([_P] * 2 + [c_int] * 3 + [_P], c_int),
    "gp_sn_se": ([_P] * 6 + [c_int] * 3 + [_P], c_int),
    "gp_sn_head": ([_P] * 12 + [c_int] * 5 + [_P], c_int),
    "gp_graph_begin": ([_P], c_int),
    "gp_graph_end": ([_P, POINTER(c_void_p)], c_int),
    "gp_graph_launch": ([_P, _P], c_int),
    "gp_graph_destroy": ([_P], c_int),
    "gp_timing_begin": ([_P], c_int),
    "gp_timing_end": ([], c_int),
    "gp_timing_report": ([c_int, POINTER(c_long), POINTER(c_double), POINTER(c_double), POINTER(c_double)], c_int),
    "gp_timing_top": ([c_int, c_char_p, c_int, POINTER(c_int), POINTER(c_long), POINTER(c_double), POINTER(c_double), POINTER(c_double)], c_int),
}

_lib = None


class GivePoseHipError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GivePoseHipError(
            f"{LIB_PATH} not found: build it with `python -m givepose_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU or PyTorch fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.gp_version() != ABI_VERSION:      # a stale build would read gp_gemm_desc with another layout
        raise GivePoseHipError(f"{LIB_PATH}: ABI version {lib.gp_version()}, this package needs {ABI_VERSION}: rebuild with "
                               "`python -m givepose_amd.build`")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().gp_last_error()
        raise GivePoseHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


_capturing = 0


def capturing():
    """True between gp_graph_begin and gp_graph_end of this process (raw hipStreamBeginCapture: torch does not see it)."""
    return _capturing > 0
