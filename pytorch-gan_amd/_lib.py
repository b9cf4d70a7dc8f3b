"""ctypes binding of libmigan.so (include/migan.h).

There is NO fallback: if the HIP library cannot be built or loaded, importing this module raises.
torch must be imported first so that libmigan.so binds to the SAME libamdhip64.so.7 instance that torch
uses (streams and device pointers are only meaningful inside one HIP runtime); this is verified below.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_ulonglong, c_void_p

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_CSRC, "libmigan.so")


def _ensure_built():
    # always ask build.py: it compares a digest of the sources and flags with the stamp of the existing binary (cheap),
    # rebuilds under a file lock when they differ, and keeps a shipped binary where there is no compiler
    import importlib.util

    spec = importlib.util.spec_from_file_location("_migan_build", os.path.join(_CSRC, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(force=bool(os.environ.get("MIGAN_REBUILD")), verbose=False)


def _hip_runtime_copies():
    seen = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    seen.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return seen


_ensure_built()
if not os.path.exists(LIB_PATH):
    raise ImportError("libmigan.so is missing and could not be built (%s)" % LIB_PATH)
lib = ctypes.CDLL(LIB_PATH)
_copies = _hip_runtime_copies()
if len(_copies) > 1:
    raise ImportError(
        "two HIP runtimes are mapped in this process (%s): libmigan.so must share torch's libamdhip64; "
        "import torch before pytorch_gan_amd" % sorted(_copies)
    )

P = c_void_p
_SIGS = {
    "migan_version": (c_char_p, []),
    "migan_error_string": (c_char_p, [c_int]),
    "migan_debug_launch_count": (ctypes.c_long, [c_char_p]),
    "migan_debug_launch_reset": (None, []),
    "migan_conv2d_fwd": (c_int, [P, P, P, P] + [c_int] * 14 + [c_float, P]),
    "migan_conv2d_dropout_fwd": (c_int, [P, P, P, P, P] + [c_int] * 14 + [c_float, P]),
    "migan_conv2d_stats_chunks": (c_int, [c_int] * 14),
    "migan_conv2d_fwd_stats": (c_int, [P, P, P, P, P] + [c_int] * 14 + [c_float, P, c_int, c_int, P]),
    "migan_upconv3x3_stats_chunks": (c_int, [c_int] * 6),
    "migan_upconv3x3_fwd_stats": (c_int, [P, P, P, P] + [c_int] * 6 + [c_float, P, c_int, c_int, P]),
    "migan_norm_stats_from_conv": (c_int, [P, c_int, P, P, P, P, P, c_float, c_float, c_int, c_int, P]),
    "migan_skinny_nt_ok": (c_int, [c_int] * 3),
    "migan_skinny_nn_ok": (c_int, [c_int] * 3),
    "migan_skinny_nt": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "migan_skinny_nn": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "migan_rgb_conv_ok": (c_int, [c_int] * 6 + [ctypes.c_longlong]),
    "migan_rgb_conv_fwd": (c_int, [P, P, P, P] + [c_int] * 13 + [c_float, c_int, P]),
    "migan_rgb_conv_wgrad_ok": (c_int, [c_int] * 6 + [ctypes.c_longlong]),
    "migan_rgb_conv_wgrad_workspace": (c_size_t, [c_int] * 3),
    "migan_rgb_conv_wgrad": (c_int, [P, P, P, P, P, P, c_size_t] + [c_int] * 12 + [c_float, c_int, c_int, P]),
    "migan_thin_toeplitz_ok": (c_int, [c_int] * 6),
    "migan_thin_toeplitz_cols": (c_int, [c_int] * 2),
    "migan_thin_toeplitz_workspace": (c_size_t, [c_int] * 5),
    "migan_thin_toeplitz_pack": (c_int, [P, P, P] + [c_int] * 4 + [P]),
    "migan_thin_toeplitz_fwd": (c_int, [P, P, P, P, P, c_size_t] + [c_int] * 13 + [c_float, P]),
    "migan_thin_toeplitz_expand": (c_int, [P, P] + [c_int] * 8 + [P]),
    "migan_thin_toeplitz_wgrad_workspace": (c_size_t, [c_int] * 7),
    "migan_thin_toeplitz_wgrad": (c_int, [P, P, P, P, c_size_t] + [c_int] * 11 + [P]),
    "migan_thin_toeplitz_dgrad_workspace": (c_size_t, [c_int] * 7),
    "migan_thin_toeplitz_dgrad": (c_int, [P, P, P, P, c_size_t] + [c_int] * 10 + [P]),
    "migan_resample_u8": (c_int, [P, P, P, P] + [c_int] * 7 + [P]),
    "migan_u8_to_f32": (c_int, [P] * 6 + [c_int] * 7 + [P]),
    "migan_norm_apply_prelu": (c_int, [P] * 8 + [c_int] * 5 + [P]),
    "migan_norm_workspace_prelu": (c_size_t, [c_int] * 3),
    "migan_norm_bwd_prelu": (c_int, [P] * 11 + [c_int] * 3 + [P, c_size_t, c_int, c_int, P, c_int, c_int, P]),
    "migan_multi_permute4d": (c_int, [P, P, c_int, P]),
    "migan_batch_mean_axpy": (c_int, [P, P, P, c_int, c_size_t, c_float, c_float, P]),
    "migan_skinny_tn_ok": (c_int, [c_int] * 3),
    "migan_skinny_tn": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "migan_c64_conv_ok": (c_int, [c_int] * 13),
    "migan_c64_pack_floats": (c_size_t, []),
    "migan_c64_pack": (c_int, [P, P, c_int, P]),
    "migan_c64_pack_multi": (c_int, [P, c_int, P]),
    "migan_c64_wgrad_workspace": (c_size_t, [c_int] * 3),
    "migan_conv2d_fwd_normed_ok": (c_int, [c_int] * 12),
    "migan_conv2d_fwd_normed": (c_int, [P] * 4 + [c_int] * 13 + [c_float, P, P, P, P, c_int, c_float, P]),
    "migan_bn_conv1_bwd_ok": (c_int, [c_int] * 4),
    "migan_bn_conv1_bwd_workspace": (c_size_t, [c_int] * 4),
    "migan_bn_conv1_bwd": (c_int, [P] * 7 + [c_int, c_float, P, P, c_int, P, c_int, P, P, c_int, P, P, c_size_t] + [c_int] * 4 + [P]),
    "migan_c64_conv_wgrad": (c_int, [P, P, P, P, c_size_t] + [c_int] * 4 + [P, c_int, P, c_int, P, P, P, P, c_int, c_float, P, P]),
    "migan_c64_conv_fwd": (c_int, [P, P, P, P] + [c_int] * 4 + [c_float, c_int, P, P, P, P, c_int, c_float, P, P]),
    "migan_fewpix_ok": (c_int, [c_int] * 3),
    "migan_fewpix_nt_workspace": (c_size_t, [c_int] * 3),
    "migan_fewpix_nt": (c_int, [P, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_float, P]),
    "migan_im2col_small": (c_int, [P, P] + [c_int] * 11 + [P]),
    "migan_col2im_small": (c_int, [P, P, P] + [c_int] * 12 + [c_float, P]),
    "migan_act_bwd_nc": (c_int, [P, P, P, P] + [c_int] * 4 + [c_float, P]),
    "migan_conv2d_dgrad": (c_int, [P, P, P, P] + [c_int] * 13 + [c_float, P]),
    "migan_conv_splitk_workspace": (c_size_t, []),
    "migan_conv_splitk_applies": (c_int, [ctypes.c_longlong, c_int, c_int, c_int]),
    "migan_conv2d_fwd_ws": (c_int, [P, P, P, P, P] + [c_int] * 14 + [c_float, P, c_size_t, P]),
    "migan_conv2d_dgrad_ws": (c_int, [P, P, P, P] + [c_int] * 13 + [c_float, P, c_size_t, P]),
    "migan_conv2d_dgrad_relu_ws": (c_int, [P, P, P, P] + [c_int] * 12 + [P, c_size_t, P]),
    "migan_igemm_tile_code": (c_int, [ctypes.c_longlong, c_int, c_int, c_int]),
    "migan_conv2d_wgrad_workspace": (c_size_t, [c_int] * 7),
    "migan_conv2d_wgrad": (c_int, [P, P, P, P, c_size_t] + [c_int] * 14 + [P, c_int, P, c_int, P]),
    "migan_conv2d_dgrad_reflect1": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "migan_conv2d_dgrad_reflect1_ring": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "migan_conv2d_dgrad_reflect1_ws": (c_int, [P, P, P] + [c_int] * 5 + [P, c_size_t, P]),
    "migan_conv2d_dgrad_reflect1_ring_ws": (c_int, [P, P, P] + [c_int] * 5 + [P, c_size_t, P]),
    "migan_conv2d_wgrad_fuses_bias": (c_int, [c_int] * 6),
    "migan_upconv3x3_pack": (c_int, [P, P, P, c_int, c_int, P]),
    "migan_upconv3x3_fwd": (c_int, [P, P, P, P] + [c_int] * 6 + [c_float, P]),
    "migan_upconv3x3_dgrad": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "migan_upconv3x3_wgrad_workspace": (c_size_t, [c_int] * 5),
    "migan_upconv3x3_wgrad": (c_int, [P, P, P, P, c_size_t] + [c_int] * 6 + [P, c_int, P, c_int, P]),
    "migan_norm_workspace": (c_size_t, [c_int] * 3),
    "migan_norm_stats": (c_int, [P, P, P, P, P, P, c_float, c_float, c_int, c_int, c_int, P, c_size_t, P]),
    "migan_norm_apply": (c_int, [P] * 7 + [c_int] * 4 + [c_float, P]),
    "migan_norm_colsum_slabs": (c_int, [c_int] * 3),
    "migan_norm_bwd": (c_int, [P] * 9 + [c_int] * 4 + [c_float, P, c_size_t, c_int, P, P]),
    "migan_norm_bwd_sums": (c_int, [P] * 9 + [c_int] * 4 + [c_float, P, c_size_t, c_int, P]),
    "migan_norm_bwd_apply": (c_int, [P] * 8 + [c_int] * 4 + [c_float, ctypes.c_longlong, P, P]),
    "migan_norm_bwd_sums_prelu": (c_int, [P] * 11 + [c_int] * 3 + [P, c_size_t, c_int, c_int, c_int, c_int, P]),
    "migan_norm_bwd_apply_prelu": (c_int, [P] * 9 + [c_int] * 3 + [ctypes.c_longlong, P, c_int, c_int, P]),
    "migan_norm_moments": (c_int, [P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    "migan_norm_sync_finalize": (c_int, [P, c_int, ctypes.c_longlong, P, P, P, P, P, c_float, c_float, c_int, P]),
    "migan_rsqrt_eps": (c_int, [P, P, c_int, c_float, P]),
    "migan_act_bwd_colsum": (c_int, [P] * 5 + [c_int] * 4 + [c_float, P]),
    "migan_act_fwd": (c_int, [P, P, c_size_t, c_int, c_float, P]),
    "migan_act_bwd": (c_int, [P, P, P, c_size_t, c_int, c_float, P]),
    "migan_act_bwd2": (c_int, [P, P, P, P, c_size_t, c_int, P]),
    "migan_norm_workspace2": (c_size_t, [c_int] * 3),
    "migan_norm_bwd2": (c_int, [P] * 9 + [c_int] * 4 + [P, c_size_t, P]),
    "migan_dragan_interp": (c_int, [P, P, P, P, P, c_size_t, P]),
    "migan_reduce_workspace": (c_size_t, []),
    "migan_prelu_fwd": (c_int, [P, P, P, c_size_t, P]),
    "migan_prelu_bwd": (c_int, [P, P, P, P, P, P, c_size_t, P]),
    "migan_axpby": (c_int, [P, c_float, P, c_float, P, c_size_t, P]),
    "migan_mul": (c_int, [P, P, P, c_size_t, P]),
    "migan_mul_nc": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "migan_rand_mask": (c_int, [P, c_size_t, c_float, c_ulonglong, P, P]),
    "migan_gather2d_fwd": (c_int, [P, P] + [c_int] * 9 + [P]),
    "migan_gather2d_bwd": (c_int, [P, P] + [c_int] * 9 + [P]),
    "migan_pixel_shuffle": (c_int, [P, P] + [c_int] * 6 + [P]),
    "migan_maxpool2_fwd": (c_int, [P, P] + [c_int] * 4 + [P]),
    "migan_maxpool2_bwd": (c_int, [P, P, P] + [c_int] * 4 + [P]),
    "migan_maxpool2_relu_bwd": (c_int, [P, P, P] + [c_int] * 4 + [P]),
    "migan_cat_channels": (c_int, [P, P, P, c_size_t, c_int, c_int, c_int, P]),
    "migan_select_rows": (c_int, [P, P, P, P, P, c_int, c_size_t, P]),
    "migan_transpose_batched": (c_int, [P, P, c_int, c_int, c_int, P]),
    "migan_permute4d": (c_int, [P, P] + [c_int] * 8 + [P]),
    "migan_colsum_workspace": (c_size_t, [c_size_t, c_int]),
    "migan_colsum": (c_int, [P, P, c_size_t, c_int, P, c_size_t, c_int, P]),
    "migan_loss_fwd": (c_int, [c_int, P, P, c_float, P, c_size_t, P, c_size_t, P]),
    "migan_loss_bwd": (c_int, [c_int, P, P, c_float, P, P, c_size_t, P]),
    "migan_embedding_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "migan_embedding_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "migan_softmax_fwd": (c_int, [P, P, c_int, c_int, P]),
    "migan_softmax_bwd": (c_int, [P, P, P, c_int, c_int, P]),
    "migan_cross_entropy_fwd": (c_int, [P, P, P, P, c_int, c_int, P]),
    "migan_cross_entropy_bwd": (c_int, [P, P, P, P, P, c_int, c_int, P]),
    "migan_rownorm_fwd": (c_int, [P, P, c_int, c_int, P]),
    "migan_rownorm_bwd": (c_int, [P, P, P, P, c_int, c_int, P]),
    "migan_rowscale": (c_int, [P, P, P, c_int, c_int, P]),
    "migan_pullaway_fwd": (c_int, [P, P, P, c_int, c_int, P]),
    "migan_pullaway_bwd": (c_int, [P, P, P, P, c_int, c_int, P]),
    "migan_critic_fused_ok": (c_int, [c_int] * 4),
    "migan_critic_fused_workspace": (c_size_t, [c_int] * 4),
    "migan_critic_fused": (c_int, [P] * 17 + [c_size_t] + [c_int] * 4 + [c_float, c_float, c_int, c_int, P]),
    "migan_mlp_fused_ok": (c_int, [c_int, c_int, P]),
    "migan_mlp_fused_workspace": (c_size_t, [c_int, c_int, P, c_int]),
    "migan_mlp_fused_bwd_workspace": (c_size_t, [c_int, c_int, P]),
    "migan_mlp_fused_fwd": (c_int, [P, P, c_int, c_int, P, P, P, P, c_size_t, c_int, P, c_int, P]),
    "migan_mlp_fused_bwd": (c_int, [P, P, P, P, P, c_int, c_int, P, P, P, P, P, c_size_t, c_int, c_int, P]),
    "migan_norm_small_ok": (c_int, [c_int, c_int, c_int]),
    "migan_norm_fwd_small": (c_int, [P] * 8 + [c_int] * 4 + [c_float, c_float, P]),
    "migan_norm_bwd_small": (c_int, [P] * 8 + [c_int] * 4 + [c_float, P, P]),
    "migan_zero": (c_int, [P, c_size_t, P]),
    "migan_adam_chunk": (c_int, []),
    "migan_adam_step": (c_int, [P, P, c_int, P, P, P, c_float, c_float, c_float, c_float, c_float, P]),
}
EXPORTS = sorted(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here == header/library drift: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def check(code, what=""):
    if code != 0:
        msg = lib.migan_error_string(int(code))
        raise RuntimeError("libmigan %s failed: hipError %d (%s)" % (what, code, msg.decode() if msg else "?"))


def version():
    return lib.migan_version().decode()
