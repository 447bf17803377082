"""ctypes binding of libfar3d_hip.so (the C-ABI declared in include/far3d_hip.h).

There is deliberately NO fallback: if the shared library is missing, or a kernel is asked to run
without a HIP device, this raises.  PyTorch is only used for device memory and streams.
"""
import ctypes
import os

# The engine's frame pipeline drives several HIP streams at once (3 camera streams + a head stream per engine); HIP multiplexes
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialise.  Takes effect only if the
# HIP runtime has not been initialised yet (import far3d_amd before the first torch.cuda call, or export it yourself).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# torch bundles its own HIP runtime (file libamdhip64.so, SONAME libamdhip64.so.7).  It MUST be mapped before our
# library so that both share ONE runtime (and therefore streams / device pointers); loading ours first would pull in
# /opt/rocm's copy and torch would then initialise a second runtime and report "no GPU".
import torch  # noqa: F401  (import order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfar3d_hip.so")

c_int, c_float, c_void_p, c_char_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_char_p
c_long = ctypes.c_long
_p = c_void_p

# name -> (restype, argtypes).  Must list every symbol of include/far3d_hip.h (tests check it).
SIGNATURES = {
    "far3d_last_error": (c_char_p, []),
    "far3d_abi_version": (c_int, []),
    "far3d_device_count": (c_int, []),
    "far3d_device_arch": (c_int, [c_int, c_char_p, c_int]),
    "far3d_msda_forward": (c_int, [_p, c_int, _p, _p, _p, _p, _p] + [c_int] * 7 + [_p]),
    "far3d_aggregate_forward": (c_int, [_p, c_int, _p, _p, _p, _p, _p, _p, _p, _p, c_int] + [c_int] * 7 +
                                [_p, _p, _p, c_float, c_float, c_int, c_int, c_int, _p, _p, c_int, _p, _p]),
    "far3d_conv2d_nhwc": (c_int, [_p, c_int, _p, c_int, _p, _p, c_int] + [c_int] * 5 + [c_long] + [c_int] * 4 +
                          [c_long] + [c_int] * 5 + [_p, c_int, c_int, c_long, c_int, c_int] +
                          [_p, c_int, c_int, c_long, _p, _p, _p, c_int, _p]),
    "far3d_attention_forward": (c_int, [_p, _p, _p, c_int, _p, c_int] + [c_int] * 8 + [c_float, _p, c_int, c_int, _p]),
    "far3d_attention_f32_variant": (c_int, [c_int]),
    "far3d_layernorm": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, c_float, c_int, _p, c_int, _p, c_int, c_int, _p, c_int, c_int, _p]),
    "far3d_layernorm_rows": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, c_float, c_int, _p, c_int, _p, c_int, c_int, _p, c_int, c_int, _p, _p]),
    "far3d_rowchain_attn_out": (c_int, [_p, c_int, _p, c_int, _p, c_int, _p, _p, _p, _p, _p, _p, c_int, _p, c_int, _p, c_int,
                                        _p, c_int, c_float, _p]),
    "far3d_rowchain_ffn": (c_int, [_p, c_int, _p, c_int, _p, c_int] + [_p] * 12 + [_p, c_int, _p, c_int, _p, c_int, c_int, c_float, _p]),
    "far3d_rowchain_qkv": (c_int, [_p, c_int, _p, c_int, _p, _p, _p, c_int, c_int, _p]),
    "far3d_rowchain_branches": (c_int, [_p, c_int] + [_p] * 10 + [c_int] + [_p] * 6 + [c_int, _p, c_int, _p, c_int, c_int, c_float, _p]),
    "far3d_ese_nhwc": (c_int, [_p, c_int, _p, _p, _p, _p, _p, c_int, c_int, c_int, c_int, c_long, c_int, c_long,
                               c_int, c_long, _p, _p]),
    "far3d_ese_fused_nhwc": (c_int, [_p, c_int, _p, _p, _p, _p, _p, _p, c_int, c_int, c_int, c_int, c_int, c_long, c_int, c_long, c_int, c_long,
                                     c_int, c_int, c_int, c_long, _p, _p]),
    "far3d_cam_embed_chain": (c_int, [_p] * 10 + [c_int, c_int, c_int, c_int, c_float, c_int, _p]),
    "far3d_groupnorm_nhwc": (c_int, [_p, c_int, _p, _p, _p, _p, c_int, c_int, c_int, c_int, c_float, c_int, _p]),
    "far3d_maxpool3x3s2_nhwc": (c_int, [_p, c_int, _p] + [c_int] * 7 + [c_long, _p]),
    "far3d_stem_im2col": (c_int, [_p, _p, c_int, c_int, c_int, c_int, _p]),
    "far3d_stem_conv": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, c_long, c_int, _p]),
    "far3d_proposal_select": (c_int, [_p, _p, c_int, c_int, c_int, c_int, _p, _p, _p, _p, _p, _p, c_int, c_float, c_int, _p]),
    "far3d_proposal_gather": (c_int, [_p, c_int, c_int, c_int, _p, _p, _p, _p, c_int, _p, _p, c_int, c_int, c_int, c_int,
                                      c_float, c_float, c_int, _p, _p, c_int, c_int, _p, c_float, _p, _p, _p, _p, c_int, _p, _p, _p]),
    "far3d_compact_rows": (c_int, [_p, _p, c_int, c_int, c_int, _p, c_int, _p, _p, _p]),
    "far3d_row_affine_ln": (c_int, [_p, _p, _p, _p, _p] + [c_int] * 6 + [c_float, c_int, _p]),
    "far3d_posemb3d": (c_int, [_p, _p, _p, c_int, _p]),
    "far3d_memory_prepare": (c_int, [_p] * 9 + [c_float, c_int, _p, c_int, c_int, c_int] + [_p] * 9),
    "far3d_head_finalize": (c_int, [_p] * 5 + [c_int] * 4 + [_p, _p, c_int, c_int, _p]),
    "far3d_memory_post_update": (c_int, [_p] * 10 + [c_int] * 4 + [_p] * 6),
    "far3d_add_cast": (c_int, [_p, _p, _p, c_int, _p, c_int, c_int, c_int, c_long, c_long, _p]),
    "far3d_agg_order": (c_int, [_p, _p, _p, c_int, c_int, _p, c_float, c_float, _p, c_int, c_int, c_int, _p, _p, c_int, c_int, c_int, _p, _p, _p]),
    "far3d_agg_tables": (c_int, [_p, _p, c_int, c_int, c_int, _p]),
    "far3d_topk": (c_int, [_p, c_int, c_int, _p, _p, _p]),
    "far3d_decode_topk": (c_int, [_p, _p, c_int, c_int, c_int, c_int, _p, _p, _p, _p, _p, _p, c_long, _p]),
    "far3d_decode_topk_mem": (c_int, [_p, _p, c_int, c_int, c_int, c_int, _p, _p, _p, _p, _p, _p, c_long, _p, c_int, c_int, _p, _p]),
    "far3d_camera_prep": (c_int, [_p, _p, _p, _p, _p, c_int, _p]),
    "far3d_nan_to_num": (c_int, [_p, _p, c_long, _p]),
    "far3d_image_resample_h": (c_int, [_p, c_long, c_int, c_int, _p, _p, _p, c_int, c_int, c_int, c_int, c_int, _p]),
    "far3d_image_resample_v": (c_int, [_p, c_int, c_int, c_int, _p, _p, c_int, c_int, c_int, c_int, c_int, _p, c_int, c_int, c_int,
                                       _p, _p, c_int, _p]),
}

_lib = None


class Far3dHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Far3dHipError(
            "libfar3d_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python far3d_amd/build.py`). There is no CPU fallback for the Far3D hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale -> loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().far3d_last_error()
        raise Far3dHipError("%s failed (code %d): %s" % (what, status, msg.decode() if msg else "?"))


def require_device():
    lib = load()
    if lib.far3d_device_count() <= 0:
        raise Far3dHipError("far3d_amd needs a HIP device (MI355X / gfx950); none is visible and there is "
                            "no CPU fallback. The CPU restatement lives in oracle/ and is test-only.")
    return lib
