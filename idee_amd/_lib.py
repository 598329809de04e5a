"""ctypes binding of libidee_hip.so (the C-ABI declared in include/idee_hip.h).

The product path has no fallback: if the library is missing, cannot be loaded, or a call
returns non-zero, this module raises.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDEE_HIP_LIB: another build of the same library (A/B timing of kernel variants on one GPU box); default: the in-tree build
LIB_PATH = os.environ.get("IDEE_HIP_LIB") or os.path.join(_HERE, "libidee_hip.so")

c_float_p = C.c_void_p  # raw device pointers travel as void*
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float


class SwinGeom(C.Structure):
    _fields_ = [(n, _I) for n in ("N", "V", "D", "H", "W", "C", "heads", "hidden", "wd", "wh", "ww", "sd", "sh", "sw")] + [
        ("scale", _F), ("act_dtype", _I), ("drop", _F), ("attn_drop", _F), ("drop_seed", C.c_uint32)]


_SWIN_PTRS = ("qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "rel_bias")


class SwinParams(C.Structure):
    _fields_ = [(n, _P) for n in _SWIN_PTRS + ("path_scale",)]


class SwinGrads(C.Structure):
    _fields_ = [(n, _P) for n in _SWIN_PTRS]


class ConvDesc(C.Structure):
    _fields_ = ([("mode", _I), ("B", _I), ("wsets", _I), ("T", _I), ("H", _I), ("W", _I), ("Cin", _I), ("Cout", _I)]
                + [(n, _L) for n in ("x_sB", "x_sT", "x_sH", "x_sW", "x_sG", "y_sB", "y_sT", "y_sH", "y_sW", "y_sG")]
                + [("relu", _I), ("x_gpu", _I), ("y_gpu", _I), ("x_sU", _L), ("y_sU", _L), ("act_dtype", _I)])


DTYPE_F32, DTYPE_BF16 = 0, 1


def act_dtype(t):
    """IDEE_DTYPE_* of an activation tensor (f32 or bf16 storage)."""
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    if t.dtype == torch.float32:
        return DTYPE_F32
    raise IdeeHipError(f"activation tensors must be float32 or bfloat16, got {t.dtype}")


CONV_K333_REPLICATE, CONV_K233_S2_ZERO, CONV_K133_ZERO = 0, 1, 2

# name -> (restype, argtypes); every symbol include/idee_hip.h declares
_SIGS = {
    "idee_version": (C.c_char_p, []),
    "idee_last_error": (C.c_char_p, []),
    "idee_check_device": (_I, [_I]),
    "idee_set_option": (_I, [C.c_char_p, C.c_char_p]),
    "idee_get_option": (_I, [C.c_char_p, C.c_char_p, _I]),
    "idee_patch_embed_ln_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P]),
    "idee_patch_embed_ln_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P]),
    "idee_stage_embed_ln_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _L, _I, _P]),
    "idee_stage_embed_ln_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _P]),
    "idee_swin_block_fwd": (_I, [_P, _P, _P, C.POINTER(SwinParams), C.POINTER(SwinGeom), _P]),
    "idee_swin_block_bwd": (_I, [_P, _P, _P, _P, _P, C.POINTER(SwinParams), C.POINTER(SwinGrads), C.POINTER(SwinGeom), _P]),
    "idee_relpos_bias_gather": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "idee_relpos_bias_scatter": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "idee_conv_packed_floats": (_L, [C.POINTER(ConvDesc)]),
    "idee_conv_wgrad_floats": (_L, [C.POINTER(ConvDesc)]),
    "idee_conv_pack_weights": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "idee_conv_fwd": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "idee_conv_dgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _P]),
    "idee_conv_dgrad_add": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "idee_conv_wgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "idee_conv_unpack_wgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P]),
    "idee_ln_relu_res_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _I, _P, _P]),
    "idee_ln_relu_res_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P, _P]),
    "idee_dropout_mask_": (_I, [_P, _L, _F, C.c_uint64, _I, _P]),
    "idee_lfq_workspace_floats": (_L, [_L, _I, _I]),
    "idee_lfq_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _F, _F, _F, _I, _P]),
    "idee_lfq_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _F, _I, _P]),
    "idee_lfq_noproj_workspace_floats": (_L, [_L, _I, _I]),
    "idee_lfq_noproj_fwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _F, _I, _P]),
    "idee_lfq_noproj_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _F, _F, _I, _P]),
    "idee_patch_merging_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "idee_patch_merging_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "idee_lfq_sign_field_fwd": (_I, [_P, _P, _I, _I, _I, _L, _I, _I, _I, _P]),
    "idee_lfq_sign_field_bwd": (_I, [_P, _P, _I, _I, _I, _L, _I, _I, _I, _P]),
    "idee_lfq_fold_weights_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "idee_lfq_fold_weights_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "idee_vq_workspace_floats": (_L, [_L]),
    "idee_vq_nn_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _P]),
    "idee_vq_nn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P]),
    "idee_vq_cluster_stats": (_I, [_P, _P, _P, _P, _L, _I, _I, _P]),
    "idee_vq_ema_update": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _P]),
    "idee_vq_kmeans_update": (_I, [_P, _P, _P, _I, _I, _P]),
    "idee_vq_expire_codes": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P]),
    "idee_bce_class_counts": (_I, [_P, _P, _L, _P, _P]),
    "idee_bce_workspace_floats": (_L, [_I, _L]),
    "idee_bce_weighted_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _P]),
    "idee_bce_weighted_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _P]),
    "idee_anomaly_workspace_floats": (_L, [_L]),
    "idee_anomaly_l1_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _P]),
    "idee_anomaly_l1_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _P]),
    "idee_anomaly_signs_workspace_floats": (_L, [_L, _I, _I]),
    "idee_anomaly_l1_signs_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P]),
    "idee_anomaly_l1_signs_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _P]),
    "idee_lfq_code_vectors": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "idee_loss_combine_fwd": (_I, [_P, _I, _I, _P, _F, _P, _P, _P]),
    "idee_loss_combine_bwd": (_I, [_P, _I, _I, _F, _P, _P, _P, _P]),
    "idee_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    "idee_confusion_counts": (_I, [_P, _P, _P, _L, _F, _P, _P]),
    "idee_anomaly_vote_accumulate": (_I, [_P, _P, _P, _P, _I, _I, _I, _L, _I, _P]),
    "idee_anomaly_vote_finalize": (_I, [_P, _P, _P, _L, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


class IdeeHipError(RuntimeError):
    pass


def load():
    """Load libidee_hip.so (once) and attach signatures.  Raises if it is missing: there is no
    CPU or eager-PyTorch fallback for the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IdeeHipError(
            f"{LIB_PATH} not found: build it with `python idee_amd/csrc/build.py` "
            "(hipcc --offload-arch=gfx950). The HIP library is required; there is no fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().idee_last_error().decode()


def check(rc: int, what: str):
    if rc != 0:
        raise IdeeHipError(f"{what} failed (code {rc}): {last_error()}")


def set_option(name: str, value=None):
    """Set a kernel-selection option of the library (include/idee_hip.h: idee_set_option); None / "" = default."""
    check(load().idee_set_option(name.encode(), None if value is None else str(value).encode()), "idee_set_option")


def get_option(name: str) -> str:
    buf = C.create_string_buffer(64)
    check(load().idee_get_option(name.encode(), buf, 64), "idee_get_option")
    return buf.value.decode()


class options:
    """Context manager: set options for the duration of a block and restore the previous values.
    `with _lib.options(IDEE_SWIN_MATH="f22"): ...`"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.saved = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            set_option(k, v)
        return False


def ptr(t):
    """Device pointer of a contiguous f32/bf16/int64 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise IdeeHipError("idee_amd kernels need tensors on a HIP device (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise IdeeHipError("idee_amd kernels need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-entry-point timing with HIP events on the launch stream (bench.py): name -> [(start, end), ...]
_event_log = None


def set_event_log(names):
    """names: iterable of C-ABI entry points to bracket with timing events, or None to stop."""
    global _event_log
    _event_log = {n: [] for n in names} if names else None
    return _event_log


def call(name, *args):
    log = _event_log
    if log is not None and name in log:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(getattr(load(), name)(*args), name)
        e1.record()
        log[name].append((e0, e1, args))
        return
    check(getattr(load(), name)(*args), name)
