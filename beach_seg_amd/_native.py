"""ctypes binding of the C ABI in `include/beach_seg_amd.h` (the HIP library `libbsg_hip.so`).

There is NO fallback: if the library is missing or no MI355X is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "libbsg_hip.so"
_lib = None


def use_library(path) -> None:
    """A/B tooling only (tools/step_ab.sh, tools/attn_probe.py): bind another build of the library instead of the in-tree
    one.  Must be called before the first `load()`; the product path never calls it and reads no environment variable."""
    global _LIB_PATH
    if _lib is not None:
        raise NativeError("use_library() after the library has been loaded")
    _LIB_PATH = Path(path).resolve()

BSG_DTYPE_F32, BSG_DTYPE_BF16, BSG_DTYPE_F16 = 0, 1, 2
BSG_MAX_TAPS, BSG_GLOBAL_WEIGHTS, BSG_LAYER_WEIGHTS = 8, 16, 20

# every exported symbol of include/beach_seg_amd.h (tests check the library exports exactly these)
SYMBOLS = (
    "bsg_create", "bsg_destroy", "bsg_workspace_bytes", "bsg_workspace_region", "bsg_forward", "bsg_backward",
    "bsg_loss_scratch_bytes", "bsg_loss_fwd_bwd", "bsg_decode_argmin", "bsg_prompt_gather",
    "bsg_prompt_grad_scatter", "bsg_adamw_step", "bsg_vote_paste", "bsg_vote_argmax", "bsg_last_error",
    "bsg_build_info", "bsg_profile_enable", "bsg_profile_read", "bsg_profile_reset", "bsg_op_gemm", "bsg_op_gemm_epilogue", "bsg_forward_ensemble", "bsg_backward_rows", "bsg_forward_rows", "bsg_op_attention_windows", "bsg_tile_frontend", "bsg_decode_hf",
    "bsg_op_attention", "bsg_op_attention_scratch_bytes", "bsg_tif_image", "bsg_train_aug", "bsg_train_aug_bwd",
    "bsg_confusion_update", "bsg_loss_fwd_bwd_ids", "bsg_mask_rgb_norm",
)


class BsgConfig(C.Structure):
    _fields_ = [
        ("hidden_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
        ("canvas_h", C.c_int), ("canvas_w", C.c_int), ("patch_size", C.c_int), ("mlp_dim", C.c_int),
        ("decoder_hidden", C.c_int), ("merge_index", C.c_int), ("num_taps", C.c_int),
        ("taps", C.c_int * BSG_MAX_TAPS), ("layer_norm_eps", C.c_float), ("dtype", C.c_int),
        ("embed_split", C.c_int), ("gemm_x3", C.c_int),
    ]


class NativeError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load (once) and return the HIP library; raises NativeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise NativeError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: it carries its own HIP runtime (torch/lib/libamdhip64.so).  If this library were loaded before torch,
    # the process would hold two runtimes and bsg_create would see no device ("no HIP device") although torch does.
    import torch  # noqa: F401
    lib = C.CDLL(os.fspath(_LIB_PATH))
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.bsg_last_error.restype = C.c_char_p
    lib.bsg_build_info.restype = C.c_char_p
    lib.bsg_create.argtypes = [C.POINTER(BsgConfig), C.POINTER(vp), i, C.POINTER(vp)]
    lib.bsg_destroy.argtypes = [vp]
    lib.bsg_destroy.restype = None
    lib.bsg_workspace_bytes.argtypes = [vp, i, i]
    lib.bsg_workspace_bytes.restype = sz
    lib.bsg_workspace_region.argtypes = [vp, i, i, C.c_char_p, i, C.POINTER(sz), C.POINTER(sz)]
    lib.bsg_forward.argtypes = [vp, vp, i, vp, vp, vp, i, vp, vp, sz, i]
    lib.bsg_forward_rows.argtypes = [vp, vp, i, vp, vp, vp, i, i, vp, vp, sz, i]
    lib.bsg_backward.argtypes = [vp, vp, i, vp, vp, vp, sz]
    lib.bsg_backward_rows.argtypes = [vp, vp, i, vp, i, vp, vp, sz]
    lib.bsg_forward_ensemble.argtypes = [vp, vp, i, vp, vp, vp, i, vp, vp, sz]
    lib.bsg_loss_scratch_bytes.argtypes = [i, i]
    lib.bsg_loss_scratch_bytes.restype = sz
    lib.bsg_loss_fwd_bwd.argtypes = [vp, i, i, i, vp, vp, vp, f, i, vp, vp, vp, sz]
    lib.bsg_loss_fwd_bwd_ids.argtypes = [vp, i, i, i, i, vp, vp, vp, f, i, vp, vp, vp, sz]
    lib.bsg_mask_rgb_norm.argtypes = [vp, i, i, i, i, vp, vp, C.c_float * 3, C.c_float * 3, vp]
    lib.bsg_decode_argmin.argtypes = [vp, i, i, i, i, vp, vp, vp, vp]
    f3 = C.c_float * 3
    lib.bsg_prompt_gather.argtypes = [vp, i, i, i, vp, vp, f3, f3, vp]
    lib.bsg_prompt_grad_scatter.argtypes = [vp, i, i, i, vp, vp, f3, vp]
    lib.bsg_adamw_step.argtypes = [vp, i, C.c_long, vp, vp, vp, vp, vp, vp, vp, vp, f, f, f, f, f, f]
    lib.bsg_vote_paste.argtypes = [vp, i, vp, i, i, i, vp, vp, i, i, i]
    lib.bsg_vote_argmax.argtypes = [vp, vp, C.c_long, i, vp]
    lib.bsg_decode_hf.argtypes = [vp, i, i, i, i, vp, vp, f3, f3, vp]
    lib.bsg_tile_frontend.argtypes = [vp, vp, i, i, i, vp, i, i, vp, vp, i, f3, f3, vp, vp]
    lib.bsg_profile_enable.argtypes = [vp, i]
    lib.bsg_profile_read.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    lib.bsg_profile_reset.argtypes = [vp]
    lib.bsg_op_gemm.argtypes = [vp, i, i, i, i, vp, vp, vp, vp]
    lib.bsg_op_gemm_epilogue.argtypes = [vp, i, i, i, i, i, vp, vp, vp, vp, vp, vp]
    lib.bsg_tif_image.argtypes = [vp, i, i, i, i, vp, vp, vp, vp]
    lib.bsg_train_aug.argtypes = [vp, i, i, i, vp, vp, vp, vp, vp, f3, f3, vp, vp, vp]
    lib.bsg_train_aug_bwd.argtypes = [vp, i, i, i, vp, vp, vp, vp, f3, vp, vp]
    lib.bsg_confusion_update.argtypes = [vp, C.c_long, i, i, vp, vp, vp, vp]
    lib.bsg_op_attention_scratch_bytes.argtypes = [i, i, i]
    lib.bsg_op_attention_scratch_bytes.restype = sz
    lib.bsg_op_attention.argtypes = [vp, i, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, sz]
    lib.bsg_op_attention_windows.argtypes = [i, i, i, i]
    _lib = lib
    return lib


def check(status: int, exc=NativeError):
    if status != 0:
        msg = load().bsg_last_error().decode()
        # HF raises ValueError for a bad embedding_type (HF:199); keep that contract
        raise (ValueError if msg.startswith("Embedding type") else exc)(msg)
