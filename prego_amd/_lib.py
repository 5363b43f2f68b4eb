"""ctypes binding of libprego_amd.so (include/prego_amd.h).

There is no CPU fallback: if the library is missing or cannot be loaded the
import of the product path fails loudly.  (`python -m prego_amd.build` or
`__graft_entry__.build()` produces the library in-tree.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEBUG_LIB_PATH = os.path.join(_HERE, "lib", "libprego_amd_debug.so")      # product ABI + the probe / unit-test entry points (prego_amd_debug.h)
# PREGO_AMD_LIB: an A/B build; PREGO_AMD_DEBUG_LIB=1: the measurement scripts under scripts/ run their engines on the debug library
LIB_PATH = os.environ.get("PREGO_AMD_LIB") or (DEBUG_LIB_PATH if os.environ.get("PREGO_AMD_DEBUG_LIB") else os.path.join(_HERE, "lib", "libprego_amd.so"))

PREGO_F32, PREGO_BF16, PREGO_F16, PREGO_F16X2 = 0, 1, 2, 3
FWD_SOFTMAX, FWD_KEEP, FWD_IN16 = 1, 2, 4
E_TIMEOUT = -4

# every symbol include/prego_amd.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "prego_abi_version", "prego_last_error",
    "prego_miniroad_create", "prego_miniroad_create_layers", "prego_miniroad_set_gru_layer", "prego_miniroad_destroy", "prego_miniroad_last_error", "prego_miniroad_set_weights",
    "prego_miniroad_max_clips", "prego_miniroad_workspace_bytes", "prego_miniroad_forward",
    "prego_miniroad_check", "prego_miniroad_timing_enable", "prego_miniroad_timing_read",
    "prego_miniroad_set_dropout", "prego_oad_loss", "prego_oad_loss_reduce",
    "prego_miniroad_backward_workspace_bytes", "prego_miniroad_backward", "prego_adamw_step", "prego_miniroad_adamw_step", "prego_window_vote", "prego_format_ids", "prego_perframe_ap_labels", "prego_onehot_labels",
    "prego_vit_create", "prego_vit_destroy", "prego_vit_num_tensors", "prego_vit_set_weights",
    "prego_vit_workspace_bytes", "prego_vit_forward",
    "prego_vit_set_dropout", "prego_vit_train_workspace_bytes", "prego_vit_forward_train", "prego_vit_backward",
    "prego_attention_layer_workspace_bytes", "prego_attention_layer_forward",
    "prego_attention_layer_create", "prego_attention_layer_destroy", "prego_attention_layer_set_weights",
    "prego_attention_layer_handle_workspace_bytes", "prego_attention_layer_handle_forward",
    "prego_attention_layer_set_dropout", "prego_attention_layer_train_workspace_bytes", "prego_attention_layer_forward_train", "prego_attention_layer_backward", "prego_vit_adamw_step", "prego_miniroad_step",
    "prego_perframe_ap_workspace_bytes", "prego_perframe_ap", "prego_vit_frames_workspace_bytes", "prego_vit_forward_frames", "prego_miniroad_backward_events", "prego_miniroad_backward_callback",
    "prego_miniroad_plan_starts", "prego_miniroad_set_feed_events", "prego_miniroad_pass_info",
    "prego_vit_set_compute_dtype", "prego_attention_layer_set_compute_dtype",
    "prego_miniroad_resident_bytes", "prego_miniroad_set_resident", "prego_miniroad_guard_publish", "prego_miniroad_set_peer_guard",
    "prego_miniroad_set_gru_layer_grads",
]
# include/prego_amd_debug.h: only in libprego_amd_debug.so
DEBUG_SYMBOLS = ["prego_miniroad_debug_stamps", "prego_debug_gemm_bf16", "prego_debug_attention_bwd", "prego_debug_attention_fwd",
                 "prego_debug_recurrence_only", "prego_debug_gemm_worker", "prego_debug_head_only",
                 "prego_debug_split_fault", "prego_debug_split_state", "prego_debug_set_abort", "prego_debug_alloc_count", "prego_debug_hog"]


class PregoError(RuntimeError):
    pass


_lib = None


def _open(path: str, debug: bool) -> C.CDLL:
    if not os.path.exists(path):
        raise PregoError(
            f"{path} not found: the HIP library has not been built. Run `python -m prego_amd.build` "
            "(needs hipcc; cross-compiles gfx950 without a GPU). There is no CPU fallback.")
    # torch ships its own libamdhip64; import it first so that the one HIP runtime in the process is
    # torch's (loading ours first makes torch's copy fail with "no ROCm-capable device")
    import torch  # noqa: F401
    lib = C.CDLL(path)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    lib.prego_abi_version.restype = i32
    lib.prego_last_error.restype = C.c_char_p
    lib.prego_miniroad_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32]
    lib.prego_miniroad_create_layers.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32, i32]
    lib.prego_miniroad_set_gru_layer.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.prego_miniroad_destroy.argtypes = [vp]
    lib.prego_miniroad_destroy.restype = None
    lib.prego_miniroad_last_error.argtypes = [vp]
    lib.prego_miniroad_last_error.restype = C.c_char_p
    lib.prego_miniroad_set_weights.argtypes = [vp] + [vp] * 10 + [vp]
    lib.prego_miniroad_max_clips.argtypes = [vp]
    lib.prego_miniroad_workspace_bytes.argtypes = [vp, i32, C.POINTER(C.c_int32), i64, i32]
    lib.prego_miniroad_workspace_bytes.restype = sz
    lib.prego_miniroad_resident_bytes.argtypes = [vp, i32, C.POINTER(C.c_int32), i32]
    lib.prego_miniroad_resident_bytes.restype = sz
    lib.prego_miniroad_set_resident.argtypes = [vp, vp, sz]
    lib.prego_miniroad_guard_publish.argtypes = [vp, vp, vp]
    lib.prego_miniroad_set_peer_guard.argtypes = [vp, vp]
    lib.prego_miniroad_set_gru_layer_grads.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.prego_miniroad_forward.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp),
                                           C.POINTER(vp), C.POINTER(vp), vp, vp, i32, vp, sz, vp]
    lib.prego_miniroad_step.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, vp]
    lib.prego_miniroad_check.argtypes = [vp, vp]
    lib.prego_miniroad_timing_enable.argtypes = [vp, i32]
    lib.prego_miniroad_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double),
                                               C.POINTER(i64), C.POINTER(C.c_double)]
    lib.prego_miniroad_set_dropout.argtypes = [vp, C.c_float, C.c_uint64]
    lib.prego_oad_loss.argtypes = [i32, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp), i32, vp, C.POINTER(vp),
                                   C.c_float, vp]
    lib.prego_oad_loss_reduce.argtypes = [i32, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(vp), i32, i32, vp, C.POINTER(vp),
                                          C.c_float, vp]
    lib.prego_miniroad_backward_workspace_bytes.argtypes = [vp, i32, C.POINTER(C.c_int32)]
    lib.prego_miniroad_backward_workspace_bytes.restype = sz
    lib.prego_miniroad_backward.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(vp)] + [vp] * 10 + [vp, sz, vp, sz, vp]
    lib.prego_miniroad_backward_events.argtypes = [vp, vp, vp]
    lib.prego_miniroad_backward_callback.argtypes = [vp, vp, vp]
    lib.prego_miniroad_plan_starts.argtypes = [vp, i32, C.POINTER(C.c_int32), i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.prego_miniroad_set_feed_events.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(vp), i32]
    lib.prego_miniroad_pass_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.prego_window_vote.argtypes = [vp, i64, i32, i32, vp, vp]
    lib.prego_format_ids.argtypes = [vp, i64, vp, vp, vp]
    lib.prego_perframe_ap_workspace_bytes.argtypes = [i64, i32]
    lib.prego_perframe_ap_workspace_bytes.restype = sz
    lib.prego_perframe_ap.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, sz, vp]
    lib.prego_perframe_ap_labels.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, sz, vp]
    lib.prego_onehot_labels.argtypes = [i32, C.POINTER(vp), C.POINTER(i64), i32, vp, C.POINTER(C.c_int32)]
    f32 = C.c_float
    lib.prego_adamw_step.argtypes = [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), i64, f32, f32, f32, f32, f32, vp]
    lib.prego_miniroad_adamw_step.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64, f32, f32, f32, f32, f32, vp]
    lib.prego_vit_adamw_step.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, f32, f32, f32, f32, f32, vp]
    lib.prego_vit_create.argtypes = [C.POINTER(vp)] + [i32] * 8
    lib.prego_vit_destroy.argtypes = [vp]
    lib.prego_vit_destroy.restype = None
    lib.prego_vit_num_tensors.argtypes = [vp]
    lib.prego_vit_set_weights.argtypes = [vp, C.POINTER(vp), i32, vp]
    lib.prego_vit_workspace_bytes.argtypes = [vp, i32]
    lib.prego_vit_workspace_bytes.restype = sz
    lib.prego_vit_forward.argtypes = [vp, i32, vp, vp, vp, i32, vp, sz, vp]
    lib.prego_vit_frames_workspace_bytes.argtypes = [vp, i32, i32]
    lib.prego_vit_frames_workspace_bytes.restype = sz
    lib.prego_vit_forward_frames.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, vp, sz, vp]
    lib.prego_vit_set_compute_dtype.argtypes = [vp, i32]
    lib.prego_attention_layer_set_compute_dtype.argtypes = [vp, i32]
    lib.prego_vit_set_dropout.argtypes = [vp, C.c_float, C.c_float, C.c_uint64]
    lib.prego_vit_train_workspace_bytes.argtypes = [vp, i32]
    lib.prego_vit_train_workspace_bytes.restype = sz
    lib.prego_vit_forward_train.argtypes = [vp, i32, vp, vp, vp, i32, vp, sz, vp]
    lib.prego_vit_backward.argtypes = [vp, i32, vp, C.POINTER(vp), i32, i32, vp, sz, vp]
    lib.prego_attention_layer_workspace_bytes.argtypes = [i32, i32, i32]
    lib.prego_attention_layer_workspace_bytes.restype = sz
    lib.prego_attention_layer_forward.argtypes = [i32] * 5 + [vp] * 10 + [vp, sz, vp]
    lib.prego_attention_layer_create.argtypes = [C.POINTER(vp), i32, i32]
    lib.prego_attention_layer_destroy.argtypes = [vp]
    lib.prego_attention_layer_destroy.restype = None
    lib.prego_attention_layer_set_weights.argtypes = [vp] + [vp] * 8 + [vp]
    lib.prego_attention_layer_handle_workspace_bytes.argtypes = [vp, i32, i32]
    lib.prego_attention_layer_handle_workspace_bytes.restype = sz
    lib.prego_attention_layer_handle_forward.argtypes = [vp, i32, i32, i32, vp, vp, vp, sz, vp]
    lib.prego_attention_layer_set_dropout.argtypes = [vp, C.c_float, C.c_uint64]
    lib.prego_attention_layer_train_workspace_bytes.argtypes = [vp, i32, i32]
    lib.prego_attention_layer_train_workspace_bytes.restype = sz
    lib.prego_attention_layer_forward_train.argtypes = [vp, i32, i32, i32, vp, vp, vp, sz, vp]
    lib.prego_attention_layer_backward.argtypes = [vp, i32, i32, i32, vp, vp, vp, i32, vp, sz, vp]
    for name in SYMBOLS:          # fail loudly at load time if the library is stale
        getattr(lib, name)
    if debug:
        lib.prego_miniroad_debug_stamps.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.prego_debug_head_only.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
        lib.prego_debug_attention_bwd.argtypes = [i32] * 5 + [vp] * 7 + [vp]
        lib.prego_debug_attention_fwd.argtypes = [i32] * 6 + [vp] * 5 + [vp]
        lib.prego_debug_recurrence_only.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        lib.prego_debug_gemm_worker.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]
        lib.prego_debug_gemm_bf16.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, vp]
        lib.prego_debug_split_fault.argtypes = [vp, i32]
        lib.prego_debug_split_state.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]
        lib.prego_debug_set_abort.argtypes = [vp, C.c_uint32, vp]
        lib.prego_debug_alloc_count.argtypes = [C.POINTER(i64), C.POINTER(i64)]
        lib.prego_debug_hog.argtypes = [i32, i32, i32, vp, vp, sz, vp, vp]
    return lib


def load() -> C.CDLL:
    """the product library (include/prego_amd.h): what every module under prego_amd/ runs on"""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH, debug=LIB_PATH == DEBUG_LIB_PATH)
    return _lib


_dbg = None


def load_debug() -> C.CDLL:
    """libprego_amd_debug.so: the product ABI plus the probe / unit-test entry points of include/prego_amd_debug.h.  Used by the
    kernel-level GPU tests and the scripts under scripts/; never by the product path."""
    global _dbg
    if _dbg is None:
        _dbg = load() if LIB_PATH == DEBUG_LIB_PATH else _open(DEBUG_LIB_PATH, debug=True)
    return _dbg




def check(rc: int):
    if rc != 0:
        raise PregoError(f"prego_amd error {rc}: {load().prego_last_error().decode()}")


def ptr_array(ptrs):
    """host array of device pointers (None -> NULL)"""
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
