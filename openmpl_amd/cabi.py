"""ctypes binding of libmpl_hip.so (the C ABI declared in include/mpl_hip.h).

This is the only place Python touches the native library.  There is no CPU or eager
fallback: if the library cannot be loaded the import of the binding raises, and every
non-zero return code becomes a ``RuntimeError`` (SURVEY.md section 8b "Error convention").
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import build as _build

MPL_MAX_VIEWS = 32
MPL_MAX_APPS = 64
ABI_VERSION = 13

# flag bits (mpl_hip.h MPL_F_*)
F_MULTI_SPT = 1 << 0
F_CONF_ADD = 1 << 1
F_CONF_MULT = 1 << 2
F_CONF_ATTN_W = 1 << 3
F_POS3D_LEARN = 1 << 4
F_POS3D_SPATIAL = 1 << 5
F_RAYS_TOKEN = 1 << 6
F_POS3D_TO_RAYS = 1 << 7
F_NO_SPT = 1 << 8
F_NO_FPT = 1 << 9
F_CONF_IN_FPT = 1 << 10
F_KPTOK = 1 << 11
F_NO_SMALL_STACK = 1 << 12
# mpl_block_stack_form(): the kernel form a block stack of a given shape takes (MPL_FORM_* of mpl_hip.h) and the kernel behind it
FORM_UNPACKED, FORM_SMALL, FORM_TEAMS, FORM_PAIRS, FORM_ROWS32, FORM_ROWS16, FORM_ROWS16_DIRECT, FORM_PER_GEMM = range(8)
FORM_KERNELS = {FORM_UNPACKED: "ln_gemm_ng_kernel", FORM_SMALL: "sm_stack_kernel", FORM_TEAMS: "h2_stack_kernel<%d>", FORM_PAIRS: "h2_stack2_kernel<2>",
                FORM_ROWS32: "h2_stackn_kernel<2>", FORM_ROWS16: "h2_stackn_kernel<2>", FORM_ROWS16_DIRECT: "h2_stackd_kernel<2>",
                FORM_PER_GEMM: "h2_gemm_kernel"}

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL = 0, 1, 2

_fp = C.c_void_p  # device float* carried as an integer address


class BlockWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                   "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                   "qkv_w16", "proj_w16", "fc1_w16", "fc2_w16",
                                   "qkv_w3", "proj_w3", "fc1_w3", "fc2_w3",
                                   "qkv_h2", "proj_h2", "fc1_h2", "fc2_h2")]


class SptSet(C.Structure):
    _fields_ = [("embed_w", _fp), ("embed_b", _fp), ("conf_w", _fp), ("conf_b", _fp),
                ("pos_embed", _fp), ("blocks", _fp)]


class Config(C.Structure):
    _fields_ = [("num_joints", C.c_int32), ("dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32),
                ("num_views", C.c_int32), ("in_chans", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_int32)]


class Weights(C.Structure):
    _fields_ = [("spt_sets", _fp),
                ("spatial_norm_w", _fp), ("spatial_norm_b", _fp),
                ("pos_3d_embed", _fp), ("pos_3d_view_coding", _fp),
                ("pos_3d_linear_w", _fp), ("pos_3d_linear_b", _fp),
                ("ray_embed_w", _fp), ("ray_embed_b", _fp),
                ("conf_fpt_w", _fp), ("conf_fpt_b", _fp),
                ("fpt_blocks", C.POINTER(BlockWeights)),
                ("view_norm_w", _fp), ("view_norm_b", _fp),
                ("wmean_w", _fp), ("wmean_b", _fp),
                ("head_ln_w", _fp), ("head_ln_b", _fp),
                ("head_w", _fp), ("head_b", _fp), ("spt_packed", C.c_uint32), ("reserved", C.c_uint32)]


class Inputs(C.Structure):
    _fields_ = [("batch", C.c_int32), ("reserved", C.c_int32),
                ("poses", _fp * MPL_MAX_VIEWS), ("rays", _fp * MPL_MAX_VIEWS), ("centers", _fp * MPL_MAX_VIEWS)]


EXPORTS = ("mpl_hip_abi_version", "mpl_hip_error_string", "mpl_fpt_width", "mpl_forward_workspace_bytes",
           "mpl_forward", "mpl_spt_tokens", "mpl_block_stack_workspace_bytes", "mpl_block_stack", "mpl_block_stack_ex",
           "mpl_ln_linear", "mpl_spt_pack_bytes", "mpl_spt_pack", "mpl_d32_pack", "mpl_pack_bf16_bytes", "mpl_pack_bf16", "mpl_pack_h2_bytes", "mpl_pack_h2", "mpl_pack_h2_scaled", "mpl_pack_h2_out_scale", "mpl_ln_linear_h2_workspace_bytes", "mpl_ln_linear_h2", "mpl_x3_debug_buffer", "mpl_x3_stack_mode", "mpl_block_stack_form", "mpl_block_stack_form_ex", "mpl_block_stack_last_form", "mpl_device_error", "mpl_device_error_clear", "mpl_x3_spin_limit", "mpl_token_attention", "mpl_fuse_head", "mpl_view_fuse", "mpl_view_norm",
           "mpl_layernorm", "mpl_linear", "mpl_pose_metrics_size", "mpl_pose_metrics", "mpl_pose_metrics_ex", "mpl_prepare_inputs", "mpl_profile_start",
           "mpl_profile_stop")
KINDS = ("spt", "row_stats", "gemm", "attention", "fuse_head", "pack")

_lib = None
_lock = threading.Lock()


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Load (building first if the .so is absent and hipcc exists).  Raises on any failure."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        # stale or missing library: rebuild (under build.py's file lock, published atomically) when a compiler is
        # there; a box without hipcc must have received a library that is at least as new as the sources
        if _build.needs_build():
            try:
                _build.build()
            except _build.CompilerMissing:
                # ONLY the missing-compiler case is tolerated (a box without hipcc that was handed a prebuilt library);
                # a compile or link failure of edited sources propagates -- never run stale kernels silently
                if not os.path.exists(path):
                    raise
                import warnings
                warnings.warn("libmpl_hip.so is older than openmpl_amd/csrc and hipcc is not available to rebuild it")
        # torch bundles its own HIP runtime (libamdhip64): it must be in the process BEFORE this library is mapped, so
        # that both resolve to ONE runtime instance -- loaded the other way round, the kernels of this library are
        # launched on a second runtime that has no initialised device ("no ROCm-capable device is detected")
        import torch  # noqa: F401
        lib = C.CDLL(path)
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError("libmpl_hip.so does not export %s" % name)
        lib.mpl_hip_abi_version.restype = C.c_int
        if lib.mpl_hip_abi_version() != ABI_VERSION:
            raise RuntimeError("libmpl_hip.so ABI %d != binding ABI %d (rebuild: python -m openmpl_amd.build --force)"
                               % (lib.mpl_hip_abi_version(), ABI_VERSION))
        lib.mpl_hip_error_string.restype = C.c_char_p
        lib.mpl_hip_error_string.argtypes = [C.c_int]
        lib.mpl_fpt_width.restype = C.c_int
        lib.mpl_fpt_width.argtypes = [C.POINTER(Config)]
        lib.mpl_forward_workspace_bytes.restype = C.c_size_t
        lib.mpl_forward_workspace_bytes.argtypes = [C.POINTER(Config), C.c_int]
        lib.mpl_forward.restype = C.c_int
        lib.mpl_forward.argtypes = [C.POINTER(Config), C.POINTER(Weights), C.POINTER(Inputs), _fp, _fp, C.c_size_t, _fp]
        lib.mpl_spt_tokens.restype = C.c_int
        lib.mpl_spt_tokens.argtypes = [C.POINTER(Config), C.POINTER(Weights), C.POINTER(Inputs), _fp, _fp]
        lib.mpl_block_stack_workspace_bytes.restype = C.c_size_t
        lib.mpl_block_stack_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.mpl_block_stack.restype = C.c_int
        lib.mpl_block_stack.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BlockWeights),
                                        C.POINTER(C.c_uint8), C.c_int, _fp, C.c_size_t, _fp]
        lib.mpl_block_stack_ex.restype = C.c_int
        lib.mpl_block_stack_ex.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BlockWeights),
                                           C.POINTER(C.c_uint8), C.c_int, _fp, C.c_size_t, C.c_uint, _fp]
        lib.mpl_ln_linear.restype = C.c_int
        lib.mpl_ln_linear.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, C.c_float, _fp, _fp, C.c_int, C.c_int, _fp,
                                      _fp, _fp, _fp]
        lib.mpl_spt_pack_bytes.restype = C.c_size_t
        lib.mpl_spt_pack_bytes.argtypes = []
        lib.mpl_spt_pack.restype = C.c_int
        lib.mpl_spt_pack.argtypes = [C.POINTER(BlockWeights), _fp, _fp]
        lib.mpl_d32_pack.restype = C.c_int
        lib.mpl_d32_pack.argtypes = [C.POINTER(BlockWeights), _fp, _fp]
        lib.mpl_pack_bf16_bytes.restype = C.c_size_t
        lib.mpl_pack_bf16_bytes.argtypes = [C.c_int, C.c_int]
        lib.mpl_pack_bf16.restype = C.c_int
        lib.mpl_pack_bf16.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp]
        lib.mpl_pack_h2_bytes.restype = C.c_size_t
        lib.mpl_pack_h2_bytes.argtypes = [C.c_int, C.c_int]
        lib.mpl_pack_h2.restype = C.c_int
        lib.mpl_pack_h2.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp]
        lib.mpl_pack_h2_scaled.restype = C.c_int
        lib.mpl_pack_h2_scaled.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp]
        lib.mpl_pack_h2_out_scale.restype = C.c_void_p
        lib.mpl_pack_h2_out_scale.argtypes = [_fp, C.c_int, C.c_int]
        lib.mpl_ln_linear_h2_workspace_bytes.restype = C.c_size_t
        lib.mpl_ln_linear_h2_workspace_bytes.argtypes = [C.c_int, C.c_int]
        lib.mpl_ln_linear_h2.restype = C.c_int
        lib.mpl_ln_linear_h2.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, C.c_int, C.c_int, _fp, _fp, _fp,
                                         _fp, C.c_size_t, _fp]
        lib.mpl_x3_stack_mode.restype = C.c_int
        lib.mpl_x3_stack_mode.argtypes = [C.c_int]
        lib.mpl_block_stack_form.restype = C.c_int
        lib.mpl_block_stack_form.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
        lib.mpl_block_stack_form_ex.restype = C.c_int
        lib.mpl_block_stack_form_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
        lib.mpl_block_stack_last_form.restype = C.c_int
        lib.mpl_block_stack_last_form.argtypes = []
        for fn in (lib.mpl_device_error, lib.mpl_device_error_clear, lib.mpl_x3_spin_limit):
            fn.restype = C.c_int
            fn.argtypes = [C.c_int]
        lib.mpl_x3_debug_buffer.restype = C.c_int
        lib.mpl_x3_debug_buffer.argtypes = [_fp]
        lib.mpl_token_attention.restype = C.c_int
        lib.mpl_token_attention.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]
        lib.mpl_fuse_head.restype = C.c_int
        lib.mpl_fuse_head.argtypes = [C.POINTER(Config), C.POINTER(Weights), _fp, C.c_int, _fp, _fp]
        lib.mpl_view_fuse.restype = C.c_int
        lib.mpl_view_fuse.argtypes = [C.POINTER(Config), C.POINTER(Weights), _fp, C.c_int, _fp, _fp]
        lib.mpl_view_norm.restype = C.c_int
        lib.mpl_view_norm.argtypes = [C.POINTER(Config), C.POINTER(Weights), _fp, C.c_int, _fp, _fp]
        lib.mpl_layernorm.restype = C.c_int
        lib.mpl_layernorm.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, C.c_float, _fp, _fp]
        lib.mpl_linear.restype = C.c_int
        lib.mpl_linear.argtypes = [_fp, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp,
                                   C.c_float, C.c_int, _fp, _fp]
        lib.mpl_pose_metrics_size.restype = C.c_int
        lib.mpl_pose_metrics_size.argtypes = [C.c_int]
        lib.mpl_pose_metrics.restype = C.c_int
        lib.mpl_pose_metrics.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), _fp, _fp]
        lib.mpl_pose_metrics_ex.restype = C.c_int
        lib.mpl_pose_metrics_ex.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, _fp, _fp]
        lib.mpl_prepare_inputs.restype = C.c_int
        lib.mpl_prepare_inputs.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                           C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), _fp]
        lib.mpl_profile_start.restype = C.c_int
        lib.mpl_profile_start.argtypes = []
        lib.mpl_profile_stop.restype = C.c_int
        lib.mpl_profile_stop.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]
        _lib = lib
    return _lib


def profile_start():
    check(load().mpl_profile_start(), "mpl_profile_start")


def profile_stop():
    """-> {kind: (total_ms, launches)} for the launches since profile_start()."""
    ms = (C.c_float * len(KINDS))()
    n = (C.c_int * len(KINDS))()
    check(load().mpl_profile_stop(ms, n, len(KINDS)), "mpl_profile_stop")
    return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(KINDS)}


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mpl_hip_error_string(rc).decode()
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg, rc))


def device_error(device: int = -1) -> bool:
    """True when a kernel of an earlier call on `device` (default: the current one) reported a lost hand-off; every
    call on that device fails with RuntimeError until clear_device_error().  Reads pinned memory, no synchronisation."""
    return bool(load().mpl_device_error(device))


def clear_device_error(device: int = -1):
    check(load().mpl_device_error_clear(device), "mpl_device_error_clear")


def raise_if_device_error(device: int = -1, synchronize: bool = False):
    """Surface a lost hand-off WITH the results it poisoned.  The forward is asynchronous: the call that produced NaN poses
    has long returned when its kernel reports the failure, and without this check the error would only show on the NEXT
    call into the library -- never, if the failing batch was the last one.  Call it wherever outputs are consumed on the host
    (GatherHandle.wait() and pose_metrics() do; a validate() loop should after its final batch, with synchronize=True, which
    waits for the device first so that a failure of work still in flight is seen too)."""
    if synchronize:
        import torch
        torch.cuda.synchronize(None if device < 0 else device)
    if device_error(device):
        check(-5, "a forward on this device")
