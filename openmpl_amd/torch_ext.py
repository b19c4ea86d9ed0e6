"""Loader of the torch extension (csrc/torch_ext.cpp -> lib/mpl_torch_ext.so): TORCH_LIBRARY operators over the C ABI.

    openmpl_amd::bind / unbind / lift      (see the header comment of csrc/torch_ext.cpp)

``ops()`` builds the extension when it is missing or older than its sources (g++, ~10 s), loads it into the dispatcher, hands it
the entry points of libmpl_hip.so -- the SAME library instance the ctypes binding uses (one per-device state: launch chain,
error word, profiler) -- and registers the fake (meta) implementation of ``lift`` so that FakeTensorMode / torch.compile see its
output shape without a GPU.  A missing compiler AND a missing prebuilt extension is an error on a GPU machine: the forward does
not silently run another host route (MultiView_MPL.use_torch_op(False) asks for the ctypes route explicitly)."""
from __future__ import annotations

import ctypes as C
import threading

import torch

from . import build as _build
from . import cabi

_ops = None
_lock = threading.Lock()
_FAKE_SHAPES = {}        # binding handle -> number of joints (for the fake implementation)


def ops():
    """torch.ops.openmpl_amd with bind / unbind / lift registered and wired to libmpl_hip.so."""
    global _ops
    if _ops is not None:
        return _ops
    with _lock:
        if _ops is not None:
            return _ops
        lib = cabi.load()
        if _build.ext_needs_build():
            try:
                _build.build_torch_ext()
            except _build.CompilerMissing:
                import os
                if not os.path.exists(_build.EXT_PATH):
                    raise
                import warnings
                warnings.warn("mpl_torch_ext.so is older than csrc/torch_ext.cpp and no C++ compiler is available to rebuild it")
        torch.ops.load_library(_build.EXT_PATH)
        o = torch.ops.openmpl_amd
        addr = lambda f: C.cast(f, C.c_void_p).value
        o.set_entry_points(addr(lib.mpl_forward), addr(lib.mpl_forward_workspace_bytes), addr(lib.mpl_hip_error_string),
                           lib.mpl_hip_abi_version())

        @torch.library.register_fake("openmpl_amd::lift")
        def _lift_fake(handle, poses, rays, centers, flags):
            return poses[0].new_empty((poses[0].shape[0], _FAKE_SHAPES.get(int(handle), poses[0].shape[1]), 3), dtype=torch.float32)

        _ops = o
    return _ops


def struct_bytes(s) -> torch.Tensor:
    """A ctypes struct (or array of structs) as a CPU uint8 tensor (a copy)."""
    return torch.frombuffer(bytearray(bytes(s)), dtype=torch.uint8)
