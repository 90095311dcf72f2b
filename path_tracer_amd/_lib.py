"""Loader of the product library (path_tracer_amd/lib/libpt_amd.so) and its ctypes signatures.

There is no fallback: if the HIP library is missing or fails to load, importing a symbol
through `lib()` raises.  Nothing here knows about the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _abi as abi

# PT_AMD_LIB selects another build of the SAME library (kernel-variant A/B runs); never a fallback
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpt_amd.so")

# every extern "C" symbol include/pt.h declares (tests check the export list against the header)
SYMBOLS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_abi_version", "pt_set_stream", "pt_use_own_stream", "pt_upload_scene",
    "pt_set_camera", "pt_resize", "pt_bind_accum", "pt_reset", "pt_render", "pt_resolve", "pt_readback",
    "pt_readback_accum", "pt_history_frames", "pt_readback_history", "pt_upload_accum", "pt_sync", "pt_accum_device_ptr", "pt_display_device_ptr",
    "pt_primary_ids", "pt_counters", "pt_reset_counters", "pt_valu_peak", "pt_make_current",
    "launchRenderKernel", "pt_builtin_scene", "pt_set_frames_in_flight", "pt_frames_in_flight", "pt_path_launches", "pt_flush", "pt_set_render_ahead", "pt_render_ahead_stats",
    "pt_fly_camera_step", "pt_reduce", "pt_render_noalloc", "pt_path_length_histogram", "pt_resolve_range", "pt_reduce_scatter",
]

_lib = None


class PtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pt error {code}: {message}")
        self.code = code


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950); "
                          "there is no CPU fallback for the render path")
    # torch bundles its own HIP runtime (soname libamdhip64.so.7).  Loading it FIRST makes this
    # library's NEEDED entry resolve to that same copy; the other order puts two HIP runtimes in
    # one process and torch then reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    ctx, u32, vp = C.c_void_p, C.c_uint32, C.c_void_p
    L.pt_create.argtypes = [C.c_int, C.POINTER(ctx)]
    L.pt_destroy.argtypes = [ctx]
    L.pt_last_error.restype = C.c_char_p
    L.pt_abi_version.restype = u32
    L.pt_set_stream.argtypes = [ctx, vp]
    L.pt_use_own_stream.argtypes = [ctx]
    L.pt_upload_scene.argtypes = [ctx, C.POINTER(abi.PtSceneSoA)]
    L.pt_set_camera.argtypes = [ctx, C.POINTER(abi.PtCamera)]
    L.pt_resize.argtypes = [ctx, u32, u32]
    L.pt_bind_accum.argtypes = [ctx, vp]
    L.pt_reset.argtypes = [ctx]
    L.pt_render.argtypes = [ctx, u32, u32, C.POINTER(abi.PtRenderParams)]
    L.pt_render_noalloc.argtypes = [ctx, vp, u32, u32, u32, u32, C.POINTER(abi.PtRenderParams)]
    L.pt_resolve.argtypes = [ctx, u32]
    L.pt_resolve_range.argtypes = [ctx, u32, u32, u32, vp]
    L.pt_readback.argtypes = [ctx, vp, C.c_size_t]
    L.pt_readback_accum.argtypes = [ctx, vp, C.c_size_t]
    L.pt_path_launches.argtypes = [ctx]
    L.pt_path_launches.restype = C.c_uint64
    L.pt_set_render_ahead.argtypes = [ctx, C.c_uint32]
    L.pt_render_ahead_stats.argtypes = [ctx, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pt_history_frames.argtypes = [ctx]
    L.pt_history_frames.restype = C.c_uint32
    L.pt_readback_history.argtypes = [ctx, C.c_uint32, vp, C.c_size_t]
    L.pt_upload_accum.argtypes = [ctx, vp, C.c_size_t]
    L.pt_sync.argtypes = [ctx]
    L.pt_accum_device_ptr.argtypes = [ctx]
    L.pt_accum_device_ptr.restype = vp
    L.pt_display_device_ptr.argtypes = [ctx]
    L.pt_display_device_ptr.restype = vp
    L.pt_primary_ids.argtypes = [ctx, vp, vp]
    L.pt_counters.argtypes = [ctx, C.POINTER(abi.PtCounters)]
    L.pt_reset_counters.argtypes = [ctx]
    L.pt_path_length_histogram.argtypes = [ctx, vp]
    L.pt_valu_peak.argtypes = [ctx, u32, u32, C.POINTER(C.c_double)]
    L.pt_make_current.argtypes = [ctx]
    L.launchRenderKernel.argtypes = [vp, u32, u32, u32, abi.PtDim3, abi.PtDim3]  # grid, block by value, like dim3
    L.launchRenderKernel.restype = None
    L.pt_set_frames_in_flight.argtypes = [ctx, u32]
    L.pt_frames_in_flight.argtypes = [ctx]
    L.pt_frames_in_flight.restype = u32
    L.pt_flush.argtypes = [ctx]
    L.pt_reduce.argtypes = [ctx, vp, C.c_int]
    L.pt_reduce_scatter.argtypes = [ctx, vp, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    L.pt_fly_camera_step.argtypes = [C.POINTER(abi.PtCamera), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, u32]
    L.pt_builtin_scene.argtypes = [C.c_char_p, u32, u32, C.POINTER(abi.PtSceneSoA), C.POINTER(abi.PtCamera)]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise PtError(rc, lib().pt_last_error().decode("utf-8", "replace"))
