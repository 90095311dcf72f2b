"""Thin Python driver of the C ABI (include/pt.h): one `Renderer` = one PtContext = one GPU.

PyTorch is plumbing here: `bind_tensor()` makes a CUDA tensor the accumulation buffer (so
torch.distributed / RCCL can reduce it) and `use_torch_stream()` enqueues the library's
launches on torch's current stream (so torch.cuda.Event brackets them).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi as abi
from ._lib import PtError,  check, lib


class FlyCamera:
    """The CPU app's mouse/keyboard camera (cpu_pathtracer/src/main.cpp:157-195) driven by a script:
    step(dx, dy, keys) updates the PtCamera and says whether the accumulation must be reset."""

    def __init__(self, camera):
        self.camera = camera
        self.yaw = C.c_double(0.0)
        self.pitch = C.c_double(0.0)

    def step(self, mouse_dx: float = 0.0, mouse_dy: float = 0.0, keys: int = 0) -> bool:
        rc = lib().pt_fly_camera_step(C.byref(self.camera), C.byref(self.yaw), C.byref(self.pitch),
                                      float(mouse_dx), float(mouse_dy), int(keys))
        if rc < 0:
            raise PtError(rc, "pt_fly_camera_step: bad arguments")
        return rc == 1


def builtin_scene_from_library(name: str, width: int, height: int):
    """The C++ host mirror's copy of a built-in scene (pt_builtin_scene); host only."""
    s, c = abi.PtSceneSoA(), abi.PtCamera()
    check(lib().pt_builtin_scene(name.encode(), width, height, C.byref(s), C.byref(c)))
    cam = abi.PtCamera()
    C.memmove(C.byref(cam), C.byref(c), C.sizeof(abi.PtCamera))
    return abi.SceneArrays.from_struct(s), cam


class Renderer:
    def __init__(self, device: int = 0):
        self._L = lib()
        self._ctx = C.c_void_p()
        check(self._L.pt_create(device, C.byref(self._ctx)))
        self.width = self.height = 0
        self._bound = None  # keeps a bound tensor alive

    def close(self):
        if self._ctx:
            self._L.pt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- set-up ---------------------------------------------------------------------------
    def set_scene(self, scene: abi.SceneArrays):
        s = scene.struct()
        check(self._L.pt_upload_scene(self._ctx, C.byref(s)))

    def set_camera(self, cam: abi.PtCamera):
        check(self._L.pt_set_camera(self._ctx, C.byref(cam)))
        if (cam.width, cam.height) != (self.width, self.height):
            self._bound = None  # the library drops a bound buffer when the frame size changes
        self.width, self.height = cam.width, cam.height

    def use_torch_stream(self):
        import torch
        check(self._L.pt_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def use_own_stream(self):
        check(self._L.pt_use_own_stream(self._ctx))

    def set_frames_in_flight(self, n: int):
        """1..8 consecutive sum-mode launches overlap on internal streams (include/pt.h)."""
        check(self._L.pt_set_frames_in_flight(self._ctx, n))

    def frames_in_flight(self) -> int:
        """The number of chains in effect (the request, capped by what GPU_MAX_HW_QUEUES leaves room for)."""
        return int(self._L.pt_frames_in_flight(self._ctx))

    def reduce(self, nccl_comm, root: int = 0):
        """In-place RCCL sum of the accumulation buffer over the communicator's ranks (root < 0: all-reduce)."""
        check(self._L.pt_reduce(self._ctx, C.c_void_p(nccl_comm), int(root)))

    def reduce_scatter(self, nccl_comm, rank: int, world: int):
        """In-place RCCL reduce-scatter: returns (first_pixel, pixel_count) of the slice that now holds all ranks' sums."""
        first, count = C.c_uint32(), C.c_uint32()
        check(self._L.pt_reduce_scatter(self._ctx, C.c_void_p(nccl_comm), rank, world, C.byref(first), C.byref(count)))
        return first.value, count.value

    def flush(self):
        """Fold the in-flight chains into the accumulation buffer (asynchronous)."""
        check(self._L.pt_flush(self._ctx))

    def unbind(self):
        check(self._L.pt_bind_accum(self._ctx, None))
        self._bound = None

    def bind_tensor(self, t):
        """t: contiguous float32 CUDA tensor [H, W, 4]; becomes the accumulation buffer."""
        assert t.is_cuda and t.is_contiguous() and t.numel() == self.width * self.height * 4
        check(self._L.pt_bind_accum(self._ctx, C.c_void_p(t.data_ptr())))
        self._bound = t

    def reset(self):
        check(self._L.pt_reset(self._ctx))

    # -- render ---------------------------------------------------------------------------
    def render(self, sample_begin: int, sample_count: int, params: abi.PtRenderParams):
        check(self._L.pt_render(self._ctx, sample_begin, sample_count, C.byref(params)))

    def render_noalloc(self, dev_ptr: int, sample_begin: int, sample_count: int, params: abi.PtRenderParams):
        """pt_render_noalloc: one serial launch into a caller-owned device buffer; allocates nothing (include/pt.h)."""
        check(self._L.pt_render_noalloc(self._ctx, C.c_void_p(dev_ptr), self.width, self.height, sample_begin, sample_count, C.byref(params)))

    def resolve(self, mode: int):
        check(self._L.pt_resolve(self._ctx, mode))

    def resolve_range(self, mode: int, first_pixel: int, pixel_count: int, out):
        """pt_resolve_range: pixels [first, first + count) of the accumulation buffer into `out`, a CUDA tensor the caller owns."""
        assert out.is_cuda and out.is_contiguous()
        check(self._L.pt_resolve_range(self._ctx, mode, first_pixel, pixel_count, C.c_void_p(out.data_ptr())))

    def sync(self):
        check(self._L.pt_sync(self._ctx))

    def readback(self, mode: int) -> np.ndarray:
        n = self.width * self.height
        if mode == abi.RESOLVE_FLOAT_MEAN:
            out = np.empty((self.height, self.width, 4), np.float32)
        elif mode == abi.RESOLVE_RGB8_PPM:
            out = np.empty((self.height, self.width, 3), np.uint8)
        else:
            out = np.empty((self.height, self.width), np.uint32)
        assert out.size * out.itemsize in (n * 16, n * 3, n * 4)
        check(self._L.pt_readback(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def path_launches(self) -> int:
        """Launches of the path kernel made so far (pt_path_launches)."""
        return int(self._L.pt_path_launches(self._ctx))

    def set_render_ahead(self, on: bool):
        """Render-ahead of the one-frame-at-a-time loop (pt_set_render_ahead): on by default."""
        check(self._L.pt_set_render_ahead(self._ctx, 1 if on else 0))

    def render_ahead_stats(self):
        """(frames launched ahead, frames a call then used) -- pt_render_ahead_stats."""
        import ctypes as C
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._L.pt_render_ahead_stats(self._ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def history_frames(self) -> int:
        """Frames in flight: how many displays older than the newest the last fold pass left (pt_history_frames)."""
        return int(self._L.pt_history_frames(self._ctx))

    def readback_history(self, frames_back: int) -> np.ndarray:
        """The RGBA8 display of the frame `frames_back` (>= 1) before the newest one (pt_readback_history)."""
        out = np.empty((self.height, self.width), np.uint32)
        check(self._L.pt_readback_history(self._ctx, int(frames_back), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def resolve_to_host(self, mode: int) -> np.ndarray:
        self.resolve(mode)
        return self.readback(mode)

    def accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        check(self._L.pt_readback_accum(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def upload_accum(self, a: np.ndarray):
        a = np.ascontiguousarray(a, np.float32)
        check(self._L.pt_upload_accum(self._ctx, a.ctypes.data_as(C.c_void_p), a.nbytes))

    # -- checks / measurement ---------------------------------------------------------------
    def primary_ids(self):
        obj = np.empty((self.height, self.width), np.uint32)
        mat = np.empty((self.height, self.width), np.uint32)
        check(self._L.pt_primary_ids(self._ctx, obj.ctypes.data_as(C.c_void_p), mat.ctypes.data_as(C.c_void_p)))
        return obj, mat

    def counters(self) -> abi.PtCounters:
        c = abi.PtCounters()
        check(self._L.pt_counters(self._ctx, C.byref(c)))
        return c

    def path_length_histogram(self) -> np.ndarray:
        """Samples by the number of path rays they traced (pt_path_length_histogram): 64 bins, the last one open."""
        out = np.zeros(64, np.uint64)
        check(self._L.pt_path_length_histogram(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return out

    def reset_counters(self):
        check(self._L.pt_reset_counters(self._ctx))

    def valu_peak(self, packed: bool, iters: int = 1 << 16) -> float:
        out = C.c_double()
        check(self._L.pt_valu_peak(self._ctx, int(packed), iters, C.byref(out)))
        return out.value

    def launch_render_kernel(self, dev_ptr: int, frame_count: int):
        """The reference's call shape: launchRenderKernel(output, w, h, frame_count, grid, block)."""
        check(self._L.pt_make_current(self._ctx))
        grid = abi.PtDim3((self.width + 15) // 16, (self.height + 15) // 16, 1)  # main.cu:208-212
        self._L.launchRenderKernel(C.c_void_p(dev_ptr), self.width, self.height, frame_count, grid, abi.PtDim3(16, 16, 1))
        err = self._L.pt_last_error()  # the function returns void, like the reference's: this is its error channel
        if err:
            raise PtError(abi.PT_ERR_STATE, err.decode("utf-8", "replace"))


# --- FLOP model of SURVEY.md §8(d) ------------------------------------------------------
FLOP_TABLE = dict(sphere=16, sphere_root=3, plane=14, triangle=51, hit_record=20, camera=30,
                  lambert=70, metal=20, dielectric=60, glossy=50, diffuse_update=12, nee=40, accumulate=10)


def algorithmic_flops(cnt: dict, ns: int, np_: int, nt: int, pixel_launches: int) -> float:
    """Algorithmic FLOPs of a render from its event counters (fixed cost table)."""
    T = FLOP_TABLE
    b = cnt["bounces"]
    f = cnt["samples"] * T["camera"]
    f += cnt["rays"] * (ns * T["sphere"] + np_ * T["plane"] + nt * T["triangle"])
    f += cnt["sphere_roots"] * T["sphere_root"] + cnt["hits"] * T["hit_record"]
    f += b[abi.MAT_LAMBERTIAN] * T["lambert"] + b[abi.MAT_METAL] * T["metal"]
    f += b[abi.MAT_DIELECTRIC] * T["dielectric"] + b[abi.MAT_GLOSSY] * T["glossy"]
    f += cnt["diffuse_updates"] * T["diffuse_update"] + cnt["nee_samples"] * T["nee"]
    f += pixel_launches * T["accumulate"]
    return float(f)


# The same table applied to the `direct` integrator (renderRegion, cpu_pathtracer/src/main.cpp:14-53), FP64: a primary ray and,
# per light sample whose cosine is positive, a shadow ray test every sphere, plane and triangle (Scene::intersect /
# Scene::isVisible are brute force too); a light sample (Sphere::getRandomPoint + direction + cosine, Scene.h:97-112,
# ImplicitObject.h:111-125) is priced like the path integrator's light sample, its contribution (colour * cos * 50 / d^2,
# main.cpp:36-41) like a throughput update.  Sphere roots are not counted (the kernel does not count them): a lower bound.
def direct_flops(cnt: dict, ns: int, np_: int, nt: int, pixel_launches: int) -> float:
    T = FLOP_TABLE
    f = cnt["samples"] * T["camera"]
    f += cnt["rays"] * (ns * T["sphere"] + np_ * T["plane"] + nt * T["triangle"])
    f += cnt["hits"] * T["hit_record"] + cnt["nee_samples"] * T["nee"] + cnt["shadow_rays"] * T["diffuse_update"]
    f += pixel_launches * T["accumulate"]
    return float(f)
