"""ctypes binding of oracle/libpt_oracle.so — the CPU checker (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from path_tracer_amd import _abi as abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libpt_oracle.so")

_PF = C.POINTER(C.c_float)
_PD = C.POINTER(C.c_double)
_PU = C.POINTER(C.c_uint32)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        S, K, P = C.POINTER(abi.PtSceneSoA), C.POINTER(abi.PtCamera), C.POINTER(abi.PtRenderParams)
        lib.pto_hash.restype = C.c_uint32
        lib.pto_hash.argtypes = [C.c_uint32]
        lib.pto_path_key.restype = C.c_uint32
        lib.pto_path_key.argtypes = [C.c_uint32] * 3
        lib.pto_draw.restype = C.c_uint32
        lib.pto_draw.argtypes = [C.c_uint32] * 3
        lib.pto_fmath.argtypes = [C.c_double, C.c_double, _PD]
        lib.pto_primary_ids.argtypes = [S, K, _PU, _PU]
        lib.pto_primary_ids_rect.argtypes = [S, K, C.c_uint32, _PD, _PU, _PU, _PU, _PU]
        lib.pto_render_direct.argtypes = [S, K, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P, _PF]
        lib.pto_direct_components.argtypes = [S, K, C.c_uint32, C.c_uint32, C.c_uint32, P, _PD]
        lib.pto_render_path.argtypes = [S, K, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_int, _PF,
                                        C.POINTER(abi.PtCounters)]
        lib.pto_path_length_histogram.argtypes = [S, K, C.c_uint32, C.c_uint32, P, C.c_int, C.POINTER(C.c_uint64)]
        lib.pto_path_sample.argtypes = [S, K, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_int, _PD]
        lib.pto_path_stats.argtypes = [S, K, C.c_uint32, C.c_uint32, P, _PD, _PD]
        lib.pto_resolve.argtypes = [_PF, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]

    @staticmethod
    def _ck(rc):
        if rc != 0:
            raise RuntimeError(f"oracle call failed: {rc}")

    def fmath(self, x: float, y: float):
        """(sin x, cos x, acos y) as the `direct` light sample evaluates them (pt_oracle.c, fm_sincos / fm_acos)."""
        out = (C.c_double * 3)()
        self._ck(self.lib.pto_fmath(x, y, out))
        return out[0], out[1], out[2]

    def primary_ids(self, scene: abi.SceneArrays, cam: abi.PtCamera):
        n = cam.width * cam.height
        obj, mat = np.empty(n, np.uint32), np.empty(n, np.uint32)
        s = scene.struct()
        self._ck(self.lib.pto_primary_ids(C.byref(s), C.byref(cam), obj.ctypes.data_as(_PU), mat.ctypes.data_as(_PU)))
        return obj.reshape(cam.height, cam.width), mat.reshape(cam.height, cam.width)

    def primary_ids_rect(self, scene: abi.SceneArrays, cam: abi.PtCamera, rects):
        """Primary ids with the rectangles [(origin, u, v, obj id, material id), ...] tested by the reference's
        Rectangle::intersect instead of through their two triangles (which are skipped)."""
        n = cam.width * cam.height
        obj, mat = np.empty(n, np.uint32), np.empty(n, np.uint32)
        ouv = np.ascontiguousarray([list(o) + list(u) + list(v) for o, u, v, _, _ in rects], np.float64).reshape(-1)
        ro = np.ascontiguousarray([r[3] for r in rects], np.uint32)
        rm = np.ascontiguousarray([r[4] for r in rects], np.uint32)
        s = scene.struct()
        self._ck(self.lib.pto_primary_ids_rect(C.byref(s), C.byref(cam), len(rects), ouv.ctypes.data_as(_PD), ro.ctypes.data_as(_PU),
                                               rm.ctypes.data_as(_PU), obj.ctypes.data_as(_PU), mat.ctypes.data_as(_PU)))
        return obj.reshape(cam.height, cam.width), mat.reshape(cam.height, cam.width)

    def render_direct(self, scene, cam, s0, n, params, accum=None, rows=None):
        if accum is None:
            accum = np.zeros((cam.height, cam.width, 4), np.float32)
        y0, y1 = rows if rows else (0, cam.height)
        s = scene.struct()
        self._ck(self.lib.pto_render_direct(C.byref(s), C.byref(cam), y0, y1, s0, n, C.byref(params),
                                            accum.ctypes.data_as(_PF)))
        return accum

    def direct_components(self, scene, cam, x, y, smp, params):
        out = np.zeros(8, np.float64)
        s = scene.struct()
        self._ck(self.lib.pto_direct_components(C.byref(s), C.byref(cam), x, y, smp, C.byref(params),
                                                out.ctypes.data_as(_PD)))
        return out

    def render_path(self, scene, cam, s0, n, params, precision=32, accum=None, rows=None, counters=False):
        if accum is None:
            accum = np.zeros((cam.height, cam.width, 4), np.float32)
        y0, y1 = rows if rows else (0, cam.height)
        cnt = abi.PtCounters() if counters else None
        s = scene.struct()
        self._ck(self.lib.pto_render_path(C.byref(s), C.byref(cam), y0, y1, s0, n, C.byref(params), precision,
                                          accum.ctypes.data_as(_PF), C.byref(cnt) if counters else None))
        return (accum, cnt) if counters else accum

    def path_length_histogram(self, scene, cam, s0, n, params, precision=32):
        """Samples by the number of path rays they traced: 64 bins, the last one open."""
        out = np.zeros(64, np.uint64)
        s = scene.struct()
        self._ck(self.lib.pto_path_length_histogram(C.byref(s), C.byref(cam), s0, n, C.byref(params), precision,
                                                    out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def path_sample(self, scene, cam, x, y, smp, params, precision=64):
        out = np.zeros(3, np.float64)
        s = scene.struct()
        self._ck(self.lib.pto_path_sample(C.byref(s), C.byref(cam), x, y, smp, C.byref(params), precision,
                                          out.ctypes.data_as(_PD)))
        return out

    def path_stats(self, scene, cam, s0, n, params):
        npix = cam.width * cam.height
        mean, var = np.zeros((npix, 3)), np.zeros((npix, 3))
        s = scene.struct()
        self._ck(self.lib.pto_path_stats(C.byref(s), C.byref(cam), s0, n, C.byref(params),
                                         mean.ctypes.data_as(_PD), var.ctypes.data_as(_PD)))
        shape = (cam.height, cam.width, 3)
        return mean.reshape(shape), var.reshape(shape)

    def resolve(self, accum, acc_mode, mode):
        accum = np.ascontiguousarray(accum, np.float32)
        npix = accum.size // 4
        if mode == abi.RESOLVE_FLOAT_MEAN:
            out = np.empty((npix, 4), np.float32)
        elif mode == abi.RESOLVE_RGB8_PPM:
            out = np.empty((npix, 3), np.uint8)
        else:
            out = np.empty(npix, np.uint32)
        self._ck(self.lib.pto_resolve(accum.ctypes.data_as(_PF), npix, acc_mode, mode, out.ctypes.data_as(C.c_void_p)))
        return out


def load_oracle(build_if_missing=True) -> Oracle:
    if not os.path.exists(ORACLE_SO) and build_if_missing:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return Oracle(C.CDLL(ORACLE_SO))
