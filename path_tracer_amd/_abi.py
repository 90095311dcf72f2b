"""ctypes mirror of include/pt.h (POD types and constants only).

Kept separate from the loader so that CPU-only tests (and the oracle binding in
tests/) can build PtSceneSoA / PtCamera values without touching the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

# --- constants (include/pt.h) -------------------------------------------------
PT_OK = 0
PT_ERR_INVALID, PT_ERR_HIP, PT_ERR_STATE, PT_ERR_CAPACITY = -1, -2, -3, -4

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_GLOSSY, MAT_EMISSIVE = range(5)
CAMERA_CUDA, CAMERA_CPU = 0, 1
INTEGRATOR_PATH, INTEGRATOR_DIRECT = 0, 1
FLAG_NEE_MIS, FLAG_LIGHT_COLOUR, FLAG_COUNTERS = 1, 2, 4
KEY_W, KEY_S, KEY_A, KEY_D, KEY_SPACE, KEY_CTRL = 1, 2, 4, 8, 16, 32  # pt_fly_camera_step
ACC_SUM, ACC_RUNNING_MEAN = 0, 1
RESOLVE_RGBA8_CUDA, RESOLVE_RGBA8_CPU, RESOLVE_FLOAT_MEAN, RESOLVE_RGB8_PPM = range(4)
MISS = 0xFFFFFFFF

_PD = C.POINTER(C.c_double)
_PU = C.POINTER(C.c_uint32)

_SCENE_FIELDS = [
    ("num_spheres", C.c_uint32), ("num_planes", C.c_uint32),
    ("num_triangles", C.c_uint32), ("num_materials", C.c_uint32),
    ("sph_cx", _PD), ("sph_cy", _PD), ("sph_cz", _PD), ("sph_r", _PD),
    ("sph_mat", _PU), ("sph_obj", _PU), ("sph_emissive", _PU),
    ("sph_em_r", _PD), ("sph_em_g", _PD), ("sph_em_b", _PD),
    ("pl_px", _PD), ("pl_py", _PD), ("pl_pz", _PD), ("pl_nx", _PD), ("pl_ny", _PD), ("pl_nz", _PD),
    ("pl_mat", _PU), ("pl_obj", _PU),
    ("tri_v0x", _PD), ("tri_v0y", _PD), ("tri_v0z", _PD),
    ("tri_v1x", _PD), ("tri_v1y", _PD), ("tri_v1z", _PD),
    ("tri_v2x", _PD), ("tri_v2y", _PD), ("tri_v2z", _PD),
    ("tri_mat", _PU), ("tri_obj", _PU),
    ("mat_type", _PU),
    ("mat_albedo_r", _PD), ("mat_albedo_g", _PD), ("mat_albedo_b", _PD),
    ("mat_roughness", _PD), ("mat_metallic", _PD), ("mat_ior", _PD),
    ("mat_em_r", _PD), ("mat_em_g", _PD), ("mat_em_b", _PD),
]


class PtSceneSoA(C.Structure):
    _fields_ = _SCENE_FIELDS


class PtCamera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3),
        ("fov_deg", C.c_double), ("aspect_ratio", C.c_double), ("aperture", C.c_double),
        ("focus_dist", C.c_double),
        ("width", C.c_uint32), ("height", C.c_uint32), ("use_dof", C.c_uint32),
        ("model", C.c_uint32), ("normalize_up", C.c_uint32),
    ]


class PtRenderParams(C.Structure):
    _fields_ = [
        ("integrator", C.c_uint32), ("max_depth", C.c_uint32), ("rr_depth", C.c_uint32),
        ("seed", C.c_uint32), ("flags", C.c_uint32), ("acc_mode", C.c_uint32),
        ("fused_rgba8", C.c_uint32), ("pixels_per_wave", C.c_uint32),
    ]


class PtDim3(C.Structure):
    """dim3 of the reference's launchRenderKernel call (raytracer_kernel.cuh:51), passed by value."""
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("z", C.c_uint32)]


class PtCounters(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("rays", C.c_uint64), ("hits", C.c_uint64),
        ("sphere_roots", C.c_uint64), ("bounces", C.c_uint64 * 5),
        ("diffuse_updates", C.c_uint64), ("nee_samples", C.c_uint64), ("shadow_rays", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        d = {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "bounces"}
        d["bounces"] = [int(v) for v in self.bounces]
        return d


DEFAULT_SEED = 0x5EED0001


def render_params(integrator=INTEGRATOR_PATH, max_depth=50, rr_depth=3, seed=DEFAULT_SEED, flags=0,
                  acc_mode=ACC_SUM, fused_rgba8=0, pixels_per_wave=0) -> PtRenderParams:
    return PtRenderParams(integrator, max_depth, rr_depth, seed, flags, acc_mode, fused_rgba8, pixels_per_wave)


class SceneArrays:
    """Owns the numpy arrays behind one PtSceneSoA (keeps them alive)."""

    _D = [n for n, t in _SCENE_FIELDS if t is _PD]
    _U = [n for n, t in _SCENE_FIELDS if t is _PU]

    def __init__(self, **arrays):
        self.arrays: dict[str, np.ndarray] = {}
        for n in self._D:
            self.arrays[n] = np.ascontiguousarray(arrays.get(n, ()), dtype=np.float64)
        for n in self._U:
            self.arrays[n] = np.ascontiguousarray(arrays.get(n, ()), dtype=np.uint32)
        a = self.arrays
        self.num_spheres, self.num_planes = len(a["sph_cx"]), len(a["pl_px"])
        self.num_triangles, self.num_materials = len(a["tri_v0x"]), len(a["mat_type"])
        for n, k in [(n, self.num_spheres) for n in a if n.startswith("sph_")] + \
                    [(n, self.num_planes) for n in a if n.startswith("pl_")] + \
                    [(n, self.num_triangles) for n in a if n.startswith("tri_")] + \
                    [(n, self.num_materials) for n in a if n.startswith("mat_")]:
            if len(a[n]) != k:
                raise ValueError(f"scene array {n} has {len(a[n])} entries, expected {k}")

    @classmethod
    def from_struct(cls, s: PtSceneSoA) -> "SceneArrays":
        """Deep-copies a PtSceneSoA filled by the library (pt_builtin_scene)."""
        counts = {"sph_": s.num_spheres, "pl_": s.num_planes, "tri_": s.num_triangles, "mat_": s.num_materials}
        arrays = {}
        for n, t in _SCENE_FIELDS:
            if t not in (_PD, _PU):
                continue
            k = next(v for p, v in counts.items() if n.startswith(p))
            ptr = getattr(s, n)
            arrays[n] = np.array([ptr[i] for i in range(k)], dtype=np.float64 if t is _PD else np.uint32)
        return cls(**arrays)

    def struct(self) -> PtSceneSoA:
        s = PtSceneSoA()
        s.num_spheres, s.num_planes = self.num_spheres, self.num_planes
        s.num_triangles, s.num_materials = self.num_triangles, self.num_materials
        for n in self._D:
            setattr(s, n, self.arrays[n].ctypes.data_as(_PD))
        for n in self._U:
            setattr(s, n, self.arrays[n].ctypes.data_as(_PU))
        return s

    def __eq__(self, other) -> bool:
        return isinstance(other, SceneArrays) and all(
            np.array_equal(self.arrays[n], other.arrays[n]) for n in self.arrays)


def make_camera(origin, target, up, fov_deg, width, height, model, aspect_ratio=None, aperture=0.0,
                focus_dist=10.0, use_dof=0, normalize_up=1) -> PtCamera:
    c = PtCamera()
    c.origin[:] = origin
    c.target[:] = target
    c.up[:] = up
    c.fov_deg = fov_deg
    c.aspect_ratio = (16.0 / 9.0) if aspect_ratio is None else aspect_ratio
    c.aperture, c.focus_dist = aperture, focus_dist
    c.width, c.height, c.use_dof, c.model, c.normalize_up = width, height, use_dof, model, normalize_up
    return c
