"""Scene construction in the reference's own vocabulary, flattened to PtSceneSoA.

`SceneBuilder` mirrors the host calls of the reference's app shells
(`new Sphere_t(center, radius, material)`, `new Plane_t(point, normal, material)`,
`new Triangle_t(v0, v1, v2, material)`, `obj->makeEmissive(colour, strength)`,
`SceneManager::addObject` — cuda_pathtracer/src/main.cu:99-200; and the CPU
`std::make_shared<Sphere>(center, radius, colour)` — cpu_pathtracer/src/main.cpp:67-131)
and `flatten()` replaces the AoS flatten + cudaMemcpyToSymbol of main.cu:214-278.

The built-in scenes restate the literals of the two `main()`s and the two synthetic
configurations of SURVEY.md §8(d).  The C++ host mirror (path_tracer_amd/host) restates
them independently; tests/test_scenes.py checks the two agree bit for bit.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _abi as abi

f32 = np.float32


def _hash(x: int) -> int:
    """The counter-RNG finaliser, also used to lay out random4096."""
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


@dataclass
class Material:
    type: int
    albedo: tuple = (0.0, 0.0, 0.0)
    roughness: float = 0.0
    metallic: float = 0.0
    ior: float = 1.0
    emission: tuple = (0.0, 0.0, 0.0)  # colour * strength


def Lambertian(albedo):  # Material.cuh:41
    return Material(abi.MAT_LAMBERTIAN, albedo=_rgb(albedo))


def Metal(albedo, roughness):  # Material.cuh:92
    return Material(abi.MAT_METAL, albedo=_rgb(albedo), roughness=roughness)


def Dielectric(ior):  # Material.cuh:147
    return Material(abi.MAT_DIELECTRIC, ior=ior)


def Glossy(albedo, roughness, metallic=1.0):  # Material.cuh:185
    return Material(abi.MAT_GLOSSY, albedo=_rgb(albedo), roughness=roughness, metallic=metallic)


def Emissive(colour, strength):  # Material.cuh:241
    c = _rgb(colour)
    return Material(abi.MAT_EMISSIVE, emission=(c[0] * strength, c[1] * strength, c[2] * strength))


def _rgb(c):
    return (c, c, c) if np.isscalar(c) else tuple(c)


@dataclass
class _Obj:
    kind: str
    data: tuple
    mat: int
    emissive: bool = False
    emission: tuple = (0.0, 0.0, 0.0)

    def makeEmissive(self, colour, strength):  # ImplicitObject.cuh:31-35 / ImplicitObject.h:30-34
        c = _rgb(colour)
        self.emissive = True
        self.emission = (c[0] * strength, c[1] * strength, c[2] * strength)
        return self


@dataclass
class SceneBuilder:
    materials: list = field(default_factory=list)
    objects: list = field(default_factory=list)

    def material(self, m: Material) -> int:
        self.materials.append(m)
        return len(self.materials) - 1

    def addObject(self, obj: _Obj) -> _Obj:  # SceneManager::addObject, Scene.cuh:266
        self.objects.append(obj)
        return obj

    def Sphere(self, center, radius, material: int) -> _Obj:
        return _Obj("sphere", (tuple(center), radius), material)

    def Plane(self, point, normal, material: int) -> _Obj:
        return _Obj("plane", (tuple(point), tuple(normal)), material)

    def Triangle(self, v0, v1, v2, material: int) -> _Obj:
        return _Obj("triangle", (tuple(v0), tuple(v1), tuple(v2)), material)

    def Rectangle(self, origin, u, v, material: int) -> _Obj:
        """Rectangle_t(origin, u, v, ...) (Polygon.cuh:113-168): two triangles with normal cross(u, v) that share
        the object id.  The reference's alpha/beta test describes the parallelogram only for perpendicular edges."""
        o, u, v = (np.asarray(x, np.float64) for x in (origin, u, v))
        if not (u @ u > 0 and v @ v > 0) or (u @ v) ** 2 > 1e-12 * (u @ u) * (v @ v):
            raise ValueError("Rectangle: edges must be non-zero and perpendicular")
        a, b, c, d = o, o + u, o + u + v, o + v
        return _Obj("polygon", (tuple(map(tuple, (a, b, c))), tuple(map(tuple, (a, c, d)))), material)

    def Tetrahedron(self, vertices, material: int) -> _Obj:
        """Four triangles in the face order of Tetrahedron.h:16-21 / main.cu:183-199."""
        p = [tuple(x) for x in vertices]
        return _Obj("polygon", tuple((p[i], p[j], p[k]) for i, j, k in ((0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3))), material)

    def regular_tetrahedron(self, radius, material: int) -> _Obj:
        """Tetrahedron::createRegular, Tetrahedron.h:38-50."""
        a, h = radius * 2.0 / np.sqrt(6.0), radius * 3.0 / np.sqrt(6.0)
        return self.Tetrahedron([(0, -radius, -h / 3), (-a / 2, radius / 3, -h / 3), (a / 2, radius / 3, -h / 3), (0, 0, 2 * h / 3)], material)

    def flatten(self) -> abi.SceneArrays:
        a = {n: [] for n, _ in abi._SCENE_FIELDS if not n.startswith("num_")}
        for idx, o in enumerate(self.objects):
            if o.kind == "polygon":
                for tri in o.data:
                    for k, v in enumerate(tri):
                        a[f"tri_v{k}x"].append(v[0]); a[f"tri_v{k}y"].append(v[1]); a[f"tri_v{k}z"].append(v[2])
                    a["tri_mat"].append(o.mat); a["tri_obj"].append(idx)
                continue
            if o.kind == "sphere":
                (cx, cy, cz), r = o.data
                a["sph_cx"].append(cx); a["sph_cy"].append(cy); a["sph_cz"].append(cz); a["sph_r"].append(r)
                a["sph_mat"].append(o.mat); a["sph_obj"].append(idx); a["sph_emissive"].append(int(o.emissive))
                a["sph_em_r"].append(o.emission[0]); a["sph_em_g"].append(o.emission[1]); a["sph_em_b"].append(o.emission[2])
            elif o.kind == "plane":
                (px, py, pz), (nx, ny, nz) = o.data
                a["pl_px"].append(px); a["pl_py"].append(py); a["pl_pz"].append(pz)
                a["pl_nx"].append(nx); a["pl_ny"].append(ny); a["pl_nz"].append(nz)
                a["pl_mat"].append(o.mat); a["pl_obj"].append(idx)
            else:
                for k, v in enumerate(o.data):
                    a[f"tri_v{k}x"].append(v[0]); a[f"tri_v{k}y"].append(v[1]); a[f"tri_v{k}z"].append(v[2])
                a["tri_mat"].append(o.mat); a["tri_obj"].append(idx)
        for m in self.materials:
            a["mat_type"].append(m.type)
            a["mat_albedo_r"].append(m.albedo[0]); a["mat_albedo_g"].append(m.albedo[1]); a["mat_albedo_b"].append(m.albedo[2])
            a["mat_roughness"].append(m.roughness); a["mat_metallic"].append(m.metallic); a["mat_ior"].append(m.ior)
            a["mat_em_r"].append(m.emission[0]); a["mat_em_g"].append(m.emission[1]); a["mat_em_b"].append(m.emission[2])
        return abi.SceneArrays(**a)


# ------------------------------------------------------------------ built-ins ---
def cpu_default(width=1280, height=720, running_app_camera=False):
    """cpu_pathtracer/src/main.cpp:63-131 (double literals). Objects carry a colour, held
    here as a Lambertian material; the light gets an Emissive material so the `path`
    integrator can run on this scene too."""
    b = SceneBuilder()
    half = 4.0 / 2.0

    def coloured(c):
        return b.material(Lambertian(c))

    b.addObject(b.Plane((0, 0, -half), (0, 0, 1), coloured((0.9, 0.9, 0.9))))
    b.addObject(b.Plane((-half, 0, 0), (1, 0, 0), coloured((0.9, 0.2, 0.2))))
    b.addObject(b.Plane((half, 0, 0), (-1, 0, 0), coloured((0.2, 0.9, 0.2))))
    b.addObject(b.Plane((0, -half, 0), (0, 1, 0), coloured((0.9, 0.9, 0.9))))
    b.addObject(b.Plane((0, half, 0), (0, -1, 0), coloured((0.9, 0.9, 0.9))))
    b.addObject(b.Sphere((-1.0, -half + 0.8, -half + 1.0), 0.8, coloured((0.95, 0.95, 0.95))))
    b.addObject(b.Sphere((1.0, -half + 0.4, -half + 1.0), 0.4, coloured((0.7, 0.3, 0.3))))
    light = b.Sphere((0, half - 0.1, 0), 0.2, b.material(Emissive((1.0, 1.0, 1.0), 100.0)))
    light.makeEmissive((1.0, 1.0, 1.0), 100.0)
    b.addObject(light)
    pos = (0.0, 0.0, half * 0.8)
    if running_app_camera:
        # main.cpp:162-168: yaw = pitch = 0 gives forward (1,0,0); lookAt(from,to,up) does not normalise up
        cam = abi.make_camera(pos, (pos[0] + 1.0, pos[1], pos[2]), (0, 1, 0), 90.0, width, height,
                              abi.CAMERA_CPU, normalize_up=0)
    else:
        cam = abi.make_camera(pos, (0, 0, -1), (0, 1, 0), 90.0, width, height, abi.CAMERA_CPU)
    return b.flatten(), cam


def _palette(b: SceneBuilder):
    """MaterialFactory::createMaterialsOnDevice, MaterialFactory.cuh:64-71 (float literals)."""
    return dict(
        white=b.material(Lambertian(float(f32(0.73)))),
        red=b.material(Lambertian((float(f32(0.65)), float(f32(0.05)), float(f32(0.05))))),
        green=b.material(Lambertian((float(f32(0.12)), float(f32(0.45)), float(f32(0.15))))),
        light=b.material(Emissive(1.0, 15.0)),
        glass=b.material(Dielectric(1.5)),
        metal=b.material(Metal(float(f32(0.95)), 0.0)),
        glossy=b.material(Glossy(float(f32(0.7)), float(f32(0.2)), float(f32(0.8)))),
        mirror=b.material(Metal(float(f32(0.98)), 0.0)),
    )


def _cornell_walls(b: SceneBuilder, m):
    """main.cu:102-121"""
    b.addObject(b.Plane((0.0, -2.0, 0.0), (0.0, 1.0, 0.0), m["white"]))
    b.addObject(b.Plane((0.0, 2.0, 0.0), (0.0, -1.0, 0.0), m["white"]))
    b.addObject(b.Plane((-2.0, 0.0, 0.0), (1.0, 0.0, 0.0), m["red"]))
    b.addObject(b.Plane((2.0, 0.0, 0.0), (-1.0, 0.0, 0.0), m["green"]))
    b.addObject(b.Plane((0.0, 0.0, -2.0), (0.0, 0.0, 1.0), m["mirror"]))
    b.addObject(b.Plane((0.0, 0.0, 2.0), (0.0, 0.0, -1.0), m["mirror"]))


def _cornell_lights(b: SceneBuilder, m):
    """main.cu:154-168"""
    F = lambda v: float(f32(v))
    b.addObject(b.Sphere((0.0, F(1.8), 0.0), F(0.3), m["light"]).makeEmissive(1.0, 15.0))
    b.addObject(b.Sphere((F(1.8), 0.0, F(-1.8)), F(0.15), m["light"]).makeEmissive((F(0.2), F(0.2), 1.0), 10.0))
    b.addObject(b.Sphere((F(-1.8), 0.0, F(-1.8)), F(0.15), m["light"]).makeEmissive((1.0, 0.5, 0.0), 10.0))


def _cuda_camera(width, height):
    """main.cu:68-84: position (0,0,half_size*0.8f), target (0,0,-1), up (0,-1,0), fov 90."""
    z = float(f32(2.0) * f32(0.8))
    return abi.make_camera((0, 0, z), (0, 0, -1), (0, -1, 0), 90.0, width, height, abi.CAMERA_CUDA,
                           aspect_ratio=float(f32(width) / f32(height)), use_dof=1)


def cuda_default(width=1280, height=720):
    """cuda_pathtracer/src/main.cu:64-200."""
    F = lambda v: float(f32(v))
    b = SceneBuilder()
    m = _palette(b)
    _cornell_walls(b, m)
    b.addObject(b.Sphere((0.0, -1.0, 0.0), F(0.7), m["glass"]))
    b.addObject(b.Sphere((F(-1.2), F(-1.2), -0.5), F(0.6), m["glossy"]))
    b.addObject(b.Sphere((F(1.2), F(-1.2), -0.5), F(0.6), m["metal"]))
    sr = 0.25
    b.addObject(b.Sphere((0.0, F(-1.4), F(-1.2)), sr, m["glass"]))
    b.addObject(b.Sphere((F(-0.6), F(-1.4), -1.0), sr, m["glossy"]))
    b.addObject(b.Sphere((F(0.6), F(-1.4), -1.0), sr, m["glossy"]))
    _cornell_lights(b, m)
    # tetrahedron, main.cu:170-200 (float arithmetic for the translation)
    tr = f32(0.6)
    verts = [(f32(0), tr, f32(0)), (-tr, -tr, tr), (tr, -tr, tr), (f32(0), -tr, -tr)]
    pos = (f32(-0.6), f32(0.0), f32(-1.0))
    verts = [tuple(float(f32(v[k] + pos[k])) for k in range(3)) for v in verts]
    for i0, i1, i2 in ((0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)):
        b.addObject(b.Triangle(verts[i0], verts[i1], verts[i2], m["glass"]))
    return b.flatten(), _cuda_camera(width, height)


def glass_stress(width=1280, height=720):
    """nested refractive spheres (max depth 32 is a render parameter)."""
    F = lambda v: float(f32(v))
    b = SceneBuilder()
    m = _palette(b)
    _cornell_walls(b, m)
    for radius, ior in ((1.0, 1.5), (0.8, 1.33), (0.6, 1.5), (0.4, 2.4)):
        b.addObject(b.Sphere((0.0, F(-0.8), F(-0.3)), F(radius), b.material(Dielectric(F(ior)))))
    for sx in (1.3, -1.3):
        b.addObject(b.Sphere((F(sx), F(-1.65), F(-0.8)), F(0.35), m["glass"]))
    _cornell_lights(b, m)
    return b.flatten(), _cuda_camera(width, height)


def _u01(i: int, k: int) -> float:
    return (_hash(_hash(0xC0FFEE + i) + k) >> 8) * (1.0 / 16777216.0)


def random4096(width=3840, height=2160, count=4096):
    """`count` hash-placed spheres over a floor plane, mixed
    materials by i mod 20, every 64th an emitter of strength 15 (max depth 16 is a render
    parameter).  All values are rounded to float, like every literal of the CUDA app."""
    F = lambda v: float(f32(v))
    b = SceneBuilder()
    b.addObject(b.Plane((0.0, -2.0, 0.0), (0.0, 1.0, 0.0), b.material(Lambertian(F(0.73)))))
    for i in range(count):
        u = [_u01(i, k) for k in range(7)]
        c = (F(-24.0 + 48.0 * u[0]), F(-1.5 + 7.5 * u[1]), F(-48.0 + 48.0 * u[2]))
        r = F(0.15 + 0.45 * u[3])
        alb = (F(0.2 + 0.7 * u[4]), F(0.2 + 0.7 * u[5]), F(0.2 + 0.7 * u[6]))
        if i % 64 == 0:
            col = (F(0.2 + 0.8 * u[4]), F(0.2 + 0.8 * u[5]), F(0.2 + 0.8 * u[6]))
            b.addObject(b.Sphere(c, F(0.4), b.material(Emissive(col, 15.0))).makeEmissive(col, 15.0))
            continue
        k = i % 20
        if k < 10:
            mat = Lambertian(alb)
        elif k < 14:
            mat = Metal(alb, 0.0)
        elif k < 17:
            mat = Glossy(alb, F(0.2), F(0.8))
        else:
            mat = Dielectric(1.5)
        b.addObject(b.Sphere(c, r, b.material(mat)))
    cam = abi.make_camera((0, 2, 6), (0, 1, -20), (0, 1, 0), 60.0, width, height, abi.CAMERA_CUDA,
                          aspect_ratio=float(f32(width) / f32(height)))
    return b.flatten(), cam


BUILTIN = {"cpu_default": cpu_default, "cuda_default": cuda_default,
           "glass_stress": glass_stress, "random4096": random4096}


def builtin(name: str, width: int, height: int):
    """A built-in scene by name; "random<N>" (N spheres of config 5's generator over its floor plane) for any N, e.g. the
    150- and 1 000-primitive scenes the launch heuristics are swept on ("random149", "random999")."""
    if name not in BUILTIN and name.startswith("random") and name[6:].isdigit():
        return random4096(width, height, count=int(name[6:]))
    return BUILTIN[name](width, height)
