"""Built-in scenes: the Python restatement (path_tracer_amd/scenes.py) and the C++ host mirror
(pt_builtin_scene, written with include/pt_host.hpp) must agree bit for bit, and carry the
literals of the reference's two main()s."""
import numpy as np
import pytest

from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import builtin_scene_from_library


@pytest.mark.parametrize("name,size", [("cpu_default", (1280, 720)), ("cuda_default", (1280, 720)),
                                        ("cuda_default", (1920, 1080)), ("glass_stress", (1280, 720)),
                                        ("random4096", (3840, 2160))])
def test_python_and_cpp_scenes_agree(name, size):
    s_py, c_py = scenes.builtin(name, *size)
    s_cc, c_cc = builtin_scene_from_library(name, *size)
    for n in s_py.arrays:
        assert np.array_equal(s_py.arrays[n], s_cc.arrays[n]), n
    assert bytes(c_py) == bytes(c_cc)


def test_cuda_default_literals():
    """cuda_pathtracer/src/main.cu:99-200 and MaterialFactory.cuh:64-71."""
    s, c = scenes.cuda_default(1920, 1080)
    a = s.arrays
    assert (s.num_spheres, s.num_planes, s.num_triangles, s.num_materials) == (9, 6, 4, 8)
    f = np.float32
    assert a["sph_r"].tolist() == [float(f(0.7)), float(f(0.6)), float(f(0.6)), 0.25, 0.25, 0.25,
                                   float(f(0.3)), float(f(0.15)), float(f(0.15))]
    assert a["sph_obj"].tolist() == list(range(6, 15)) and a["pl_obj"].tolist() == list(range(6))
    assert a["tri_obj"].tolist() == [15, 16, 17, 18]
    assert a["mat_type"].tolist() == [0, 0, 0, 4, 2, 1, 3, 1]
    assert a["sph_mat"].tolist() == [4, 6, 5, 4, 6, 6, 3, 3, 3]
    assert a["pl_mat"].tolist() == [0, 0, 1, 2, 7, 7]          # back and front walls are mirrors (D2)
    assert a["sph_emissive"].tolist() == [0] * 6 + [1, 1, 1]
    assert a["mat_em_r"][3] == 15.0 and a["sph_em_b"].tolist()[6:] == [15.0, 10.0, 0.0]
    assert a["mat_ior"][4] == 1.5 and a["mat_roughness"][6] == float(f(0.2)) and a["mat_metallic"][6] == float(f(0.8))
    # tetrahedron: top vertex (0, 0.6, 0) + (-0.6, 0, -1.0) in float
    assert (a["tri_v0x"][0], a["tri_v0y"][0], a["tri_v0z"][0]) == (float(f(-0.6)), float(f(0.6)), -1.0)
    assert list(c.origin) == [0.0, 0.0, float(f(2.0) * f(0.8))] and list(c.up) == [0.0, -1.0, 0.0]
    assert c.fov_deg == 90.0 and c.model == abi.CAMERA_CUDA and (c.width, c.height) == (1920, 1080)


def test_cpu_default_literals():
    """cpu_pathtracer/src/main.cpp:63-131."""
    s, c = scenes.cpu_default()
    a = s.arrays
    assert (s.num_spheres, s.num_planes, s.num_triangles) == (3, 5, 0)
    assert a["pl_obj"].tolist() == [0, 1, 2, 3, 4] and a["sph_obj"].tolist() == [5, 6, 7]
    assert a["sph_cy"].tolist() == [-2.0 + 0.8, -2.0 + 0.4, 2.0 - 0.1] and a["sph_r"].tolist() == [0.8, 0.4, 0.2]
    assert a["sph_emissive"].tolist() == [0, 0, 1] and a["sph_em_r"][2] == 100.0
    assert c.aspect_ratio == 16.0 / 9.0 and c.focus_dist == 10.0 and c.model == abi.CAMERA_CPU


def test_random4096_shape():
    s, _ = scenes.random4096()
    a = s.arrays
    assert s.num_spheres == 4096 and s.num_planes == 1 and s.num_materials == 4097
    assert int(a["sph_emissive"].sum()) == 64
    assert np.all(a["sph_r"][a["sph_emissive"] == 1] == float(np.float32(0.4)))
    assert a["sph_cx"].min() >= -24 and a["sph_cx"].max() <= 24 and a["sph_cz"].min() >= -48 and a["sph_cz"].max() <= 0
    types = a["mat_type"][a["sph_mat"]]
    assert (types == abi.MAT_EMISSIVE).sum() == 64 and (types == abi.MAT_DIELECTRIC).sum() > 500


def test_scene_arrays_validation():
    with pytest.raises(ValueError):
        abi.SceneArrays(sph_cx=[0.0], sph_cy=[0.0])  # ragged
    empty = abi.SceneArrays()
    assert empty.num_spheres == 0 and empty.struct().num_materials == 0


# ------------------------------------------------------------- composite objects (§8f.4) ---
def test_rectangle_and_tetrahedron_flatten_to_triangles_sharing_the_object_id():
    from path_tracer_amd.scenes import Lambertian, SceneBuilder
    b = SceneBuilder()
    m = b.material(Lambertian(0.5))
    b.addObject(b.Sphere((0, 0, 0), 1.0, m))
    b.addObject(b.Rectangle((-1, -1, -2), (2, 0, 0), (0, 2, 0), m))
    b.addObject(b.Tetrahedron([(0, 1, 0), (-1, -1, 1), (1, -1, 1), (0, -1, -1)], m))
    b.addObject(b.regular_tetrahedron(0.5, m))
    s = b.flatten()
    assert (s.num_spheres, s.num_planes, s.num_triangles) == (1, 0, 2 + 4 + 4)
    assert s.arrays["tri_obj"].tolist() == [1, 1, 2, 2, 2, 2, 3, 3, 3, 3]
    v = np.stack([np.stack([s.arrays[f"tri_v{k}{c}"] for c in "xyz"], -1) for k in range(3)], 1)  # [tri, vertex, xyz]
    n = np.cross(v[:2, 1] - v[:2, 0], v[:2, 2] - v[:2, 0])
    assert np.allclose(n / np.linalg.norm(n, axis=1, keepdims=True), [[0, 0, 1], [0, 0, 1]])  # cross(u, v), Polygon.cuh:126-128
    assert {tuple(p) for p in v[:2].reshape(-1, 3)} == {(-1, -1, -2), (1, -1, -2), (1, 1, -2), (-1, 1, -2)}
    # face order of Tetrahedron.h:16-21
    p = [(0, 1, 0), (-1, -1, 1), (1, -1, 1), (0, -1, -1)]
    want = [[p[i], p[j], p[k]] for i, j, k in ((0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3))]
    assert np.array_equal(v[2:6], np.array(want, dtype=np.float64))
    with pytest.raises(ValueError):
        b.Rectangle((0, 0, 0), (1, 0, 0), (1, 1, 0), m)   # not perpendicular: the reference's test is not a parallelogram test


def test_cpp_host_mirror_flattens_composites_like_the_python_builder(tmp_path):
    """Rectangle_t / Tetrahedron_t of include/pt_host.hpp (host only; compiled with g++ here)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "poly.cpp"
    src.write_text(r'''
#include "pt_host.hpp"
#include <cstdio>
using namespace pt;
int main() {
    SceneManager sm;
    Material_t* m = new Lambertian_t(Color_t(0.5, 0.5, 0.5));
    sm.addObject(new Sphere_t(Point3f_t(0, 0, 0), 1.0, m));
    sm.addObject(new Rectangle_t(Point3f_t(-1, -1, -2), Vec3f_t(2, 0, 0), Vec3f_t(0, 2, 0), m));
    const Point3f_t v[4] = {{0, 1, 0}, {-1, -1, 1}, {1, -1, 1}, {0, -1, -1}};
    sm.addObject(new Tetrahedron_t(v, m));
    sm.addObject(Tetrahedron_t::createRegular(0.5, m));
    FlatScene f = sm.flattenSoA();
    for (size_t i = 0; i < f.tri_obj.size(); ++i) {
        std::printf("%u", f.tri_obj[i]);
        for (int k = 0; k < 9; ++k) std::printf(" %.17g", f.tri_v[k][i]);
        std::printf("\n");
    }
    bool threw = false;
    try { Rectangle_t bad(Point3f_t(0, 0, 0), Vec3f_t(1, 0, 0), Vec3f_t(1, 1, 0), m); } catch (const Error&) { threw = true; }
    std::printf("threw %d\n", threw ? 1 : 0);
    return 0;
}
''')
    exe = tmp_path / "poly"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert lines[-1] == "threw 1"
    got = np.array([[float(x) for x in l.split()] for l in lines[:-1]])
    from path_tracer_amd.scenes import Lambertian, SceneBuilder
    b = SceneBuilder()
    m = b.material(Lambertian(0.5))
    b.addObject(b.Sphere((0, 0, 0), 1.0, m))
    b.addObject(b.Rectangle((-1, -1, -2), (2, 0, 0), (0, 2, 0), m))
    b.addObject(b.Tetrahedron([(0, 1, 0), (-1, -1, 1), (1, -1, 1), (0, -1, -1)], m))
    b.addObject(b.regular_tetrahedron(0.5, m))
    s = b.flatten()
    want = np.column_stack([s.arrays["tri_obj"]] + [s.arrays[f"tri_v{k}{c}"] for k in range(3) for c in "xyz"])
    assert got.shape == want.shape == (10, 10)
    assert np.array_equal(got, want)
