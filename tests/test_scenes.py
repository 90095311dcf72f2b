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
