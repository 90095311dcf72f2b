"""The reference's own host code against include/pt_host.hpp.

When the reference tree is present (this container; never the GPU box), the scene-building block,
the launch-shape declarations and the launchRenderKernel call of cuda_pathtracer/src/main.cu are
read AT RUN TIME, pasted unchanged between a prelude and an epilogue written here, compiled with
g++ against pt_host.hpp and linked with libpt_amd.so.  The program then checks that the scene those
lines build is, array for array, the library's built-in "cuda_default", and that the reference's
call `launchRenderKernel(d_output, w, h, frame_count, grid, block)` binds to the C ABI's by-value
PtDim3 signature.  Nothing of the reference is committed or shipped: the pasted text lives in
pytest's tmp_path only.  No GPU is needed (no context is created; the launch reports "no current
context" through pt_last_error()).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN_CU = "/root/reference/cuda_pathtracer/src/main.cu"

PRELUDE = r"""
#include "pt_host.hpp"
#include <cstdio>
#include <cstring>
using namespace pt;
struct FakeWindow { uint32_t getWidth() const { return 1920; } uint32_t getHeight() const { return 1080; } };
template <class T> static bool same(const std::vector<T>& a, const T* b, uint32_t n, const char* what) {
    if (a.size() != n || (n && std::memcmp(a.data(), b, n * sizeof(T)) != 0)) { std::printf("MISMATCH %s\n", what); return false; }
    return true;
}
int main() {
    FakeWindow window;
    float4* d_output = reinterpret_cast<float4*>(0x1000);  // never dereferenced: no context is current
    uint32_t frame_count = 0;
"""

EPILOGUE = r"""
    // ---- checks (this file's own code again) ----
    // the launch above ran without a current context: void return, the message is the error channel
    const std::string launch_error = pt_last_error();
    FlatScene f = scene_manager.flattenSoA();
    PtSceneSoA b; PtCamera bc;
    if (pt_builtin_scene("cuda_default", 1920, 1080, &b, &bc) != PT_OK) { std::printf("no builtin\n"); return 2; }
    bool ok = true;
#define SAME(field) ok = same(f.field, b.field, n, #field) && ok
    uint32_t n = b.num_spheres;
    SAME(sph_cx); SAME(sph_cy); SAME(sph_cz); SAME(sph_r); SAME(sph_obj); SAME(sph_emissive);
    SAME(sph_em_r); SAME(sph_em_g); SAME(sph_em_b);
    n = b.num_planes;
    SAME(pl_px); SAME(pl_py); SAME(pl_pz); SAME(pl_nx); SAME(pl_ny); SAME(pl_nz); SAME(pl_obj);
    n = b.num_triangles;
    ok = same(f.tri_v[0], b.tri_v0x, n, "tri_v0x") && same(f.tri_v[1], b.tri_v0y, n, "tri_v0y") && same(f.tri_v[2], b.tri_v0z, n, "tri_v0z") && ok;
    ok = same(f.tri_v[3], b.tri_v1x, n, "tri_v1x") && same(f.tri_v[4], b.tri_v1y, n, "tri_v1y") && same(f.tri_v[5], b.tri_v1z, n, "tri_v1z") && ok;
    ok = same(f.tri_v[6], b.tri_v2x, n, "tri_v2x") && same(f.tri_v[7], b.tri_v2y, n, "tri_v2y") && same(f.tri_v[8], b.tri_v2z, n, "tri_v2z") && ok;
    SAME(tri_obj);
    // material indices are assigned in order of first use here and in palette order by the built-in: compare
    // the material each primitive is bound to, field by field
    auto same_material = [&](uint32_t mf, uint32_t mb) {
        return f.mat_type[mf] == b.mat_type[mb] && f.mat_albedo_r[mf] == b.mat_albedo_r[mb] && f.mat_albedo_g[mf] == b.mat_albedo_g[mb] &&
               f.mat_albedo_b[mf] == b.mat_albedo_b[mb] && f.mat_roughness[mf] == b.mat_roughness[mb] && f.mat_metallic[mf] == b.mat_metallic[mb] &&
               f.mat_ior[mf] == b.mat_ior[mb] && f.mat_em_r[mf] == b.mat_em_r[mb] && f.mat_em_g[mf] == b.mat_em_g[mb] && f.mat_em_b[mf] == b.mat_em_b[mb];
    };
    if (f.mat_type.size() != b.num_materials) { std::printf("MISMATCH material count\n"); ok = false; }
    for (uint32_t i = 0; ok && i < b.num_spheres; ++i) ok = same_material(f.sph_mat[i], b.sph_mat[i]);
    for (uint32_t i = 0; ok && i < b.num_planes; ++i) ok = same_material(f.pl_mat[i], b.pl_mat[i]);
    for (uint32_t i = 0; ok && i < b.num_triangles; ++i) ok = same_material(f.tri_mat[i], b.tri_mat[i]);
    PtCamera c = camera.toPt(PT_CAMERA_CUDA);
    for (int k = 0; k < 3; ++k) ok = ok && c.origin[k] == bc.origin[k] && c.target[k] == bc.target[k] && c.up[k] == bc.up[k];
    ok = ok && c.fov_deg == bc.fov_deg && c.width == bc.width && c.height == bc.height;
    if (!ok) { std::printf("camera or scene differs from the built-in cuda_default\n"); return 3; }
    if (grid.x != 120 || grid.y != 68 || grid.z != 1 || block.x != 16 || block.y != 16 || block.z != 1) { std::printf("dim3\n"); return 4; }
    static_assert(sizeof(dim3) == sizeof(PtDim3) && sizeof(PtDim3) == 12, "dim3 layout");
    if (launch_error.find("no current context") == std::string::npos) { std::printf("launch error channel: '%s'\n", launch_error.c_str()); return 5; }
    if (pt_builtin_scene("cuda_default", 64, 36, &b, &bc) != PT_OK || *pt_last_error()) { std::printf("message not cleared\n"); return 6; }
    MaterialFactory::cleanup(d_white_diffuse, d_red_diffuse, d_green_diffuse, d_light, d_glass, d_metal, d_glossy, d_mirror);
    std::printf("reference call sites OK: %u spheres, %u planes, %u triangles\n", b.num_spheres, b.num_planes, b.num_triangles);
    return 0;
}
"""


def _lines(text, first, last):
    """Lines first..last (1-based, inclusive) of the reference file, as they are."""
    return "\n".join(text.splitlines()[first - 1:last]) + "\n"


@pytest.mark.skipif(not os.path.exists(MAIN_CU), reason="the reference tree is not on this machine")
def test_reference_scene_block_and_launch_compile_against_pt_host(tmp_path):
    text = open(MAIN_CU).read()
    body = (_lines(text, 63, 84)        # room dimensions, Camera_t::Settings, Camera_t
            + _lines(text, 86, 200)     # materials, SceneManager, walls, spheres, lights, tetrahedron
            + _lines(text, 207, 212)    # dim3 block(16, 16); dim3 grid(...)
            + _lines(text, 287, 288))   # launchRenderKernel(d_output, w, h, frame_count, grid, block);
    assert "launchRenderKernel(d_output" in body and "dim3 block(16, 16)" in body and "MaterialFactory::createMaterialsOnDevice" in body
    src = tmp_path / "callsite.cpp"
    src.write_text(PRELUDE + body + EPILOGUE)
    exe = tmp_path / "callsite"
    libdir = os.path.join(ROOT, "path_tracer_amd", "lib")
    cc = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                         "-L", libdir, "-lpt_amd", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "reference call sites OK: 9 spheres, 6 planes, 4 triangles" in run.stdout
