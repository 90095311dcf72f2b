// pt_builtin_scenes.cpp — the built-in scenes, written with the host mirror of the reference's
// Scene / Camera / Material interfaces (include/pt_host.hpp).  Host only; no GPU call.
//   cpu_default  : cpu_pathtracer/src/main.cpp:63-131
//   cuda_default : cuda_pathtracer/src/main.cu:64-200, palette MaterialFactory.cuh:64-71
//   glass_stress, random4096 : SURVEY.md §8(d) configs 4 and 5
// path_tracer_amd/scenes.py restates the same literals independently; tests/test_scenes.py
// requires the two to agree bit for bit.
#include "../../include/pt_host.hpp"
#include "../csrc/pt_error.h"

#include <cstring>
#include <map>
#include <string>

using namespace pt;

namespace {

struct Palette { Material_t *white, *red, *green, *light, *glass, *metal, *glossy, *mirror; };

Palette make_palette(SceneManager& sm, std::vector<std::unique_ptr<Material_t>>& keep) {
    Palette p{};
    MaterialFactory::createMaterialsOnDevice(&p.white, &p.red, &p.green, &p.light, &p.glass, &p.metal, &p.glossy, &p.mirror);
    for (Material_t* m : {p.white, p.red, p.green, p.light, p.glass, p.metal, p.glossy, p.mirror}) {
        keep.emplace_back(m);
        sm.addMaterial(m);
    }
    return p;
}

void cornell_walls(SceneManager& sm, const Palette& m) {  // main.cu:102-121
    sm.addObject(new Plane_t(Point3f_t(0.0f, -2.0f, 0.0f), Vec3f_t(0.0f, 1.0f, 0.0f), m.white));
    sm.addObject(new Plane_t(Point3f_t(0.0f, 2.0f, 0.0f), Vec3f_t(0.0f, -1.0f, 0.0f), m.white));
    sm.addObject(new Plane_t(Point3f_t(-2.0f, 0.0f, 0.0f), Vec3f_t(1.0f, 0.0f, 0.0f), m.red));
    sm.addObject(new Plane_t(Point3f_t(2.0f, 0.0f, 0.0f), Vec3f_t(-1.0f, 0.0f, 0.0f), m.green));
    sm.addObject(new Plane_t(Point3f_t(0.0f, 0.0f, -2.0f), Vec3f_t(0.0f, 0.0f, 1.0f), m.mirror));
    sm.addObject(new Plane_t(Point3f_t(0.0f, 0.0f, 2.0f), Vec3f_t(0.0f, 0.0f, -1.0f), m.mirror));
}

void cornell_lights(SceneManager& sm, const Palette& m) {  // main.cu:154-168
    auto main_light = new Sphere_t(Point3f_t(0.0f, 1.8f, 0.0f), 0.3f, m.light);
    main_light->makeEmissive(Color_t(1.0f), 15.0f);
    sm.addObject(main_light);
    auto blue_light = new Sphere_t(Point3f_t(1.8f, 0.0f, -1.8f), 0.15f, m.light);
    blue_light->makeEmissive(Color_t(0.2f, 0.2f, 1.0f), 10.0f);
    sm.addObject(blue_light);
    auto orange_light = new Sphere_t(Point3f_t(-1.8f, 0.0f, -1.8f), 0.15f, m.light);
    orange_light->makeEmissive(Color_t(1.0f, 0.5f, 0.0f), 10.0f);
    sm.addObject(orange_light);
}

Camera_t cuda_camera(uint32_t w, uint32_t h) {  // main.cu:68-84
    const float half_size = 4.0f / 2.0f;
    Camera_t::Settings s{90.0f, (float)w / static_cast<float>(h), 0.0f, 10.0f, w, h, true};
    return Camera_t(Point3f_t(0, 0, half_size * 0.8f), Point3f_t(0, 0, -1), Vec3f_t(0, -1, 0), s);
}

struct Built { FlatScene flat; PtCamera cam; };

Built build_cpu_default(uint32_t w, uint32_t h) {  // main.cpp:63-131
    SceneManager scene;
    const double half_size = 4.0 / 2.0;
    scene.addImplicitObject(new Plane(Point3d(0, 0, -half_size), Vec3d(0, 0, 1), ColorDBL(0.9, 0.9, 0.9)));
    scene.addImplicitObject(new Plane(Point3d(-half_size, 0, 0), Vec3d(1, 0, 0), ColorDBL(0.9, 0.2, 0.2)));
    scene.addImplicitObject(new Plane(Point3d(half_size, 0, 0), Vec3d(-1, 0, 0), ColorDBL(0.2, 0.9, 0.2)));
    scene.addImplicitObject(new Plane(Point3d(0, -half_size, 0), Vec3d(0, 1, 0), ColorDBL(0.9, 0.9, 0.9)));
    scene.addImplicitObject(new Plane(Point3d(0, half_size, 0), Vec3d(0, -1, 0), ColorDBL(0.9, 0.9, 0.9)));
    scene.addImplicitObject(new Sphere(Point3d(-1.0, -half_size + 0.8, -half_size + 1.0), 0.8, ColorDBL(0.95, 0.95, 0.95)));
    scene.addImplicitObject(new Sphere(Point3d(1.0, -half_size + 0.4, -half_size + 1.0), 0.4, ColorDBL(0.7, 0.3, 0.3)));
    auto light = new Sphere(Point3d(0, half_size - 0.1, 0), 0.2, ColorDBL(1.0, 1.0, 1.0));
    light->makeEmissive(ColorDBL(1.0, 1.0, 1.0), 100.0);
    scene.addImplicitObject(light);
    Camera::Settings cs;  // aspect_ratio 16/9 and focus_dist 10 stay at their defaults, main.cpp:98-102
    cs.width = w; cs.height = h; cs.fov = 90.0; cs.use_dof = false;
    Camera camera(Point3d(0, 0, half_size * 0.8), Point3d(0, 0, -1), Vec3d(0, 1, 0), cs);
    return Built{scene.flattenSoA(), camera.toPt(PT_CAMERA_CPU)};
}

Built build_cuda_default(uint32_t w, uint32_t h) {  // main.cu:64-200
    SceneManager sm;
    std::vector<std::unique_ptr<Material_t>> keep;
    Palette m = make_palette(sm, keep);
    cornell_walls(sm, m);
    sm.addObject(new Sphere_t(Point3f_t(0.0f, -1.0f, 0.0f), 0.7f, m.glass));
    sm.addObject(new Sphere_t(Point3f_t(-1.2f, -1.2f, -0.5f), 0.6f, m.glossy));
    sm.addObject(new Sphere_t(Point3f_t(1.2f, -1.2f, -0.5f), 0.6f, m.metal));
    const float small_radius = 0.25f;
    sm.addObject(new Sphere_t(Point3f_t(0.0f, -1.4f, -1.2f), small_radius, m.glass));
    sm.addObject(new Sphere_t(Point3f_t(-0.6f, -1.4f, -1.0f), small_radius, m.glossy));
    sm.addObject(new Sphere_t(Point3f_t(0.6f, -1.4f, -1.0f), small_radius, m.glossy));
    cornell_lights(sm, m);
    // tetrahedron, main.cu:170-200; the translation is float arithmetic
    const float tr = 0.6f;
    float tv[4][3] = {{0, tr, 0}, {-tr, -tr, tr}, {tr, -tr, tr}, {0, -tr, -tr}};
    const float tp[3] = {-0.6f, 0.0f, -1.0f};
    for (auto& v : tv) for (int k = 0; k < 3; ++k) v[k] = v[k] + tp[k];
    const uint8_t faces[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};
    for (auto& f : faces) {
        auto P = [&](int i) { return Point3f_t(tv[i][0], tv[i][1], tv[i][2]); };
        sm.addObject(new Triangle_t(P(f[0]), P(f[1]), P(f[2]), m.glass));
    }
    return Built{sm.flattenSoA(), cuda_camera(w, h).toPt(PT_CAMERA_CUDA)};
}

Built build_glass_stress(uint32_t w, uint32_t h) {
    SceneManager sm;
    std::vector<std::unique_ptr<Material_t>> keep;
    Palette m = make_palette(sm, keep);
    cornell_walls(sm, m);
    const float radii[4] = {1.0f, 0.8f, 0.6f, 0.4f}, iors[4] = {1.5f, 1.33f, 1.5f, 2.4f};
    for (int i = 0; i < 4; ++i) {
        keep.emplace_back(new Dielectric_t(iors[i]));
        sm.addObject(new Sphere_t(Point3f_t(0.0f, -0.8f, -0.3f), radii[i], keep.back().get()));
    }
    sm.addObject(new Sphere_t(Point3f_t(1.3f, -1.65f, -0.8f), 0.35f, m.glass));
    sm.addObject(new Sphere_t(Point3f_t(-1.3f, -1.65f, -0.8f), 0.35f, m.glass));
    cornell_lights(sm, m);
    return Built{sm.flattenSoA(), cuda_camera(w, h).toPt(PT_CAMERA_CUDA)};
}

uint32_t hash32(uint32_t x) {  // the counter-RNG finaliser, DESIGN.md §RNG
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
double u01(uint32_t i, uint32_t k) { return (double)(hash32(hash32(0xC0FFEEu + i) + k) >> 8) * (1.0 / 16777216.0); }

Built build_random4096(uint32_t w, uint32_t h) {
    SceneManager sm;
    std::vector<std::unique_ptr<Material_t>> keep;
    keep.emplace_back(new Lambertian_t(Color_t(0.73f)));
    sm.addObject(new Plane_t(Point3f_t(0.0f, -2.0f, 0.0f), Vec3f_t(0.0f, 1.0f, 0.0f), keep.back().get()));
    auto F = [](double v) { return (double)(float)v; };
    for (uint32_t i = 0; i < 4096; ++i) {
        double u[7];
        for (uint32_t k = 0; k < 7; ++k) u[k] = u01(i, k);
        Point3f_t c(F(-24.0 + 48.0 * u[0]), F(-1.5 + 7.5 * u[1]), F(-48.0 + 48.0 * u[2]));
        const double r = F(0.15 + 0.45 * u[3]);
        Color_t alb(F(0.2 + 0.7 * u[4]), F(0.2 + 0.7 * u[5]), F(0.2 + 0.7 * u[6]));
        if (i % 64 == 0) {
            Color_t col(F(0.2 + 0.8 * u[4]), F(0.2 + 0.8 * u[5]), F(0.2 + 0.8 * u[6]));
            keep.emplace_back(new Emissive_t(col, 15.0));
            auto s = new Sphere_t(c, F(0.4), keep.back().get());
            s->makeEmissive(col, 15.0);
            sm.addObject(s);
            continue;
        }
        const uint32_t k = i % 20;
        if (k < 10) keep.emplace_back(new Lambertian_t(alb));
        else if (k < 14) keep.emplace_back(new Metal_t(alb, 0.0));
        else if (k < 17) keep.emplace_back(new Glossy_t(alb, F(0.2), F(0.8)));
        else keep.emplace_back(new Dielectric_t(1.5));
        sm.addObject(new Sphere_t(c, r, keep.back().get()));
    }
    Camera_t::Settings s{60.0, (double)((float)w / (float)h), 0.0, 10.0, w, h, false};
    Camera_t cam(Point3f_t(0, 2, 6), Point3f_t(0, 1, -20), Vec3f_t(0, 1, 0), s);
    return Built{sm.flattenSoA(), cam.toPt(PT_CAMERA_CUDA)};
}

thread_local std::map<std::string, Built> g_built;

}  // namespace

extern "C" int pt_builtin_scene(const char* name, uint32_t width, uint32_t height, PtSceneSoA* scene, PtCamera* camera) {
    pt_clear_error();
    if (!name || !scene || !camera || width == 0 || height == 0) return pt_fail(PT_ERR_INVALID, "pt_builtin_scene: null argument or zero frame size");
    const std::string n(name);
    Built b;
    if (n == "cpu_default") b = build_cpu_default(width, height);
    else if (n == "cuda_default") b = build_cuda_default(width, height);
    else if (n == "glass_stress") b = build_glass_stress(width, height);
    else if (n == "random4096") b = build_random4096(width, height);
    else return pt_fail(PT_ERR_INVALID, "pt_builtin_scene: no scene named '%s' (cpu_default, cuda_default, glass_stress, random4096)", name);
    Built& slot = g_built[n];
    slot = std::move(b);
    *scene = slot.flat.view();
    *camera = slot.cam;
    return PT_OK;
}
