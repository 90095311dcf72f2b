// pt_host.hpp — C++ host-side mirror of the reference's Scene / Camera / Material interfaces,
// sitting directly above the C ABI of pt.h.
//
// The names, constructor shapes and argument meaning follow the reference so that its app
// shells port by changing includes:
//   Sphere_t(center, radius, material) / Plane_t(point, normal, material) /
//   Triangle_t(v0, v1, v2, material), obj->makeEmissive(colour, strength)
//       — cuda_pathtracer/include/ImplicitObject.cuh:31-35,71,162; Polygon.cuh:48
//   Sphere(center, radius, colour) / Plane(point, normal, colour)   (cpu_pathtracer shape)
//       — cpu_pathtracer/include/ImplicitObject.h:50,140
//   Lambertian_t(albedo) / Metal_t(albedo, roughness) / Dielectric_t(ior) /
//   Glossy_t(albedo, roughness, metallic = 1) / Emissive_t(colour, strength)
//       — cuda_pathtracer/include/Material.cuh:41,92,147,185,241
//   MaterialFactory::createMaterialsOnDevice(&white, &red, &green, &light, &glass, &metal,
//       &glossy, &mirror) — MaterialFactory.cuh:41-75 (here: host descriptors, same palette)
//   Camera_t(origin, target, up, Settings{fov, aspect_ratio, aperture, focus_dist, width,
//       height, use_dof}), lookAt, setPosition, move*, getters — Camera.cuh:12-20,39,53-71
//   SceneManager::addObject / uploadToGPU — Scene.cuh:266-281
// What differs: materials are plain host descriptors (no device vtables), the scene is
// flattened to the SoA of pt.h, and every device interaction goes through the C ABI.
// Errors from the ABI surface as pt::Error (the reference throws const char*, Error.cuh:6-16).
#pragma once

#include "pt.h"

#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) {
    if (rc != PT_OK) throw Error(rc, pt_last_error());
}

// ---- value types (Vec3.cuh / Vec3.h, Color.cuh / Color.h) --------------------------------
struct Vec3_t {
    double x{0}, y{0}, z{0};
    Vec3_t() = default;
    Vec3_t(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    double length() const { return std::sqrt(x * x + y * y + z * z); }
    Vec3_t normalized() const {
        double l = length();
        return l > 0 ? Vec3_t(x * (1 / l), y * (1 / l), z * (1 / l)) : *this;
    }
};
using Vec3f_t = Vec3_t;
using Point3f_t = Vec3_t;
using Vec3d = Vec3_t;
using Point3d = Vec3_t;
inline Vec3_t operator+(const Vec3_t& a, const Vec3_t& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3_t operator-(const Vec3_t& a, const Vec3_t& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3_t operator*(const Vec3_t& a, double t) { return {a.x * t, a.y * t, a.z * t}; }
inline Vec3_t& operator+=(Vec3_t& a, const Vec3_t& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }  // Vec3.cuh:71-76 (main.cu:181)
inline Vec3_t& operator-=(Vec3_t& a, const Vec3_t& b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
inline double dot(const Vec3_t& a, const Vec3_t& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3_t cross(const Vec3_t& u, const Vec3_t& v) {
    return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}

struct Color_t {
    double r{0}, g{0}, b{0};
    Color_t() = default;
    Color_t(double t) : r(t), g(t), b(t) {}  // Color.cuh:13
    Color_t(double r_, double g_, double b_) : r(r_), g(g_), b(b_) {}
};
using ColorDBL = Color_t;

// ---- launch-shape and pixel types of the reference's call sites ------------------------------
// main.cu:55,208-212,287-288 declares `float4* d_output`, `dim3 block(16, 16)`, `dim3 grid(...)` and passes both
// dim3 by value to launchRenderKernel.  A host built with hipcc has HIP's own types (same layout; the overload
// below forwards them); a plain g++ host gets them here.
#ifdef HIP_INCLUDE_HIP_HIP_RUNTIME_API_H
inline void launchRenderKernel(void* output, uint32_t width, uint32_t height, uint32_t frame_count, const ::dim3& grid, const ::dim3& block) {
    ::launchRenderKernel(output, width, height, frame_count, PtDim3{grid.x, grid.y, grid.z}, PtDim3{block.x, block.y, block.z});
}
#else
struct dim3 : PtDim3 {
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : PtDim3{x_, y_, z_} {}
};
struct float4 { float x, y, z, w; };
#endif

// ---- materials ---------------------------------------------------------------------------
class Material_t {
public:
    virtual ~Material_t() = default;
    virtual uint32_t type() const = 0;
    Color_t albedo{};
    double roughness{0}, metallic{0}, ior{1};
    Color_t emission{};  // colour * strength
};
class Lambertian_t : public Material_t {
public:
    explicit Lambertian_t(const Color_t& a) { albedo = a; }
    uint32_t type() const override { return PT_MAT_LAMBERTIAN; }
};
class Metal_t : public Material_t {
public:
    Metal_t(const Color_t& a, double rough) { albedo = a; roughness = rough; }
    uint32_t type() const override { return PT_MAT_METAL; }
};
class Dielectric_t : public Material_t {
public:
    explicit Dielectric_t(double i) { ior = i; }
    uint32_t type() const override { return PT_MAT_DIELECTRIC; }
};
class Glossy_t : public Material_t {
public:
    Glossy_t(const Color_t& a, double rough, double metal = 1.0) { albedo = a; roughness = rough; metallic = metal; }
    uint32_t type() const override { return PT_MAT_GLOSSY; }
};
class Emissive_t : public Material_t {
public:
    Emissive_t(const Color_t& c, double strength) { emission = Color_t(c.r * strength, c.g * strength, c.b * strength); }
    uint32_t type() const override { return PT_MAT_EMISSIVE; }
};

// The fixed 8-material palette of MaterialFactory.cuh:64-71 (float literals).
struct MaterialFactory {
    static void createMaterialsOnDevice(Material_t** white, Material_t** red, Material_t** green, Material_t** light,
                                        Material_t** glass, Material_t** metal, Material_t** glossy, Material_t** mirror) {
        *white = new Lambertian_t(Color_t(0.73f));
        *red = new Lambertian_t(Color_t(0.65f, 0.05f, 0.05f));
        *green = new Lambertian_t(Color_t(0.12f, 0.45f, 0.15f));
        *light = new Emissive_t(Color_t(1.0f), 15.0f);
        *glass = new Dielectric_t(1.5f);
        *metal = new Metal_t(Color_t(0.95f), 0.0f);
        *glossy = new Glossy_t(Color_t(0.7f, 0.7f, 0.7f), 0.2f, 0.8f);
        *mirror = new Metal_t(Color_t(0.98f, 0.98f, 0.98f), 0.0f);
    }
    static void cleanup(Material_t* a, Material_t* b, Material_t* c, Material_t* d, Material_t* e, Material_t* f,
                        Material_t* g, Material_t* h) {
        delete a; delete b; delete c; delete d; delete e; delete f; delete g; delete h;
    }
};

// ---- objects -----------------------------------------------------------------------------
class ImplicitObject_t {
protected:
    Material_t* material_;
    std::unique_ptr<Material_t> owned_;  // colour-constructed objects (cpu_pathtracer shape) own a Lambertian
    bool isEmissive_{false};
    Color_t emissionColor_{};
    double emissionStrength_{0};

public:
    explicit ImplicitObject_t(Material_t* m) : material_(m) {}
    explicit ImplicitObject_t(const Color_t& colour) : material_(nullptr), owned_(new Lambertian_t(colour)) { material_ = owned_.get(); }
    virtual ~ImplicitObject_t() = default;
    void makeEmissive(const Color_t& c, double strength) {
        isEmissive_ = true; emissionColor_ = c; emissionStrength_ = strength;
        if (owned_) {  // a colour-constructed light also gets an Emissive material so `path` can render it
            owned_.reset(new Emissive_t(c, strength));
            material_ = owned_.get();
        }
    }
    bool isEmissive() const { return isEmissive_; }
    const Color_t& getEmissionColor() const { return emissionColor_; }
    double getEmissionStrength() const { return emissionStrength_; }
    Material_t* getMaterial() const { return material_; }
    virtual bool isSphere() const { return false; }
    virtual bool isPlane() const { return false; }
    virtual bool isTriangle() const { return false; }
    virtual bool isPolygon() const { return false; }  // Rectangle_t, Tetrahedron_t: a set of triangles
};

class Sphere_t final : public ImplicitObject_t {
    Point3f_t center_; double radius_;
public:
    Sphere_t(const Point3f_t& c, double r, Material_t* m) : ImplicitObject_t(m), center_(c), radius_(r) {}
    Sphere_t(const Point3d& c, double r, const ColorDBL& colour) : ImplicitObject_t(colour), center_(c), radius_(r) {}
    const Point3f_t& getCenter() const { return center_; }
    double getRadius() const { return radius_; }
    bool isSphere() const override { return true; }
};
class Plane_t final : public ImplicitObject_t {
    Point3f_t point_; Vec3f_t normal_;
public:
    Plane_t(const Point3f_t& p, const Vec3f_t& n, Material_t* m) : ImplicitObject_t(m), point_(p), normal_(n) {}
    Plane_t(const Point3d& p, const Vec3d& n, const ColorDBL& colour) : ImplicitObject_t(colour), point_(p), normal_(n) {}
    const Point3f_t& getPoint() const { return point_; }
    const Vec3f_t& getNormal() const { return normal_; }  // normalised by the library at upload
    bool isPlane() const override { return true; }
};
class Triangle_t final : public ImplicitObject_t {
    Point3f_t v_[3];
public:
    Triangle_t(const Point3f_t& v0, const Point3f_t& v1, const Point3f_t& v2, Material_t* m) : ImplicitObject_t(m), v_{v0, v1, v2} {}
    const Point3f_t* getVertices() const { return v_; }
    bool isTriangle() const override { return true; }
};
// Rectangle_t(origin, u, v, ...) (Polygon.cuh:113-168 / Polygon.h:103-153) and the tetrahedron of four
// triangles (Tetrahedron.h:9-35, built inline in main.cu:170-200) are present in the reference's headers
// but never reach its render loops.  Here they are composite objects: flattenSoA() emits their faces
// as triangles that share the object's id, so the kernels need no new primitive kind.  The
// reference's rectangle test projects the hit point on u and v separately (alpha, beta in [0,1]),
// which describes the parallelogram only for perpendicular edges: others are rejected.
class Polygon_t : public ImplicitObject_t {
protected:
    std::vector<Point3f_t> faces_;  // three vertices per triangle
public:
    explicit Polygon_t(Material_t* m) : ImplicitObject_t(m) {}
    explicit Polygon_t(const Color_t& c) : ImplicitObject_t(c) {}
    bool isPolygon() const override { return true; }
    size_t triangleCount() const { return faces_.size() / 3; }
    const Point3f_t* triangle(size_t i) const { return &faces_[3 * i]; }
};
class Rectangle_t final : public Polygon_t {
    void build(const Point3f_t& o, const Vec3f_t& u, const Vec3f_t& v) {
        const double uv = dot(u, v), uu = dot(u, u), vv = dot(v, v);
        if (!(uu > 0) || !(vv > 0) || uv * uv > 1e-12 * uu * vv) throw Error(PT_ERR_INVALID, "Rectangle_t: edges must be non-zero and perpendicular");
        const Point3f_t a = o, b = o + u, c = o + u + v, d = o + v;
        faces_ = {a, b, c, a, c, d};  // both with normal cross(u, v), as computeNormal (Polygon.cuh:126-128)
    }
public:
    Rectangle_t(const Point3f_t& origin, const Vec3f_t& u, const Vec3f_t& v, Material_t* m) : Polygon_t(m) { build(origin, u, v); }
    Rectangle_t(const Point3d& origin, const Vec3d& u, const Vec3d& v, const ColorDBL& colour) : Polygon_t(colour) { build(origin, u, v); }
};
class Tetrahedron_t final : public Polygon_t {
    void build(const Point3f_t (&p)[4]) {
        static const int FACE[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};  // Tetrahedron.h:16-21, main.cu:183-199
        for (auto& f : FACE) for (int k : f) faces_.push_back(p[k]);
    }
public:
    Tetrahedron_t(const Point3f_t (&vertices)[4], Material_t* m) : Polygon_t(m) { build(vertices); }
    Tetrahedron_t(const Point3f_t (&vertices)[4], const ColorDBL& colour) : Polygon_t(colour) { build(vertices); }
    // Tetrahedron::createRegular, Tetrahedron.h:38-50
    static Tetrahedron_t* createRegular(double radius, Material_t* m) {
        const double a = radius * 2.0 / std::sqrt(6.0), h = radius * 3.0 / std::sqrt(6.0);
        const Point3f_t v[4] = {{0, -radius, -h / 3}, {-a / 2, radius / 3, -h / 3}, {a / 2, radius / 3, -h / 3}, {0, 0, 2 * h / 3}};
        return new Tetrahedron_t(v, m);
    }
};
using Sphere = Sphere_t;
using Plane = Plane_t;
using Triangle = Triangle_t;
using Rectangle = Rectangle_t;
using Tetrahedron = Tetrahedron_t;

// ---- camera ------------------------------------------------------------------------------
class Camera_t {
public:
    struct Settings {  // Camera.cuh:12-20 / Camera.h:18-26
        double fov{60.0};
        double aspect_ratio{16.0 / 9.0};
        double aperture{0.0};
        double focus_dist{10.0};
        uint32_t width{1920};
        uint32_t height{1080};
        bool use_dof{false};
    };

private:
    Point3f_t origin_, target_;
    Vec3f_t up_;
    Settings settings_;
    bool up_normalized_{true};
    double pitch_{0}, yaw_{0};

public:
    Camera_t(const Point3f_t& origin, const Point3f_t& target, const Vec3f_t& up, const Settings& s)
        : origin_(origin), target_(target), up_(up), settings_(s) {}
    void lookAt(const Point3f_t& target) { target_ = target; }
    // Camera_t::lookAt normalises up (Camera.cu:101-106); the CPU Camera::lookAt(from,to,up) does not (Camera.h:251-256)
    void lookAt(const Point3f_t& pos, const Point3f_t& target, const Vec3f_t& up, bool normalize_up = true) {
        origin_ = pos; target_ = target; up_ = up; up_normalized_ = normalize_up;
    }
    void setPosition(const Point3f_t& p) { origin_ = p; }
    Vec3f_t w() const { return (origin_ - target_).normalized(); }
    Vec3f_t u() const { return cross(up_normalized_ ? up_.normalized() : up_, w()).normalized(); }
    Vec3f_t v() const { return cross(w(), u()); }
    void moveForward(double d) { origin_ = origin_ - w() * d; }  // Camera.cu:113-116
    void moveRight(double d) { origin_ = origin_ + u() * d; }
    void moveUp(double d) { origin_ = origin_ + v() * d; }
    void rotateYaw(double a) { yaw_ = a; updateRotation(); }
    void rotatePitch(double a) { pitch_ = a; updateRotation(); }
    const Point3f_t& getPosition() const { return origin_; }
    const Point3f_t& getTarget() const { return target_; }
    const Settings& getSettings() const { return settings_; }
    uint32_t getWidth() const { return settings_.width; }
    uint32_t getHeight() const { return settings_.height; }
    Vec3f_t getForward() const { Vec3f_t ww = w(); return {-ww.x, -ww.y, -ww.z}; }
    Vec3f_t getRight() const { return u(); }
    Vec3f_t getUp() const { return v(); }

    PtCamera toPt(uint32_t model) const {
        PtCamera c{};
        c.origin[0] = origin_.x; c.origin[1] = origin_.y; c.origin[2] = origin_.z;
        c.target[0] = target_.x; c.target[1] = target_.y; c.target[2] = target_.z;
        c.up[0] = up_.x; c.up[1] = up_.y; c.up[2] = up_.z;
        c.fov_deg = settings_.fov; c.aspect_ratio = settings_.aspect_ratio; c.aperture = settings_.aperture;
        c.focus_dist = settings_.focus_dist; c.width = settings_.width; c.height = settings_.height;
        c.use_dof = settings_.use_dof ? 1u : 0u; c.model = model; c.normalize_up = up_normalized_ ? 1u : 0u;
        return c;
    }

private:
    void updateRotation() {  // Camera.cu:138-147
        target_ = origin_ + Vec3f_t(std::cos(pitch_) * std::cos(yaw_), std::sin(pitch_), std::cos(pitch_) * std::sin(yaw_));
    }
};
using Camera = Camera_t;

// The CPU app's interactive camera (cpu_pathtracer/src/main.cpp:133-205) without its X11 event pump:
// feed it (mouse_dx, mouse_dy, key mask) per frame; `moved` tells the frame loop to reset the accumulation.
class FlyCamera {
    PtCamera cam_;
    double yaw_{0.0}, pitch_{0.0};

public:
    explicit FlyCamera(const PtCamera& c) : cam_(c) {}
    bool step(double mouse_dx, double mouse_dy, uint32_t keys) {
        const int rc = pt_fly_camera_step(&cam_, &yaw_, &pitch_, mouse_dx, mouse_dy, keys);
        if (rc < 0) throw Error(rc, "pt_fly_camera_step: bad arguments");
        return rc == 1;
    }
    const PtCamera& camera() const { return cam_; }
    double yaw() const { return yaw_; }
    double pitch() const { return pitch_; }
};

// ---- scene -------------------------------------------------------------------------------
// Owning SoA produced by SceneManager::flattenSoA(); view() is what pt_upload_scene takes.
struct FlatScene {
    std::vector<double> sph_cx, sph_cy, sph_cz, sph_r, sph_em_r, sph_em_g, sph_em_b;
    std::vector<uint32_t> sph_mat, sph_obj, sph_emissive;
    std::vector<double> pl_px, pl_py, pl_pz, pl_nx, pl_ny, pl_nz;
    std::vector<uint32_t> pl_mat, pl_obj;
    std::vector<double> tri_v[9];
    std::vector<uint32_t> tri_mat, tri_obj;
    std::vector<uint32_t> mat_type;
    std::vector<double> mat_albedo_r, mat_albedo_g, mat_albedo_b, mat_roughness, mat_metallic, mat_ior, mat_em_r, mat_em_g, mat_em_b;

    PtSceneSoA view() const {
        PtSceneSoA s{};
        s.num_spheres = (uint32_t)sph_cx.size(); s.num_planes = (uint32_t)pl_px.size();
        s.num_triangles = (uint32_t)tri_mat.size(); s.num_materials = (uint32_t)mat_type.size();
        s.sph_cx = sph_cx.data(); s.sph_cy = sph_cy.data(); s.sph_cz = sph_cz.data(); s.sph_r = sph_r.data();
        s.sph_mat = sph_mat.data(); s.sph_obj = sph_obj.data(); s.sph_emissive = sph_emissive.data();
        s.sph_em_r = sph_em_r.data(); s.sph_em_g = sph_em_g.data(); s.sph_em_b = sph_em_b.data();
        s.pl_px = pl_px.data(); s.pl_py = pl_py.data(); s.pl_pz = pl_pz.data();
        s.pl_nx = pl_nx.data(); s.pl_ny = pl_ny.data(); s.pl_nz = pl_nz.data();
        s.pl_mat = pl_mat.data(); s.pl_obj = pl_obj.data();
        s.tri_v0x = tri_v[0].data(); s.tri_v0y = tri_v[1].data(); s.tri_v0z = tri_v[2].data();
        s.tri_v1x = tri_v[3].data(); s.tri_v1y = tri_v[4].data(); s.tri_v1z = tri_v[5].data();
        s.tri_v2x = tri_v[6].data(); s.tri_v2y = tri_v[7].data(); s.tri_v2z = tri_v[8].data();
        s.tri_mat = tri_mat.data(); s.tri_obj = tri_obj.data();
        s.mat_type = mat_type.data();
        s.mat_albedo_r = mat_albedo_r.data(); s.mat_albedo_g = mat_albedo_g.data(); s.mat_albedo_b = mat_albedo_b.data();
        s.mat_roughness = mat_roughness.data(); s.mat_metallic = mat_metallic.data(); s.mat_ior = mat_ior.data();
        s.mat_em_r = mat_em_r.data(); s.mat_em_g = mat_em_g.data(); s.mat_em_b = mat_em_b.data();
        return s;
    }
};

class SceneManager {
    std::vector<std::unique_ptr<ImplicitObject_t>> objects_;  // addObject takes ownership, as Scene.cuh:253-258 deletes them
    std::vector<const Material_t*> materials_;                // registration order = material index

    uint32_t materialIndex(const Material_t* m, std::vector<const Material_t*>& table) const {
        for (size_t i = 0; i < table.size(); ++i) if (table[i] == m) return (uint32_t)i;
        table.push_back(m);
        return (uint32_t)table.size() - 1;
    }

public:
    // Optional: fix a material's index up front (the palette order of MaterialFactory).
    uint32_t addMaterial(const Material_t* m) { return materialIndex(m, materials_); }
    void addObject(ImplicitObject_t* object) { objects_.emplace_back(object); }  // Scene.cuh:266
    void addImplicitObject(ImplicitObject_t* object) { addObject(object); }      // cpu Scene.h:38
    size_t size() const { return objects_.size(); }

    // Replaces the AoS flatten loops of main.cu:214-258.  Object id = insertion index.
    FlatScene flattenSoA() const {
        FlatScene f;
        std::vector<const Material_t*> table = materials_;
        for (size_t idx = 0; idx < objects_.size(); ++idx) {
            const ImplicitObject_t* o = objects_[idx].get();
            const uint32_t mat = materialIndex(o->getMaterial(), table);
            if (o->isSphere()) {
                auto* s = static_cast<const Sphere_t*>(o);
                f.sph_cx.push_back(s->getCenter().x); f.sph_cy.push_back(s->getCenter().y); f.sph_cz.push_back(s->getCenter().z);
                f.sph_r.push_back(s->getRadius()); f.sph_mat.push_back(mat); f.sph_obj.push_back((uint32_t)idx);
                f.sph_emissive.push_back(o->isEmissive() ? 1u : 0u);
                const Color_t& c = o->getEmissionColor(); const double k = o->getEmissionStrength();
                f.sph_em_r.push_back(c.r * k); f.sph_em_g.push_back(c.g * k); f.sph_em_b.push_back(c.b * k);
            } else if (o->isPlane()) {
                auto* p = static_cast<const Plane_t*>(o);
                f.pl_px.push_back(p->getPoint().x); f.pl_py.push_back(p->getPoint().y); f.pl_pz.push_back(p->getPoint().z);
                f.pl_nx.push_back(p->getNormal().x); f.pl_ny.push_back(p->getNormal().y); f.pl_nz.push_back(p->getNormal().z);
                f.pl_mat.push_back(mat); f.pl_obj.push_back((uint32_t)idx);
            } else if (o->isTriangle()) {
                auto* t = static_cast<const Triangle_t*>(o);
                for (int k = 0; k < 3; ++k) {
                    f.tri_v[3 * k + 0].push_back(t->getVertices()[k].x);
                    f.tri_v[3 * k + 1].push_back(t->getVertices()[k].y);
                    f.tri_v[3 * k + 2].push_back(t->getVertices()[k].z);
                }
                f.tri_mat.push_back(mat); f.tri_obj.push_back((uint32_t)idx);
            } else if (o->isPolygon()) {
                auto* g = static_cast<const Polygon_t*>(o);
                for (size_t i = 0; i < g->triangleCount(); ++i) {
                    for (int k = 0; k < 3; ++k) {
                        f.tri_v[3 * k + 0].push_back(g->triangle(i)[k].x);
                        f.tri_v[3 * k + 1].push_back(g->triangle(i)[k].y);
                        f.tri_v[3 * k + 2].push_back(g->triangle(i)[k].z);
                    }
                    f.tri_mat.push_back(mat); f.tri_obj.push_back((uint32_t)idx);
                }
            }
        }
        for (const Material_t* m : table) {
            f.mat_type.push_back(m->type());
            f.mat_albedo_r.push_back(m->albedo.r); f.mat_albedo_g.push_back(m->albedo.g); f.mat_albedo_b.push_back(m->albedo.b);
            f.mat_roughness.push_back(m->roughness); f.mat_metallic.push_back(m->metallic); f.mat_ior.push_back(m->ior);
            f.mat_em_r.push_back(m->emission.r); f.mat_em_g.push_back(m->emission.g); f.mat_em_b.push_back(m->emission.b);
        }
        return f;
    }

    void uploadToGPU(PtContext* ctx) const {  // Scene.cuh:274-281 + main.cu:261-278
        FlatScene f = flattenSoA();
        PtSceneSoA v = f.view();
        check(pt_upload_scene(ctx, &v));
    }
};

// ---- renderer: RAII over the C ABI ---------------------------------------------------------
class Renderer {
    PtContext* ctx_{nullptr};

public:
    explicit Renderer(int device = 0) { check(pt_create(device, &ctx_)); }
    ~Renderer() { pt_destroy(ctx_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    PtContext* ctx() const { return ctx_; }
    void setScene(const SceneManager& s) { s.uploadToGPU(ctx_); }
    void setScene(const PtSceneSoA& s) { check(pt_upload_scene(ctx_, &s)); }
    void setCamera(const PtCamera& c) { check(pt_set_camera(ctx_, &c)); }  // initializeGPUData
    void reset() { check(pt_reset(ctx_)); }
    void render(uint32_t sample_begin, uint32_t sample_count, const PtRenderParams& p) { check(pt_render(ctx_, sample_begin, sample_count, &p)); }
    void resolve(uint32_t mode) { check(pt_resolve(ctx_, mode)); }
    void readback(void* dst, size_t bytes) { check(pt_readback(ctx_, dst, bytes)); }
    void readbackAccum(float* dst, size_t bytes) { check(pt_readback_accum(ctx_, dst, bytes)); }
    void sync() { check(pt_sync(ctx_)); }
    void setFramesInFlight(uint32_t n) { check(pt_set_frames_in_flight(ctx_, n)); }  // overlap consecutive launches (pt.h)
    void flush() { check(pt_flush(ctx_)); }
    void reduce(void* nccl_comm, int root) { check(pt_reduce(ctx_, nccl_comm, root)); }  // multi-GPU sum, one process per GPU
};

inline PtRenderParams defaultParams() {
    PtRenderParams p{};
    p.integrator = PT_INTEGRATOR_PATH; p.max_depth = 50; p.rr_depth = 3; p.seed = 0x5EED0001u;
    p.flags = 0; p.acc_mode = PT_ACC_SUM; p.fused_rgba8 = 0; p.pixels_per_wave = 0;
    return p;
}

}  // namespace pt
