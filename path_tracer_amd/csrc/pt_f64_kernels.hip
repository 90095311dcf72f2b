// pt_f64_kernels.hip — the FP64 side: primary-hit ids and the `direct` integrator.
//
// These restate cpu_pathtracer (FP64 throughout): Scene::intersect (Scene.h:52-72),
// Sphere::intersect / Plane::intersect (ImplicitObject.h:57-95,146-172), Triangle::intersect
// (Polygon.h:58-94), Camera::getRayForPixel (Camera.h:124-142), Scene::isVisible (Scene.h:75-94),
// Scene::sampleLights (Scene.h:97-112), renderRegion (src/main.cpp:14-53).
//
// COMPILED WITH -ffp-contract=off: every +,-,*,/,sqrt is then a single IEEE-754 double
// operation in the reference's evaluation order, so primary-hit ids are bit-exact against a
// CPU build of the same formulas (tests/test_parity_gpu.py).  One thread = one pixel; scene
// operands are wave-uniform scalar loads.
#include "pt_device.h"

namespace {

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator*(D3 a, double t) { return d3(a.x * t, a.y * t, a.z * t); }
__device__ __forceinline__ D3 operator-(D3 a) { return d3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // Vec3.h:128-131
__device__ __forceinline__ D3 cross(D3 u, D3 v) {                                                 // Vec3.h:133-140
    return d3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__device__ __forceinline__ double length(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // Vec3.h:40-46
__device__ __forceinline__ D3 normalized(D3 a) {                                                   // Vec3.h:48-63
    const double len = length(a);
    if (len > 0) { const double il = 1 / len; a.x *= il; a.y *= il; a.z *= il; }
    return a;
}
__device__ __forceinline__ D3 ld3(const double* p) { return d3(p[0], p[1], p[2]); }

struct DRay { D3 o, d; double min_t, max_t; };
// Ray(origin, direction): normalises; min_t 1e-8, max_t 1e30 (Ray.h:43-58)
__device__ __forceinline__ DRay make_ray(D3 o, D3 d) { return DRay{o, normalized(d), 1e-8, 1e30}; }
__device__ __forceinline__ bool valid_t(const DRay& r, double t) { return t >= r.min_t && t <= r.max_t; }  // Ray.h:192-194

struct DHit { bool hit; double t; uint32_t obj; uint32_t mat; int kind; int prim; };

// Closest hit in INSERTION order with a strict '<' (Scene.h:57-69): the SoA splits objects by class, so exact distance
// ties go to the smaller object id.  The loops carry the distance and the winner's index only (a 64-bit and a 32-bit
// select per candidate instead of the whole record's seven); the object id is looked up where two distances are equal
// to the bit, and the record is filled in once, after the loops (+1.8 % on BASELINE config 1, same bits; round 3).
__device__ __forceinline__ uint32_t obj_of(const DevSceneD& sc, int g) {   // g: spheres, then planes, then triangles
    return g < sc.ns ? sc.sph_info[3 * g + 1] : (g < sc.ns + sc.np ? sc.pl_info[2 * (g - sc.ns) + 1] : sc.tri_info[2 * (g - sc.ns - sc.np) + 1]);
}
__device__ __forceinline__ void consider(const DevSceneD& sc, double& bt, int& bg, double t, int g) {
    if (t < bt || (t == bt && bg >= 0 && obj_of(sc, g) < obj_of(sc, bg))) { bt = t; bg = g; }
}

__device__ __forceinline__ bool sphere_t(const double* s, const DRay& ray, double& root) {  // ImplicitObject.h:57-79
    const D3 oc = ray.o - d3(s[0], s[1], s[2]);
    const double a = 1.0;
    const double half_b = dot(oc, ray.d);
    const double c = dot(oc, oc) - s[3] * s[3];
    const double disc = half_b * half_b - a * c;
    if (disc < 0) return false;
    const double sqrtd = sqrt(disc);
    root = (-half_b - sqrtd) / a;
    if (!valid_t(ray, root)) {
        root = (-half_b + sqrtd) / a;
        if (!valid_t(ray, root)) return false;
    }
    return true;
}
__device__ __forceinline__ bool plane_t(const double* p, const DRay& ray, double& t) {  // ImplicitObject.h:146-158
    const D3 n = d3(p[3], p[4], p[5]);
    const double denom = dot(n, ray.d);
    if (fabs(denom) < 1e-8) return false;
    t = dot(d3(p[0], p[1], p[2]) - ray.o, n) / denom;
    return valid_t(ray, t);
}
__device__ __forceinline__ bool tri_t(const double* v, const DRay& ray, double& t) {  // Polygon.h:58-94
    const D3 v0 = ld3(v), e1 = ld3(v + 3) - v0, e2 = ld3(v + 6) - v0;
    const D3 h = cross(ray.d, e2);
    const double a = dot(e1, h);
    if (fabs(a) < 1e-8) return false;
    const double f = 1.0 / a;
    const D3 s = ray.o - v0;
    const double u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return false;
    const D3 q = cross(s, e1);
    const double w = f * dot(ray.d, q);
    if (w < 0.0 || u + w > 1.0) return false;
    t = f * dot(e2, q);
    return valid_t(ray, t);
}

__device__ DHit scene_hit(const DevSceneD& sc, const DRay& ray) {
    double bt = 1.0e300;   // (beyond any valid distance: max_t is 1e30)
    int bg = -1;
    for (int i = 0; i < sc.ns; ++i) { double t; if (sphere_t(sc.sph + 4 * i, ray, t)) consider(sc, bt, bg, t, i); }
    for (int i = 0; i < sc.np; ++i) { double t; if (plane_t(sc.pl + 6 * i, ray, t)) consider(sc, bt, bg, t, sc.ns + i); }
    for (int i = 0; i < sc.nt; ++i) { double t; if (tri_t(sc.tri + 9 * i, ray, t)) consider(sc, bt, bg, t, sc.ns + sc.np + i); }
    if (bg < 0) return DHit{false, 0.0, 0u, 0xFFFFFFFFu, 0, 0};
    if (bg < sc.ns) return DHit{true, bt, sc.sph_info[3 * bg + 1], sc.sph_info[3 * bg], 0, bg};
    if (bg < sc.ns + sc.np) return DHit{true, bt, sc.pl_info[2 * (bg - sc.ns) + 1], sc.pl_info[2 * (bg - sc.ns)], 1, bg - sc.ns};
    const int k = bg - sc.ns - sc.np;
    return DHit{true, bt, sc.tri_info[2 * k + 1], sc.tri_info[2 * k], 2, k};
}

// Camera::getRayForPixel (Camera.h:124-142, use_dof == false) or generateCameraRay in double.
template <int MODEL = -1>   // the camera model as a template argument (the `direct` kernel: only one model's 24 SGPRs of camera stay live in the sample loop), or read from the camera
__device__ __forceinline__ DRay camera_ray(const DevCamD& k, uint32_t x, uint32_t y, double jx, double jy) {
    const D3 origin = ld3(k.origin);
    if (MODEL < 0 ? k.model == 1u : MODEL == 1) {
        const double u = ((double)x + jx) / (double)(k.width - 1);
        const double v = 1.0 - ((double)y + jy) / (double)(k.height - 1);
        const D3 pixel_loc = ld3(k.top_left) + ld3(k.horizontal) * u + ld3(k.vertical) * v;
        return make_ray(origin, normalized(pixel_loc - origin));
    }
    const double u = ((double)x + jx) / (double)k.width;
    const double v = ((double)y + jy) / (double)k.height;
    const double xo = (2.0 * u - 1.0) * k.half_w, yo = (2.0 * v - 1.0) * k.half_h;
    return make_ray(origin, normalized(ld3(k.fwd) + ld3(k.right) * xo + ld3(k.up) * yo));
}

__global__ __launch_bounds__(256) void pt_primary_ids_kernel(const DevSceneD sc, const DevCamD cam, uint32_t* obj,
                                                             uint32_t* mat) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= cam.width * cam.height) return;
    const uint32_t y = p / cam.width, x = p - y * cam.width;
    const DRay ray = camera_ray(cam, x, y, 0.5, 0.5);  // getRayForPixel(x, y, false)
    const DHit h = scene_hit(sc, ray);
    if (obj) obj[p] = h.hit ? h.obj : 0xFFFFFFFFu;
    if (mat) mat[p] = h.hit ? h.mat : 0xFFFFFFFFu;
}

// Counter RNG
__device__ __forceinline__ uint32_t pt_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t pt_draw(uint32_t key, uint32_t slot, uint32_t dim) {
    return pt_hash(key + (slot * 8u + dim + 1u) * 0x9E3779B9U);
}
__device__ __forceinline__ double duni(uint32_t bits) { return (double)bits * (1.0 / 4294967296.0); }  // [0,1), Random::get

// Scene::isVisible, Scene.h:75-94
__device__ bool visible(const DevSceneD& sc, D3 from, D3 to) {
    const D3 direction = to - from;
    const double distance = length(direction);
    DRay sh = make_ray(from, normalized(direction));
    sh.max_t = length(to - sh.o);  // setEndpoint, Ray.h:175-178
    for (int i = 0; i < sc.ns; ++i) { double t; if (sphere_t(sc.sph + 4 * i, sh, t) && t < distance) return false; }
    for (int i = 0; i < sc.np; ++i) { double t; if (plane_t(sc.pl + 6 * i, sh, t) && t < distance) return false; }
    for (int i = 0; i < sc.nt; ++i) { double t; if (tri_t(sc.tri + 9 * i, sh, t) && t < distance) return false; }
    return true;
}

// renderRegion, cpu_pathtracer/src/main.cpp:14-53: ambient 0.1 + (emission | one sample per
// light * colour*cos*50/d^2), background 0.2.  One thread = one pixel, samples in a loop.
// COUNT: the instrumented build (PT_FLAG_COUNTERS) adds its event counts to `counters` (PtCounters layout): samples,
// rays (primary + shadow), hits (primary), nee_samples (light samples drawn), shadow_rays (Scene::isVisible calls).
// Register budget: left alone the kernel takes 134 VGPRs (3 waves/SIMD); asked for 4 it fits 126 with six spilled and
// runs 6 % faster (21.8 -> 23.2 Gsamples/s on BASELINE config 1); 5 waves (96 VGPRs, 39 spilled) lose 4 %.
// ---- sin, cos, acos of the light sample (Sphere::getRandomPoint, ImplicitObject.h:111-125) ----------------------------
// The reference calls std::cos / std::sin / std::acos, whose last bits depend on its libm and its -ffast-math build; through
// the tie of D6 (hit.distance < distance on the sampled light itself) one such bit decides whether a light sample counts.
// Like the random streams, the three functions are therefore specified: plain IEEE double
// +, -, *, /, sqrt without contraction, within 1 ulp of the libm values, evaluated the same way by the oracle -- frames of
// the `direct` integrator are equal to the oracle's bit for bit (tests/test_parity_gpu.py).
//  sincos, 0 <= x < 8: n = the nearest multiple of pi/2; r = (x - n A) - n B with A = pi/2 cut to 33 bits (n A is exact) and
//  B = the next 53 bits; an odd and an even polynomial of degree 13 and 14 in r on [-pi/4, pi/4]; the quadrant swaps them.
//  acos, -1 <= x <= 1: |x| < 1/2: pi/2 - (x + x R(x^2)); else with z = (1 -+ x) / 2, s = sqrt z: 2 (s + s R(z)), or pi minus
//  that; R(z) = z P(z) / Q(z) of degrees 5 and 4.
//  (against the library's sincos + acos: -0.6 %)
constexpr double FM_INVPIO2 = 6.36619772367581382433e-01, FM_PIO2_A = 1.57079632673412561417e+00, FM_PIO2_B = 6.07710050650619224932e-11;
constexpr double FM_S1 = -1.66666666666666324348e-01, FM_S2 = 8.33333333332248946124e-03, FM_S3 = -1.98412698298579493134e-04,
                 FM_S4 = 2.75573137070700676789e-06, FM_S5 = -2.50507602534068634195e-08, FM_S6 = 1.58969099521155010221e-10;
constexpr double FM_C1 = 4.16666666666666019037e-02, FM_C2 = -1.38888888888741095749e-03, FM_C3 = 2.48015872894767294178e-05,
                 FM_C4 = -2.75573143513906633035e-07, FM_C5 = 2.08757232129817482790e-09, FM_C6 = -1.13596475577881948265e-11;
constexpr double FM_PIO2_HI = 1.57079632679489655800e+00, FM_PIO2_LO = 6.12323399573676603587e-17, FM_PI = 3.14159265358979311600e+00;
constexpr double FM_P0 = 1.66666666666666657415e-01, FM_P1 = -3.25565818622400915405e-01, FM_P2 = 2.01212532134862925881e-01,
                 FM_P3 = -4.00555345006794114027e-02, FM_P4 = 7.91534994289814532176e-04, FM_P5 = 3.47933107596021167570e-05;
constexpr double FM_Q1 = -2.40339491173441421878e+00, FM_Q2 = 2.02094576023350569471e+00, FM_Q3 = -6.88283971605453293030e-01,
                 FM_Q4 = 7.70381505559019352791e-02;

__device__ __forceinline__ void fm_sincos(double x, double& sn, double& cs) {
    const int n = (int)(x * FM_INVPIO2 + 0.5);
    const double r = (x - (double)n * FM_PIO2_A) - (double)n * FM_PIO2_B;
    const double z = r * r;
    const double ps = r + r * (z * (FM_S1 + z * (FM_S2 + z * (FM_S3 + z * (FM_S4 + z * (FM_S5 + z * FM_S6))))));
    const double pc = 1.0 - z * (0.5 - z * (FM_C1 + z * (FM_C2 + z * (FM_C3 + z * (FM_C4 + z * (FM_C5 + z * FM_C6))))));
    const bool swap = (n & 1) != 0;
    const double a = swap ? pc : ps, b = swap ? ps : pc;      // quadrant 0: (ps, pc), 1: (pc, -ps), 2: (-ps, -pc), 3: (-pc, ps)
    sn = (n & 2) ? -a : a;
    cs = (((n + 1) & 2) != 0) ? -b : b;
}
__device__ __forceinline__ double fm_ratio(double z) {
    const double p = z * (FM_P0 + z * (FM_P1 + z * (FM_P2 + z * (FM_P3 + z * (FM_P4 + z * FM_P5)))));
    const double q = 1.0 + z * (FM_Q1 + z * (FM_Q2 + z * (FM_Q3 + z * FM_Q4)));
    return p / q;
}
__device__ __forceinline__ double fm_acos(double x) {
    // the three ranges share their rational and their square root: every wave holds lanes of all three, and evaluated range
    // by range the quotient (an IEEE division) would run twice.  Per lane the operations are those of the description above.
    const double ax = fabs(x);
    const bool small = ax < 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;      // (1 + x for x < 0 and 1 - x for x > 0 are the same operation on |x|)
    const double r = fm_ratio(z);
    const double s = sqrt(z);
    const double rs = r * s;
    const double mid = FM_PIO2_HI - (x - (FM_PIO2_LO - x * r));
    const double pos = 2.0 * (s + rs);
    const double neg = FM_PI - 2.0 * (s + (rs - FM_PIO2_LO));
    return small ? mid : (x < 0 ? neg : pos);
}

#ifndef PT_DIRECT_WAVES
#define PT_DIRECT_WAVES 4
#endif
template <bool COUNT, int MODEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PT_DIRECT_WAVES))) void pt_direct_kernel(const DevSceneD sc, const DevCamD cam, float4* accum, uint32_t s0,
                                                        uint32_t count, uint32_t seed, uint64_t* counters) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= cam.width * cam.height) return;
    const uint32_t y = p / cam.width, x = p - y * cam.width;
    double px[3] = {0, 0, 0};
    uint32_t c_hits = 0, c_lights = 0, c_shadow = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t key = pt_hash(pt_hash(pt_hash(seed) + p) + (s0 + i));
        const DRay ray = camera_ray<MODEL>(cam, x, y, duni(pt_draw(key, 0, 0)), duni(pt_draw(key, 0, 1)));
        const DHit h = scene_hit(sc, ray);
        if (COUNT) c_hits += h.hit ? 1u : 0u;
        if (!h.hit) { px[0] += 0.2; px[1] += 0.2; px[2] += 0.2; continue; }
        px[0] += 0.1; px[1] += 0.1; px[2] += 0.1;
        if (h.kind == 0 && sc.sph_info[3 * h.prim + 2] != 0u) {
            px[0] += sc.sph_em[3 * h.prim]; px[1] += sc.sph_em[3 * h.prim + 1]; px[2] += sc.sph_em[3 * h.prim + 2];
            continue;
        }
        // hit record of the winner
        const D3 hp = ray.o + ray.d * h.t;
        D3 n; const uint32_t m = h.mat;
        if (h.kind == 0) {
            const double* s = sc.sph + 4 * h.prim;
            n = (hp - d3(s[0], s[1], s[2])) * (1 / s[3]);  // getNormalAt, ImplicitObject.h:97-99 + Vec3.h:118-121
        } else if (h.kind == 1) {
            n = ld3(sc.pl + 6 * h.prim + 3);
        } else {
            const double* v = sc.tri + 9 * h.prim;
            n = normalized(cross(ld3(v + 3) - ld3(v), ld3(v + 6) - ld3(v)));
        }
        if (!(dot(ray.d, n) < 0)) n = -n;  // setFaceNormal, Intersection.h:16-19
        double direct[3] = {0, 0, 0};
        uint32_t light = 0;
        for (int l = 0; l < sc.ns; ++l) {
            if (sc.sph_info[3 * l + 2] == 0u) continue;
            const double u = duni(pt_draw(key, 1 + light, 0)), v = duni(pt_draw(key, 1 + light, 1));
            light++;
            if (COUNT) c_lights++;
            // Sphere::getRandomPoint, ImplicitObject.h:111-125
            const double theta = 2.0 * M_PI * u;
            const double phi = fm_acos(2.0 * v - 1.0);
            double sin_theta, cos_theta_, sin_phi, cos_phi;
            fm_sincos(theta, sin_theta, cos_theta_);
            fm_sincos(phi, sin_phi, cos_phi);
            const D3 dir = d3(cos_theta_ * sin_phi, sin_theta * sin_phi, cos_phi);
            const double* s = sc.sph + 4 * l;
            const D3 lp = d3(s[0], s[1], s[2]) + dir * s[3];
            const D3 light_dir = normalized(lp - hp);
            const double cos_theta = fmax(0.0, dot(n, light_dir));
            if (COUNT) c_shadow += cos_theta > 0 ? 1u : 0u;
            if (cos_theta > 0 && visible(sc, hp, lp)) {
                const double ld = length(lp - hp);
                const double att = 1.0 / (ld * ld);
                for (int ch = 0; ch < 3; ++ch) direct[ch] += sc.mat_albedo[3 * m + ch] * cos_theta * att * 50.0;
            }
        }
        px[0] += direct[0]; px[1] += direct[1]; px[2] += direct[2];  // pixel_color += direct_light, main.cpp:42
    }
    float4 a = accum[p];
    a.x += (float)px[0]; a.y += (float)px[1]; a.z += (float)px[2]; a.w += (float)count;
    accum[p] = a;
    if (COUNT && counters) {   // PtCounters: [0] samples, [1] rays, [2] hits, [10] nee_samples, [11] shadow_rays
        atomicAdd((unsigned long long*)&counters[0], (unsigned long long)count);
        atomicAdd((unsigned long long*)&counters[1], (unsigned long long)(count + c_shadow));
        atomicAdd((unsigned long long*)&counters[2], (unsigned long long)c_hits);
        atomicAdd((unsigned long long*)&counters[10], (unsigned long long)c_lights);
        atomicAdd((unsigned long long*)&counters[11], (unsigned long long)c_shadow);
    }
}

}  // namespace

hipError_t pt_launch_primary_ids(const DevSceneD& sc, const DevCamD& cam, uint32_t* obj, uint32_t* mat, hipStream_t stream) {
    const uint32_t n = cam.width * cam.height;
    hipLaunchKernelGGL(pt_primary_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, cam, obj, mat);
    return hipGetLastError();
}

hipError_t pt_launch_direct(const DevSceneD& sc, const DevCamD& cam, float4* accum, uint32_t s0, uint32_t count,
                            uint32_t seed, uint64_t* counters, hipStream_t stream) {
    const uint32_t n = cam.width * cam.height;
    const dim3 grid((n + 255) / 256), block(256);
    const bool cpu_cam = cam.model == 1u;
    if (counters) {
        if (cpu_cam) hipLaunchKernelGGL((pt_direct_kernel<true, 1>), grid, block, 0, stream, sc, cam, accum, s0, count, seed, counters);
        else hipLaunchKernelGGL((pt_direct_kernel<true, 0>), grid, block, 0, stream, sc, cam, accum, s0, count, seed, counters);
    } else {
        if (cpu_cam) hipLaunchKernelGGL((pt_direct_kernel<false, 1>), grid, block, 0, stream, sc, cam, accum, s0, count, seed, counters);
        else hipLaunchKernelGGL((pt_direct_kernel<false, 0>), grid, block, 0, stream, sc, cam, accum, s0, count, seed, counters);
    }
    return hipGetLastError();
}
