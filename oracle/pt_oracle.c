/*
 * pt_oracle.c — CPU oracle for the per-pixel render path.  TEST INFRASTRUCTURE.
 * See pt_oracle.h for scope, citations and the "parity unpinned" statement.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp; no FMA contraction so
 * the FP64 paths are IEEE-reproducible and comparable bit-for-bit with the HIP
 * FP64 kernels, which are compiled with contraction off as well).
 */
#define _USE_MATH_DEFINES
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ RNG --- */
/* Counter-based generator.  hash = "lowbias32" integer finaliser
 * (two 32-bit multiplies).  A path is keyed by (seed, pixel, sample); a draw by
 * (slot, dim): slot 0 = camera, slot b+1 = bounce b; dim < 8. */
uint32_t pto_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
uint32_t pto_path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return pto_hash(pto_hash(pto_hash(seed) + pixel) + sample);
}
uint32_t pto_draw(uint32_t key, uint32_t slot, uint32_t dim) {
    return pto_hash(key + (slot * 8u + dim + 1u) * 0x9E3779B9U);
}

/* ----------------------------------------------------- path, two builds --- */
#define FN(n) f32_##n
#define REAL float
#define R_SQRT sqrtf
#define R_SIN sinf
#define R_COS cosf
#define R_TAN tanf
#define R_POW powf
#define R_FABS fabsf
#define R_FMIN fminf
#define R_FMAX fmaxf
#include "path_impl.inc"
#undef FN
#undef REAL
#undef R_SQRT
#undef R_SIN
#undef R_COS
#undef R_TAN
#undef R_POW
#undef R_FABS
#undef R_FMIN
#undef R_FMAX

#define FN(n) f64_##n
#define REAL double
#define R_SQRT sqrt
#define R_SIN sin
#define R_COS cos
#define R_TAN tan
#define R_POW pow
#define R_FABS fabs
#define R_FMIN fmin
#define R_FMAX fmax
#include "path_impl.inc"
#undef FN
#undef REAL

typedef struct {
    uint32_t nlights;
    uint32_t* light_sphere;
    int32_t* sphere_light;
} LightList;

static int build_lights(const PtSceneSoA* sc, LightList* ll) {
    ll->nlights = 0;
    ll->light_sphere = (uint32_t*)malloc(sizeof(uint32_t) * (sc->num_spheres + 1));
    ll->sphere_light = (int32_t*)malloc(sizeof(int32_t) * (sc->num_spheres + 1));
    if (!ll->light_sphere || !ll->sphere_light) return -1;
    for (uint32_t i = 0; i < sc->num_spheres; i++) {
        if (sc->mat_type[sc->sph_mat[i]] == PT_MAT_EMISSIVE) {
            ll->sphere_light[i] = (int32_t)ll->nlights;
            ll->light_sphere[ll->nlights++] = i;
        } else {
            ll->sphere_light[i] = -1;
        }
    }
    return 0;
}
static void free_lights(LightList* ll) { free(ll->light_sphere); free(ll->sphere_light); }

static void counters_add(PtCounters* d, const PtCounters* s) {
    d->samples += s->samples; d->rays += s->rays; d->hits += s->hits; d->sphere_roots += s->sphere_roots;
    for (int i = 0; i < 5; i++) d->bounces[i] += s->bounces[i];
    d->diffuse_updates += s->diffuse_updates; d->nee_samples += s->nee_samples; d->shadow_rays += s->shadow_rays;
}

int pto_render_path(const PtSceneSoA* sc, const PtCamera* cam, uint32_t y0, uint32_t y1, uint32_t s0, uint32_t n,
                    const PtRenderParams* prm, int precision, float* accum, PtCounters* counters) {
    if (!sc || !cam || !prm || !accum || (precision != 32 && precision != 64)) return PT_ERR_INVALID;
    LightList ll;
    if (build_lights(sc, &ll)) return PT_ERR_INVALID;
    const uint32_t W = cam->width;
    f32_Ctx c32 = {sc, ll.nlights, ll.light_sphere, ll.sphere_light, prm->flags, prm->max_depth, prm->rr_depth};
    f64_Ctx c64 = {sc, ll.nlights, ll.light_sphere, ll.sphere_light, prm->flags, prm->max_depth, prm->rr_depth};
    f32_Cam k32 = f32_make_cam(cam);
    f64_Cam k64 = f64_make_cam(cam);
    PtCounters total; memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        PtCounters local; memset(&local, 0, sizeof local);
        PtCounters* cp = counters ? &local : NULL;
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = y0; y < (int64_t)y1; y++) {
            for (uint32_t x = 0; x < W; x++) {
                float* px = accum + 4 * ((size_t)y * W + x);
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t s = s0 + i;
                    float c[3];
                    if (precision == 32) {
                        f32_V3 L = f32_sample(&c32, &k32, x, (uint32_t)y, s, prm->seed, cp);
                        c[0] = L.x; c[1] = L.y; c[2] = L.z;
                    } else {
                        f64_V3 L = f64_sample(&c64, &k64, x, (uint32_t)y, s, prm->seed, cp);
                        c[0] = (float)L.x; c[1] = (float)L.y; c[2] = (float)L.z;
                    }
                    if (prm->acc_mode == PT_ACC_RUNNING_MEAN) {
                        /* renderKernel :297-304; s plays frame_count */
                        if (s > 0) {
                            float t = 1.0f / (float)(s + 1);
                            c[0] = c[0] * t + px[0] * (1.0f - t);
                            c[1] = c[1] * t + px[1] * (1.0f - t);
                            c[2] = c[2] * t + px[2] * (1.0f - t);
                        }
                        px[0] = c[0]; px[1] = c[1]; px[2] = c[2]; px[3] = 1.0f;
                    } else {
                        /* the HIP kernel sums a launch's samples in registers, then adds once;
                         * one-sample launches make the two orders identical */
                        px[0] += c[0]; px[1] += c[1]; px[2] += c[2]; px[3] += 1.0f;
                    }
                }
            }
        }
        if (counters) {
#pragma omp critical
            counters_add(&total, &local);
        }
    }
    if (counters) counters_add(counters, &total);
    free_lights(&ll);
    return PT_OK;
}

/* Path lengths: the rays a sample traces are the iterations of traceRay's loop (raytracer_kernel.cu:206-208), read
 * here off the event counters (rays - shadow_rays) before and after the sample. */
int pto_path_length_histogram(const PtSceneSoA* sc, const PtCamera* cam, uint32_t s0, uint32_t n, const PtRenderParams* prm,
                              int precision, uint64_t* hist64) {
    if (!sc || !cam || !prm || !hist64 || (precision != 32 && precision != 64)) return PT_ERR_INVALID;
    LightList ll;
    if (build_lights(sc, &ll)) return PT_ERR_INVALID;
    f32_Ctx c32 = {sc, ll.nlights, ll.light_sphere, ll.sphere_light, prm->flags, prm->max_depth, prm->rr_depth};
    f64_Ctx c64 = {sc, ll.nlights, ll.light_sphere, ll.sphere_light, prm->flags, prm->max_depth, prm->rr_depth};
    f32_Cam k32 = f32_make_cam(cam);
    f64_Cam k64 = f64_make_cam(cam);
    memset(hist64, 0, 64 * sizeof(uint64_t));
#pragma omp parallel
    {
        uint64_t local[64]; memset(local, 0, sizeof local);
        PtCounters c; memset(&c, 0, sizeof c);
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = 0; y < (int64_t)cam->height; y++)
            for (uint32_t x = 0; x < cam->width; x++)
                for (uint32_t i = 0; i < n; i++) {
                    const uint64_t before = c.rays - c.shadow_rays;
                    if (precision == 32) (void)f32_sample(&c32, &k32, x, (uint32_t)y, s0 + i, prm->seed, &c);
                    else (void)f64_sample(&c64, &k64, x, (uint32_t)y, s0 + i, prm->seed, &c);
                    const uint64_t len = c.rays - c.shadow_rays - before;
                    local[len < 63 ? len : 63]++;
                }
#pragma omp critical
        for (int k = 0; k < 64; k++) hist64[k] += local[k];
    }
    free_lights(&ll);
    return PT_OK;
}

int pto_path_sample(const PtSceneSoA* sc, const PtCamera* cam, uint32_t x, uint32_t y, uint32_t s,
                    const PtRenderParams* prm, int precision, double* rgb) {
    if (!sc || !cam || !prm || !rgb) return PT_ERR_INVALID;
    LightList ll;
    if (build_lights(sc, &ll)) return PT_ERR_INVALID;
    if (precision == 32) {
        f32_Ctx c = {sc, ll.nlights, ll.light_sphere, ll.sphere_light, prm->flags, prm->max_depth, prm->rr_depth};
        f32_Cam k = f32_make_cam(cam);
        f32_V3 L = f32_sample(&c, &k, x, y, s, prm->seed, NULL);
        rgb[0] = L.x; rgb[1] = L.y; rgb[2] = L.z;
    } else {
        f64_Ctx c = {sc, ll.nlights, ll.light_sphere, ll.sphere_light, prm->flags, prm->max_depth, prm->rr_depth};
        f64_Cam k = f64_make_cam(cam);
        f64_V3 L = f64_sample(&c, &k, x, y, s, prm->seed, NULL);
        rgb[0] = L.x; rgb[1] = L.y; rgb[2] = L.z;
    }
    free_lights(&ll);
    return PT_OK;
}

int pto_path_stats(const PtSceneSoA* sc, const PtCamera* cam, uint32_t s0, uint32_t n, const PtRenderParams* prm,
                   double* mean, double* var) {
    if (!sc || !cam || !prm || !mean || !var || n < 2) return PT_ERR_INVALID;
    LightList ll;
    if (build_lights(sc, &ll)) return PT_ERR_INVALID;
    f64_Ctx c = {sc, ll.nlights, ll.light_sphere, ll.sphere_light, prm->flags, prm->max_depth, prm->rr_depth};
    f64_Cam k = f64_make_cam(cam);
    const uint32_t W = cam->width, H = cam->height;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t y = 0; y < (int64_t)H; y++) {
        for (uint32_t x = 0; x < W; x++) {
            double m[3] = {0, 0, 0}, q[3] = {0, 0, 0};
            for (uint32_t i = 0; i < n; i++) {
                f64_V3 L = f64_sample(&c, &k, x, (uint32_t)y, s0 + i, prm->seed, NULL);
                double v[3] = {L.x, L.y, L.z};
                for (int ch = 0; ch < 3; ch++) { /* Welford */
                    double d = v[ch] - m[ch];
                    m[ch] += d / (double)(i + 1);
                    q[ch] += d * (v[ch] - m[ch]);
                }
            }
            size_t p = (size_t)y * W + x;
            for (int ch = 0; ch < 3; ch++) { mean[3 * p + ch] = m[ch]; var[3 * p + ch] = q[ch] / (double)(n - 1); }
        }
    }
    free_lights(&ll);
    return PT_OK;
}

/* ------------------------------------------------ FP64 side: cpu_pathtracer --- */
typedef struct { double x, y, z; } D3;
static inline D3 d3(double x, double y, double z) { D3 r = {x, y, z}; return r; }
static inline D3 dadd(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline D3 dsub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline D3 dmul(D3 a, double t) { return d3(a.x * t, a.y * t, a.z * t); }            /* Vec3.h:108-111 */
static inline D3 ddiv(D3 a, double t) { return dmul(a, 1 / t); }                              /* Vec3.h:118-121 */
static inline D3 dneg(D3 a) { return d3(-a.x, -a.y, -a.z); }
static inline double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          /* Vec3.h:128-131 */
static inline D3 dcross(D3 u, D3 v) {                                                         /* Vec3.h:133-140 */
    return d3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
static inline double dlen(D3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }           /* Vec3.h:40-46 */
static inline D3 dnormalized(D3 a) {                                                          /* Vec3.h:48-63 */
    double len = dlen(a);
    if (len > 0) { double il = 1 / len; a.x *= il; a.y *= il; a.z *= il; }
    return a;
}

typedef struct { D3 o, d; double min_t, max_t; } DRay;
/* Ray(origin, direction): normalises, min_t 1e-8, max_t 1e30 (Ray.h:43-58) */
static inline DRay dray(D3 o, D3 d) { DRay r; r.o = o; r.d = dnormalized(d); r.min_t = 1e-8; r.max_t = 1e30; return r; }
static inline int dvalid(const DRay* r, double t) { return t >= r->min_t && t <= r->max_t; } /* Ray.h:192-194 */

typedef struct {
    D3 origin, u, v, w, horizontal, vertical, top_left;
    /* CUDA-model camera evaluated in double for the id pass */
    D3 fwd, right, up; double half_w, half_h;
    uint32_t width, height, model;
} DCam;

static DCam make_dcam(const PtCamera* c) {
    DCam k; memset(&k, 0, sizeof k);
    k.width = c->width; k.height = c->height; k.model = c->model;
    if (c->model == PT_CAMERA_CPU) {
        /* Camera::initialize, cpu_pathtracer/include/Camera.h:65-85 */
        D3 o = d3(c->origin[0], c->origin[1], c->origin[2]);
        D3 tg = d3(c->target[0], c->target[1], c->target[2]);
        D3 up = d3(c->up[0], c->up[1], c->up[2]);
        if (c->normalize_up) up = dnormalized(up);
        k.origin = o;
        k.w = dnormalized(dsub(o, tg));
        k.u = dnormalized(dcross(up, k.w));
        k.v = dcross(k.w, k.u);
        double theta = c->fov_deg * M_PI / 180.0;
        double vh = 2.0 * tan(theta / 2.0);
        double vw = vh * c->aspect_ratio;
        k.horizontal = dmul(dmul(k.u, vw), c->focus_dist);
        k.vertical = dmul(dmul(k.v, vh), c->focus_dist);
        k.top_left = dsub(dadd(dsub(o, ddiv(k.horizontal, 2.0)), ddiv(k.vertical, 2.0)), dmul(k.w, c->focus_dist));
    } else {
        f64_Cam f = f64_make_cam(c); /* float basis (Camera.cu:3-23), widened */
        k.origin = d3(f.origin.x, f.origin.y, f.origin.z);
        k.fwd = d3(f.fwd.x, f.fwd.y, f.fwd.z);
        k.right = d3(f.right.x, f.right.y, f.right.z);
        k.up = d3(f.up.x, f.up.y, f.up.z);
        k.half_w = f.half_w; k.half_h = f.half_h;
    }
    return k;
}

/* Camera::getRayForPixel, Camera.h:124-142 (use_dof == false branch), or the CUDA
 * formula in double for PT_CAMERA_CUDA. jx, jy replace Random::get() / 0.5. */
static DRay dcam_ray(const DCam* k, uint32_t x, uint32_t y, double jx, double jy) {
    if (k->model == PT_CAMERA_CPU) {
        double u = ((double)x + jx) / (double)(k->width - 1);
        double v = 1.0 - ((double)y + jy) / (double)(k->height - 1);
        D3 pixel_loc = dadd(dadd(k->top_left, dmul(k->horizontal, u)), dmul(k->vertical, v));
        D3 dir = dnormalized(dsub(pixel_loc, k->origin));
        return dray(k->origin, dir);
    }
    double u = ((double)x + jx) / (double)k->width;
    double v = ((double)y + jy) / (double)k->height;
    double xo = (2.0 * u - 1.0) * k->half_w, yo = (2.0 * v - 1.0) * k->half_h;
    D3 dir = dnormalized(dadd(dadd(k->fwd, dmul(k->right, xo)), dmul(k->up, yo)));
    return dray(k->origin, dir);
}

typedef struct { int hit; double t; D3 p, n; int front; uint32_t obj, mat; int kind; uint32_t prim; } DHit;

static inline void dface(DHit* h, const DRay* r, D3 outward) { /* Intersection.h:16-19 */
    h->front = ddot(r->d, outward) < 0;
    h->n = h->front ? outward : dneg(outward);
}

/* Sphere::intersect, ImplicitObject.h:57-95 */
static DHit dsphere(const PtSceneSoA* sc, uint32_t i, const DRay* ray) {
    DHit h; memset(&h, 0, sizeof h);
    D3 c = d3(sc->sph_cx[i], sc->sph_cy[i], sc->sph_cz[i]);
    double radius = sc->sph_r[i];
    D3 oc = dsub(ray->o, c);
    double a = 1.0;
    double half_b = ddot(oc, ray->d);
    double cc = ddot(oc, oc) - radius * radius;
    double disc = half_b * half_b - a * cc;
    if (disc < 0) return h;
    double sqrtd = sqrt(disc);
    double root = (-half_b - sqrtd) / a;
    if (!dvalid(ray, root)) {
        root = (-half_b + sqrtd) / a;
        if (!dvalid(ray, root)) return h;
    }
    h.hit = 1; h.t = root; h.p = dadd(ray->o, dmul(ray->d, root));
    D3 n = ddiv(dsub(h.p, c), radius);
    h.obj = sc->sph_obj[i]; h.mat = sc->sph_mat[i]; h.kind = 0; h.prim = i;
    dface(&h, ray, n);
    return h;
}

/* Plane::intersect, ImplicitObject.h:146-172 (constructor normalises the normal, :143) */
static DHit dplane(const PtSceneSoA* sc, uint32_t i, const DRay* ray) {
    DHit h; memset(&h, 0, sizeof h);
    D3 n = dnormalized(d3(sc->pl_nx[i], sc->pl_ny[i], sc->pl_nz[i]));
    D3 p0 = d3(sc->pl_px[i], sc->pl_py[i], sc->pl_pz[i]);
    double denom = ddot(n, ray->d);
    if (fabs(denom) < 1e-8) return h;
    double t = ddot(dsub(p0, ray->o), n) / denom;
    if (!dvalid(ray, t)) return h;
    h.hit = 1; h.t = t; h.p = dadd(ray->o, dmul(ray->d, t));
    h.obj = sc->pl_obj[i]; h.mat = sc->pl_mat[i]; h.kind = 1; h.prim = i;
    dface(&h, ray, n);
    return h;
}

/* Triangle::intersect, cpu_pathtracer/include/Polygon.h:58-94 (Möller–Trumbore) */
static DHit dtri(const PtSceneSoA* sc, uint32_t i, const DRay* ray) {
    DHit h; memset(&h, 0, sizeof h);
    D3 v0 = d3(sc->tri_v0x[i], sc->tri_v0y[i], sc->tri_v0z[i]);
    D3 v1 = d3(sc->tri_v1x[i], sc->tri_v1y[i], sc->tri_v1z[i]);
    D3 v2 = d3(sc->tri_v2x[i], sc->tri_v2y[i], sc->tri_v2z[i]);
    D3 e1 = dsub(v1, v0), e2 = dsub(v2, v0);
    D3 hh = dcross(ray->d, e2);
    double a = ddot(e1, hh);
    if (fabs(a) < 1e-8) return h;
    double f = 1.0 / a;
    D3 s = dsub(ray->o, v0);
    double u = f * ddot(s, hh);
    if (u < 0.0 || u > 1.0) return h;
    D3 q = dcross(s, e1);
    double v = f * ddot(ray->d, q);
    if (v < 0.0 || u + v > 1.0) return h;
    double t = f * ddot(e2, q);
    if (!dvalid(ray, t)) return h;
    h.hit = 1; h.t = t; h.p = dadd(ray->o, dmul(ray->d, t));
    D3 n = dnormalized(dcross(e1, e2));
    h.obj = sc->tri_obj[i]; h.mat = sc->tri_mat[i]; h.kind = 2; h.prim = i;
    dface(&h, ray, n);
    return h;
}

/* Scene::intersect, Scene.h:52-72: closest over objects in INSERTION order with a
 * strict <.  The SoA splits objects by class, so insertion order is restored by
 * breaking exact distance ties on the smaller object id. */
static inline void dconsider(DHit* best, const DHit* h) {
    if (!h->hit) return;
    if (!best->hit || h->t < best->t || (h->t == best->t && h->obj < best->obj)) *best = *h;
}
static DHit dscene(const PtSceneSoA* sc, const DRay* ray) {
    DHit best; memset(&best, 0, sizeof best);
    for (uint32_t i = 0; i < sc->num_spheres; i++) { DHit h = dsphere(sc, i, ray); dconsider(&best, &h); }
    for (uint32_t i = 0; i < sc->num_planes; i++) { DHit h = dplane(sc, i, ray); dconsider(&best, &h); }
    for (uint32_t i = 0; i < sc->num_triangles; i++) { DHit h = dtri(sc, i, ray); dconsider(&best, &h); }
    return best;
}

int pto_primary_ids(const PtSceneSoA* sc, const PtCamera* cam, uint32_t* obj_ids, uint32_t* mat_ids) {
    if (!sc || !cam) return PT_ERR_INVALID;
    DCam k = make_dcam(cam);
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)cam->height; y++)
        for (uint32_t x = 0; x < cam->width; x++) {
            DRay r = dcam_ray(&k, x, (uint32_t)y, 0.5, 0.5); /* getRayForPixel(x, y, false) */
            DHit h = dscene(sc, &r);
            size_t p = (size_t)y * cam->width + x;
            if (obj_ids) obj_ids[p] = h.hit ? h.obj : 0xFFFFFFFFu;
            if (mat_ids) mat_ids[p] = h.hit ? h.mat : 0xFFFFFFFFu;
        }
    return PT_OK;
}

/* Rectangle::intersect, cpu_pathtracer/include/Polygon.h:103-153 (the CUDA class has the same body in float,
 * cuda_pathtracer/include/Polygon.cuh:130-168): the plane of the rectangle, then the hit point projected on the two
 * edges separately, alpha = (h . u) / (u . u) and beta = (h . v) / (v . v), both in [0, 1].  Statement order and
 * constants as in the CPU class; the distance test is the ray's own (dvalid), as for every other object here. */
static DHit drect(const double* o9, uint32_t obj, uint32_t mat, const DRay* ray) {
    DHit h; memset(&h, 0, sizeof h);
    D3 origin = d3(o9[0], o9[1], o9[2]), u = d3(o9[3], o9[4], o9[5]), v = d3(o9[6], o9[7], o9[8]);
    D3 normal = dnormalized(dcross(u, v));                 /* computeNormal, :115-117 */
    const double denom = ddot(normal, ray->d);
    if (fabs(denom) < 1e-8) return h;                      /* :124-125 */
    D3 p0_ro = dsub(origin, ray->o);
    const double t = ddot(p0_ro, normal) / denom;          /* :127-128 */
    if (!dvalid(ray, t)) return h;                         /* :130-131 */
    D3 hit_point = dadd(ray->o, dmul(ray->d, t));
    D3 hit_vec = dsub(hit_point, origin);
    const double alpha = ddot(hit_vec, u) / ddot(u, u);    /* :138-139 */
    const double beta = ddot(hit_vec, v) / ddot(v, v);
    if (alpha < 0.0 || alpha > 1.0 || beta < 0.0 || beta > 1.0) return h;   /* :142-143 */
    h.hit = 1; h.t = t; h.p = hit_point; h.obj = obj; h.mat = mat; h.kind = 3; h.prim = 0;
    dface(&h, ray, normal);
    return h;
}

/* pto_primary_ids for a scene whose rectangles are tested by the reference's own formula instead of through the two
 * triangles the host flattens them to: the triangles whose object id is one of rect_obj[] are skipped and the
 * rectangle (origin, u, v: 9 doubles each) stands in their place.  Pins the flattening (include/pt_host.hpp,
 * path_tracer_amd/scenes.py) against Rectangle::intersect: the two id maps may differ only where a ray meets the
 * rectangle's border or the diagonal its triangles share, within rounding. */
int pto_primary_ids_rect(const PtSceneSoA* sc, const PtCamera* cam, uint32_t n_rect, const double* rect_o_u_v,
                         const uint32_t* rect_obj, const uint32_t* rect_mat, uint32_t* obj_ids, uint32_t* mat_ids) {
    if (!sc || !cam || (n_rect && (!rect_o_u_v || !rect_obj || !rect_mat))) return PT_ERR_INVALID;
    DCam k = make_dcam(cam);
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)cam->height; y++)
        for (uint32_t x = 0; x < cam->width; x++) {
            DRay r = dcam_ray(&k, x, (uint32_t)y, 0.5, 0.5);
            DHit best; memset(&best, 0, sizeof best);
            for (uint32_t i = 0; i < sc->num_spheres; i++) { DHit h = dsphere(sc, i, &r); dconsider(&best, &h); }
            for (uint32_t i = 0; i < sc->num_planes; i++) { DHit h = dplane(sc, i, &r); dconsider(&best, &h); }
            for (uint32_t i = 0; i < sc->num_triangles; i++) {
                int of_a_rectangle = 0;
                for (uint32_t q = 0; q < n_rect; q++) of_a_rectangle |= sc->tri_obj[i] == rect_obj[q];
                if (of_a_rectangle) continue;
                DHit h = dtri(sc, i, &r); dconsider(&best, &h);
            }
            for (uint32_t q = 0; q < n_rect; q++) { DHit h = drect(rect_o_u_v + 9 * q, rect_obj[q], rect_mat[q], &r); dconsider(&best, &h); }
            size_t p = (size_t)y * cam->width + x;
            if (obj_ids) obj_ids[p] = best.hit ? best.obj : 0xFFFFFFFFu;
            if (mat_ids) mat_ids[p] = best.hit ? best.mat : 0xFFFFFFFFu;
        }
    return PT_OK;
}

/* Scene::isVisible, Scene.h:75-94 (polygons_ is empty in main.cpp; triangles of the
 * SoA are tested the same way for completeness). */
static int dvisible(const PtSceneSoA* sc, D3 from, D3 to) {
    D3 direction = dsub(to, from);
    double distance = dlen(direction);
    DRay sh = dray(from, dnormalized(direction));
    sh.max_t = dlen(dsub(to, sh.o)); /* setEndpoint, Ray.h:175-178 */
    for (uint32_t i = 0; i < sc->num_spheres; i++) { DHit h = dsphere(sc, i, &sh); if (h.hit && h.t < distance) return 0; }
    for (uint32_t i = 0; i < sc->num_planes; i++) { DHit h = dplane(sc, i, &sh); if (h.hit && h.t < distance) return 0; }
    for (uint32_t i = 0; i < sc->num_triangles; i++) { DHit h = dtri(sc, i, &sh); if (h.hit && h.t < distance) return 0; }
    return 1;
}

/* ---- sin, cos, acos of the light sample (Sphere::getRandomPoint, ImplicitObject.h:111-125) ----------------------
 * The reference calls std::cos / std::sin / std::acos; their last bits depend on its libm and on its -ffast-math build,
 * and through the tie of D6 (hit.distance < distance on the sampled light itself) one such bit decides whether a light
 * sample counts.  Like the random streams, the three functions are therefore SPECIFIED here and
 * evaluated the same way by the HIP kernel (pt_f64_kernels.hip restates them; no shared header): +, -, *, /, sqrt in
 * IEEE double without contraction, so that oracle and GPU agree to the bit.  Within 1 ulp of the libm values
 * (tests/test_oracle_pins.py).
 *  sincos, 0 <= x < 8: n = the nearest multiple of pi/2; r = (x - n A) - n B with A = pi/2 cut to 33 bits (n A is exact)
 *  and B = the next 53 bits; odd and even polynomials of degree 13 and 14 in r on [-pi/4, pi/4]; the quadrant swaps them.
 *  acos, -1 <= x <= 1: |x| < 1/2: pi/2 - (x + x R(x^2)); else with z = (1 -+ x) / 2, s = sqrt z: 2 (s + s R(z)), or pi
 *  minus that; R = z P(z) / Q(z) of degrees 5 and 4. */
static const double FM_INVPIO2 = 6.36619772367581382433e-01, FM_PIO2_A = 1.57079632673412561417e+00, FM_PIO2_B = 6.07710050650619224932e-11;
static const double FM_S1 = -1.66666666666666324348e-01, FM_S2 = 8.33333333332248946124e-03, FM_S3 = -1.98412698298579493134e-04,
                    FM_S4 = 2.75573137070700676789e-06, FM_S5 = -2.50507602534068634195e-08, FM_S6 = 1.58969099521155010221e-10;
static const double FM_C1 = 4.16666666666666019037e-02, FM_C2 = -1.38888888888741095749e-03, FM_C3 = 2.48015872894767294178e-05,
                    FM_C4 = -2.75573143513906633035e-07, FM_C5 = 2.08757232129817482790e-09, FM_C6 = -1.13596475577881948265e-11;
static const double FM_PIO2_HI = 1.57079632679489655800e+00, FM_PIO2_LO = 6.12323399573676603587e-17, FM_PI = 3.14159265358979311600e+00;
static const double FM_P0 = 1.66666666666666657415e-01, FM_P1 = -3.25565818622400915405e-01, FM_P2 = 2.01212532134862925881e-01,
                    FM_P3 = -4.00555345006794114027e-02, FM_P4 = 7.91534994289814532176e-04, FM_P5 = 3.47933107596021167570e-05;
static const double FM_Q1 = -2.40339491173441421878e+00, FM_Q2 = 2.02094576023350569471e+00, FM_Q3 = -6.88283971605453293030e-01,
                    FM_Q4 = 7.70381505559019352791e-02;

static void fm_sincos(double x, double* sn, double* cs) {
    const int n = (int)(x * FM_INVPIO2 + 0.5);
    const double r = (x - (double)n * FM_PIO2_A) - (double)n * FM_PIO2_B;
    const double z = r * r;
    const double ps = r + r * (z * (FM_S1 + z * (FM_S2 + z * (FM_S3 + z * (FM_S4 + z * (FM_S5 + z * FM_S6))))));
    const double pc = 1.0 - z * (0.5 - z * (FM_C1 + z * (FM_C2 + z * (FM_C3 + z * (FM_C4 + z * (FM_C5 + z * FM_C6))))));
    switch (n & 3) {
        case 0: *sn = ps; *cs = pc; break;
        case 1: *sn = pc; *cs = -ps; break;
        case 2: *sn = -ps; *cs = -pc; break;
        default: *sn = -pc; *cs = ps; break;
    }
}
static double fm_ratio(double z) {
    const double p = z * (FM_P0 + z * (FM_P1 + z * (FM_P2 + z * (FM_P3 + z * (FM_P4 + z * FM_P5)))));
    const double q = 1.0 + z * (FM_Q1 + z * (FM_Q2 + z * (FM_Q3 + z * FM_Q4)));
    return p / q;
}
static double fm_acos(double x) {
    if (fabs(x) < 0.5) return FM_PIO2_HI - (x - (FM_PIO2_LO - x * fm_ratio(x * x)));
    if (x < 0) {
        const double z = (1.0 + x) * 0.5, s = sqrt(z);
        const double w = fm_ratio(z) * s - FM_PIO2_LO;
        return FM_PI - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5, s = sqrt(z);
    return 2.0 * (s + fm_ratio(z) * s);
}
/* for the tests: out = {sin x, cos x, acos y} as specified above */
int pto_fmath(double x, double y, double* out3) {
    if (!out3) return PT_ERR_INVALID;
    fm_sincos(x, &out3[0], &out3[1]);
    out3[2] = fm_acos(y);
    return PT_OK;
}

static inline double duni(uint32_t bits) { return (double)bits * (1.0 / 4294967296.0); } /* [0,1) like Random::get */

/* One sample of renderRegion's body, cpu_pathtracer/src/main.cpp:23-46.
 * comp: [0]=primary id, [1]=ambient/background, [2..4]=emission, [5..7]=direct. */
static void direct_sample(const PtSceneSoA* sc, const DCam* k, uint32_t x, uint32_t y, uint32_t s, uint32_t seed, double* comp) {
    const uint32_t key = pto_path_key(seed, y * k->width + x, s);
    double jx = duni(pto_draw(key, 0, 0)), jy = duni(pto_draw(key, 0, 1));
    DRay ray = dcam_ray(k, x, y, jx, jy);
    DHit hit = dscene(sc, &ray);
    for (int i = 0; i < 8; i++) comp[i] = 0;
    if (!hit.hit) { comp[0] = -1; comp[1] = 0.2; return; }
    comp[0] = (double)hit.obj; comp[1] = 0.1;
    if (hit.kind == 0 && sc->sph_emissive[hit.prim]) {
        comp[2] = sc->sph_em_r[hit.prim]; comp[3] = sc->sph_em_g[hit.prim]; comp[4] = sc->sph_em_b[hit.prim];
        return;
    }
    /* hit->color is the object's colour: carried as the albedo of its material */
    const double col[3] = {sc->mat_albedo_r[hit.mat], sc->mat_albedo_g[hit.mat], sc->mat_albedo_b[hit.mat]};
    uint32_t light = 0;
    for (uint32_t i = 0; i < sc->num_spheres; i++) { /* lights_ = emissive objects in insertion order, Scene.h:38-44 */
        if (!sc->sph_emissive[i]) continue;
        /* Scene::sampleLights(1), Scene.h:97-112 + Sphere::getRandomPoint, ImplicitObject.h:111-125 */
        double u = duni(pto_draw(key, 1 + light, 0)), v = duni(pto_draw(key, 1 + light, 1));
        light++;
        double theta = 2.0 * M_PI * u;
        double phi = fm_acos(2.0 * v - 1.0);
        double sin_theta, cos_theta_, sin_phi, cos_phi;   /* cos(theta), sin(theta), sin(phi), cos(phi) as specified above */
        fm_sincos(theta, &sin_theta, &cos_theta_);
        fm_sincos(phi, &sin_phi, &cos_phi);
        D3 direction = d3(cos_theta_ * sin_phi, sin_theta * sin_phi, cos_phi);
        D3 lp = dadd(d3(sc->sph_cx[i], sc->sph_cy[i], sc->sph_cz[i]), dmul(direction, sc->sph_r[i]));
        D3 light_dir = dnormalized(dsub(lp, hit.p));
        double cos_theta = fmax(0.0, ddot(hit.n, light_dir));
        if (cos_theta > 0 && dvisible(sc, hit.p, lp)) {
            double ld = dlen(dsub(lp, hit.p));
            double att = 1.0 / (ld * ld);
            for (int ch = 0; ch < 3; ch++) comp[5 + ch] += col[ch] * cos_theta * att * 50.0;
        }
    }
}

int pto_direct_components(const PtSceneSoA* sc, const PtCamera* cam, uint32_t x, uint32_t y, uint32_t s,
                          const PtRenderParams* prm, double* out8) {
    if (!sc || !cam || !prm || !out8) return PT_ERR_INVALID;
    DCam k = make_dcam(cam);
    direct_sample(sc, &k, x, y, s, prm->seed, out8);
    return PT_OK;
}

int pto_render_direct(const PtSceneSoA* sc, const PtCamera* cam, uint32_t y0, uint32_t y1, uint32_t s0, uint32_t n,
                      const PtRenderParams* prm, float* accum) {
    if (!sc || !cam || !prm || !accum) return PT_ERR_INVALID;
    DCam k = make_dcam(cam);
    const uint32_t W = cam->width;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t y = y0; y < (int64_t)y1; y++)
        for (uint32_t x = 0; x < W; x++) {
            double px[3] = {0, 0, 0}; /* pixel_color, main.cpp:20 */
            for (uint32_t i = 0; i < n; i++) {
                double c[8];
                direct_sample(sc, &k, x, (uint32_t)y, s0 + i, prm->seed, c);
                /* pixel_color += ambient; += emission | direct; or += background (main.cpp:25-46) */
                for (int ch = 0; ch < 3; ch++) { px[ch] += c[1]; px[ch] += c[2 + ch] + c[5 + ch]; }
            }
            float* a = accum + 4 * ((size_t)y * W + x);
            a[0] += (float)px[0]; a[1] += (float)px[1]; a[2] += (float)px[2]; a[3] += (float)n;
        }
    return PT_OK;
}

/* ---------------------------------------------------------------- resolve --- */
int pto_resolve(const float* accum, uint32_t npix, uint32_t acc_mode, uint32_t mode, void* out) {
    if (!accum || !out) return PT_ERR_INVALID;
    for (uint32_t p = 0; p < npix; p++) {
        const float* a = accum + 4 * (size_t)p;
        float m[3];
        if (acc_mode == PT_ACC_RUNNING_MEAN) { m[0] = a[0]; m[1] = a[1]; m[2] = a[2]; }
        else { float inv = a[3] > 0 ? 1.0f / a[3] : 0.0f; m[0] = a[0] * inv; m[1] = a[1] * inv; m[2] = a[2] * inv; }
        if (mode == PT_RESOLVE_FLOAT_MEAN) {
            float* o = (float*)out + 4 * (size_t)p;
            o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = 1.0f;
        } else if (mode == PT_RESOLVE_RGBA8_CUDA) { /* Window.cuh:25-29 */
            uint32_t c[3];
            for (int ch = 0; ch < 3; ch++) c[ch] = (uint8_t)fminf(fmaxf(m[ch] * 255.0f, 0.0f), 255.0f);
            ((uint32_t*)out)[p] = (c[0] << 16) | (c[1] << 8) | c[2];
        } else if (mode == PT_RESOLVE_RGBA8_CPU) { /* main.cpp:228-236 + WindowX11.h:130-139 */
            uint32_t c[3];
            for (int ch = 0; ch < 3; ch++) {
                double v = pow(fmin(fmax((double)m[ch], 0.0), 1.0), 1.0 / 2.2);
                c[ch] = (uint8_t)(v * 255.0);
            }
            ((uint32_t*)out)[p] = (c[0] << 16) | (c[1] << 8) | c[2];
        } else if (mode == PT_RESOLVE_RGB8_PPM) { /* Camera.h:172-176, Color.h:45-51,86-91 */
            uint8_t* o = (uint8_t*)out + 3 * (size_t)p;
            for (int ch = 0; ch < 3; ch++) {
                double g = pow((double)m[ch], 1.0 / 2.2);
                double cl = fmin(fmax(g, 0.0), 1.0);
                o[ch] = (uint8_t)(cl * 255.0 + 0.5);
            }
        } else {
            return PT_ERR_INVALID;
        }
    }
    return PT_OK;
}
