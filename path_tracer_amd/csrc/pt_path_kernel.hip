// pt_path_kernel.hip — the FP32 `path` integrator for CDNA4 (gfx950, wave64), plus the
// resolve pass and the FP32 VALU ceiling micro-benchmark.
//
// What it computes is the reference's renderKernel / traceRay
// (cuda_pathtracer/src/raytracer_kernel.cu:202-305 with Material.cuh:36-259); how it computes
// it is not the reference's one-thread-one-pixel-one-sample launch:
//
//  * persistent lanes with path regeneration: a wave owns a pool of 4x4 pixel units drawn from
//    a global ticket counter; every lane traces ONE ray per loop iteration and, the moment its
//    path ends, continues with the camera ray it holds prepared (its pixel's next sample, or a
//    pixel pulled from the pool with a wave-level ballot+mbcnt prefix: no LDS, no atomics).
//    Bounce-count divergence (mirror corridors run to depth 50 with no Russian roulette,
//    SURVEY D12) therefore costs idle lanes only in a wave's last few iterations.
//  * a launch may hold several FRAMES (frames in flight, pt_api.cpp): the work items are
//    (frame, pixel) pairs from one ticket space, each frame's sums go to a plane of their own.
//  * a (frame, pixel) is owned by exactly one lane: its samples are summed in registers in
//    sample order and stored with one 16-byte store (serial launches: added to the float4
//    accumulation buffer with one 16-byte load and one 16-byte store) — deterministic, no
//    float atomics.
//  * the scene is read through wave-uniform indices, i.e. scalar loads into SGPRs (the scalar
//    data cache is CDNA's constant memory); only the winning primitive is gathered per lane.
//  * virtual dispatch is replaced by a material tag switch; curand by a counter RNG keyed on
//    (pixel, sample, bounce) so a sample's value does not depend on which
//    lane, launch or GPU traced it.
//  * next-event estimation (template NEE) is a second ray KIND in the same loop, so shadow
//    rays and path rays of different lanes share one intersection loop body.
#include "pt_device.h"

#include <float.h>

#include <mutex>

// build switches of the A/B variants (make variant NAME=x VFLAGS=-DPT_...=0); the defaults are the measured winners
#ifndef PT_UPD
#define PT_UPD 1
#endif
#ifndef PT_ONB
#define PT_ONB 1
#endif
#ifndef PT_PEEL
#define PT_PEEL 1
#endif
// 1: the regeneration block is skipped outright in trips that have nothing to flush, prepare or take (a wave that
//    drains its last long paths: no kernarg re-reads, no ballots): serial 720p / 1 spp frames +1 %
#ifndef PT_SKIPREGEN
#define PT_SKIPREGEN 1
#endif
// 1: the lane-parallel closest hit reads the lane's LDS rows together and tests without branches (130 -> 100
//    instructions per ray, two LDS waits instead of five): +0.3 %
#ifndef PT_FEWFAST
#define PT_FEWFAST 1
#endif
// (round 3, measured and dropped: s_setprio 2 / 3 for waves that hold a path of depth >= 6 / 10 / 16 / 24, always or
//  once the tickets are gone: -1 ... +0.6 %; graded 1 / 2 / 3 by the wave's deepest path (>= 4 / 10 / 20 and >= 2 / 5 / 10)
//  once the tickets are gone: -3 % on serial frames, +-1 % on 20 frames, and the per-trip log of tools/gpu_trace.py shows
//  the favoured waves' trips as long as before -- the instruction arbiter of this part does not act on it; no hashes for a smooth metal's unused draws in the lane-parallel builds:
//  +0.2 %)

// ---- scratch of the entry point that allocates nothing (pt_render_noalloc, include/pt.h) ------------------
// The path kernel's only memory beyond its inputs and outputs is its ticket counters.  pt_render takes them from a
// hipMalloc made at pt_create; callers that cannot allocate (HIP graph capture, serving loops) get them from this
// file-scope array instead: 4 KB, one copy per GPU the library runs on, held until the process ends.  In HOST code
// the name `scratch` is only a handle for HIP_SYMBOL: its device address is resolved below, once per device, and
// that pointer — never the name — is what kernels, memsets and copies are given.
static __device__ __align__(256) uint32_t scratch[PT_SCRATCH_BYTES / sizeof(uint32_t)];

namespace {
constexpr int PT_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
uint32_t* g_scratch_ptr[PT_MAX_DEVICES] = {};  // per device: each GPU holds its own copy at its own address
}  // namespace

// Called by pt_create with `device` current.
hipError_t pt_scratch_setup(int device, uint32_t** out) {
    if (device < 0 || device >= PT_MAX_DEVICES || !out) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<uint32_t*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

namespace {

constexpr uint32_t MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_EMISSIVE = 4;  // 3 = Glossy (the else branch)
constexpr float RAY_MIN_T = 1e-4f;  // Ray.cuh:38
// RAY_MAX_T = 1e4f (Ray.cuh:39) is folded into T_BEST_INIT below
constexpr float RAY_OFFSET = 0.001f;  // Material.cuh:64,115,173
constexpr float INV_PI = 0.318309886183790671538f;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float t) { return v3(a.x * t, a.y * t, a.z * t); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 u, V3 v) {
    return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// sin / cos of 2*pi*x: v_sin_f32 / v_cos_f32 take their argument in revolutions.
__device__ __forceinline__ float sin2pi(float x) { return __builtin_amdgcn_sinf(x); }
__device__ __forceinline__ float cos2pi(float x) { return __builtin_amdgcn_cosf(x); }
__device__ __forceinline__ V3 normalized(V3 a) {  // Vec3.cuh:49-61
    float l2 = dot(a, a);
    return l2 > 0.0f ? a * rsq(l2) : a;
}
__device__ __forceinline__ V3 xyz(float4 f) { return v3(f.x, f.y, f.z); }

// Counter RNG.  Two 32-bit multiplies per draw.
__device__ __forceinline__ uint32_t pt_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t pt_draw(uint32_t key, uint32_t slot, uint32_t dim) {
    return pt_hash(key + (slot * 8u + dim + 1u) * 0x9E3779B9U);
}
// (0,1], the range of curand_uniform which the reference's samplers are written for.
__device__ __forceinline__ float uni(uint32_t bits) { return (float)((bits >> 8) + 1u) * (1.0f / 16777216.0f); }

// ONB of Lambertian_t::scatter, Material.cuh:50-53:  a = |w.x| > 0.9 ? (0,1,0) : (1,0,0);  v = normalize(w x a);
// u = v x w.  With a an axis the first cross product is a permutation, (0, w.z, -w.y) or (-w.z, 0, w.x), its
// squared length two terms and one component of v zero, so the second cross product is two terms short: the
// same numbers in 17 instructions instead of 28 (the general formulas multiplied the zeros out).
__device__ __forceinline__ void onb(V3 w, V3& u, V3& v) {
#if PT_ONB == 1
    const bool ay = fabsf(w.x) > 0.9f;
    const float q = ay ? w.x : w.y;                    // the other non-zero component of w x a, up to sign
    const float l2 = fmaf(q, q, w.z * w.z);
    const float il = l2 > 0.0f ? rsq(l2) : 1.0f;       // normalized() leaves a zero vector as it is
    const float vz = (ay ? q : -q) * il, vo = w.z * il;
    v = ay ? v3(-vo, 0.0f, vz) : v3(0.0f, vo, vz);
    // u = v x w with v.x == 0 or v.y == 0
    u = ay ? v3(-(v.z * w.y), v.z * w.x - v.x * w.z, v.x * w.y)
           : v3(v.y * w.z - v.z * w.y, v.z * w.x, -(v.y * w.x));
#else
    V3 a = (fabsf(w.x) > 0.9f) ? v3(0, 1, 0) : v3(1, 0, 0);
    v = normalized(cross(w, a));
    u = cross(v, w);
#endif
}

// Scene operands of the intersection loops are read through the CONSTANT address space with a
// wave-uniform index: hipcc then emits scalar loads (s_load_dwordx16 through the scalar data
// cache) and the operands live in SGPRs.  Through a plain global pointer it emits one vector
// load plus an s_waitcnt vmcnt(0) per primitive, i.e. a serialized L2 round trip per test
// (profiles/r1_baseline: 0.51 ms per 720p frame).  The scene arena is written only by
// pt_upload_scene, never by a kernel, so the constant view is valid.
#define PT_CONST __attribute__((address_space(4)))
typedef float f4v __attribute__((ext_vector_type(4)));  // HIP's float4 class cannot be read from address space 4
typedef const PT_CONST f4v* cf4;
__device__ __forceinline__ cf4 as_const(const float4* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (cf4)p;
#pragma clang diagnostic pop
}

// The running closest hit.  The reference accepts a candidate when `t >= 1e-4 && t < tbest` (and t <= 1e4, which is
// folded into the initial tbest: the next float above 1e4, so `t < tbest` implies `t <= 1e4`).  Non-negative floats
// order like their bit patterns, so with k = bits(t) - bits(1e-4) in unsigned arithmetic that condition is exactly
// `k < bits(tbest) - bits(1e-4)`: a t below 1e-4 wraps to a huge k, a negative t, an infinity or a NaN has bits above
// every valid tbest.  The minimum is therefore carried biased (`tb`), one unsigned compare per candidate, and
// unbiased once after the loops.  The update is a real branch over two moves (an exec-masked region), not two
// selects: on gfx950 a v_cmp + s_and + 2 v_cndmask update costs 5.7 FMA issue slots, v_cmp + 2 v_mov under exec 4.2
// (tools/ubench/valu_rates.hip, profiles/r2_valu_rates.json); the empty asm keeps hipcc from if-converting it back.
constexpr float T_BEST_INIT = 10000.0009765625f;
constexpr uint32_t EPS_BITS = 0x38D1B717u;                 // bits of 1e-4f (RAY_MIN_T)
constexpr uint32_t TB_INIT = 0x461C4001u - EPS_BITS;        // bits(T_BEST_INIT) - EPS_BITS
__device__ __forceinline__ uint32_t t_key(float t) { return __float_as_uint(t) - EPS_BITS; }
__device__ __forceinline__ float t_of_key(uint32_t k) { return __uint_as_float(k + EPS_BITS); }
__device__ __forceinline__ void closer(bool also, float t, int id, uint32_t& tb, int& ib) {
    const uint32_t k = t_key(t);
#if PT_UPD == 2
    // (kept out of if-conversion by a scheduling barrier instead of an empty asm: hipcc then leaves out the branch that
    //  skips the two moves when no lane updates -- Cornell box -0.6 %, 4096 spheres +3 %)
    if (also && k < tb) { __builtin_amdgcn_sched_barrier(0); tb = k; ib = id; }
#elif PT_UPD == 1
    if (also && k < tb) { asm volatile(""); tb = k; ib = id; }
#else
    if (also && k < tb) { tb = k; ib = id; }
#endif
}

// Primitive id stored as float bits.  The argument is taken BY VALUE on purpose: hipcc 7.2 reads
// element 0 of the vector when __builtin_bit_cast is applied directly to an ext_vector element
// (`__builtin_bit_cast(uint32_t, e.y)` returned the bits of e.x).
__device__ __forceinline__ int as_id(float f) { return (int)__float_as_uint(f); }

template <bool COUNT>
__device__ __forceinline__ void sphere_test(const f4v s, int i, V3 o, V3 d, uint32_t& tb, int& ibest, uint32_t& n_roots) {
    // intersectSphere, raytracer_kernel.cu:44-78.  Most rays miss most spheres: the root block is
    // skipped by the whole wave when no lane has a real root (a branch-free form measured 10 %
    // slower on the Cornell box and 2x slower on 4096 spheres, profiles/r1_variants.json).
    const V3 oc = v3(o.x - s.x, o.y - s.y, o.z - s.z);
    const float half_b = dot(oc, d);
    const float c = fmaf(oc.x, oc.x, fmaf(oc.y, oc.y, fmaf(oc.z, oc.z, -s.w)));  // |oc|^2 - r^2 in three FMAs
    const float disc = half_b * half_b - c;
    if (disc >= 0.0f) {
        if (COUNT) n_roots++;
        const float sq = fsqrt(disc);
        const float r0 = -half_b - sq, r1 = sq - half_b;
        const float root = r0 >= RAY_MIN_T ? r0 : r1;  // r0 > 1e4 fails `< tbest` below, as r1 > r0 would
        closer(true, root, i, tb, ibest);
    }
}

// intersectPlane, raytracer_kernel.cu:80-103, general plane: pn = (unit normal, -(p.n));
// dot(p - o, n) is expanded to p.n - o.n, so t = -(o.n + w) / (n.d).
__device__ __forceinline__ void plane_test(const f4v pn, int id, V3 o, V3 d, uint32_t& tb, int& ibest) {
    const float denom = pn.x * d.x + pn.y * d.y + pn.z * d.z;
    const float q = fmaf(pn.x, o.x, fmaf(pn.y, o.y, fmaf(pn.z, o.z, pn.w)));
    const float t = q * rcp(-denom);
    closer(fabsf(denom) >= 1e-6f, t, id, tb, ibest);
}

// The same test for planes whose unit normal is +-e_k (every wall of the Cornell boxes): n.d =
// +-d_k and (p - o).n = +-(p_k - o_k), the signs cancel and t = (p_k - o_k) / d_k, with 1/d_k shared
// by all such planes of the ray (NaN where the reference rejects |n.d| < 1e-6).  Each axis has its
// own list (two planes per float4: p_k, bits(id), p_k, bits(id); NaN-padded), so the choice of
// form is made at upload, not by a branch chain in the loop.
__device__ __forceinline__ void axis_planes(cf4 list, int n, float o_k, float inv_dk, uint32_t& tb, int& ibest) {
    for (int i = 0; i < n; i += 2) {
        const f4v e = list[i >> 1];
        const float t0 = (e.x - o_k) * inv_dk, t1 = (e.z - o_k) * inv_dk;
        closer(true, t0, as_id(e.y), tb, ibest);
        closer(true, t1, as_id(e.w), tb, ibest);
    }
}

__device__ __forceinline__ void tri_test(const f4v r0, const f4v r1, const f4v r2, int id, V3 o, V3 d, uint32_t& tb, int& ibest) {
    // intersectTriangle (raytracer_kernel.cu:106-150) computes Moller-Trumbore's (t, u, v) from
    // the vertices on every test.  The same three numbers come out of an affine map fixed per
    // triangle (rows r0, r1, r2 = inverse of [e1 e2 n], translated by v0; built at upload in
    // double): height(x) = r2.x + r2.w vanishes on the plane, u = r0.x + r0.w, v = r1.x + r1.w.
    const float oz = fmaf(r2.x, o.x, fmaf(r2.y, o.y, fmaf(r2.z, o.z, r2.w)));
    const float dz = r2.x * d.x + r2.y * d.y + r2.z * d.z;
    const float t = oz * rcp(-dz);  // dz == 0 (parallel, :121): inf or NaN, rejected below
    const V3 h = v3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
    const float u = fmaf(r0.x, h.x, fmaf(r0.y, h.y, fmaf(r0.z, h.z, r0.w)));
    const float v = fmaf(r1.x, h.x, fmaf(r1.y, h.y, fmaf(r1.z, h.z, r1.w)));
    closer(u >= 0.0f && v >= 0.0f && u + v <= 1.0f, t, id, tb, ibest);
}

// Closest hit over spheres, planes, triangles in that order with a strict '<'
// (intersectScene, raytracer_kernel.cu:152-184).  Only (t, primitive id) are tracked; the hit
// record is built once for the winner.
template <bool COUNT, bool PEEL>
__device__ __forceinline__ void closest_hit(const DevSceneF& sc, V3 o, V3 d, float& t_out, int& ibest,
                                            uint32_t& n_roots) {
    uint32_t tbest = TB_INIT;  // biased, see closer()
    ibest = -1;
    const cf4 sph = as_const(sc.sph), pl = as_const(sc.pl), tri = as_const(sc.tri);
    const float qnan = __builtin_nanf("");
    const V3 inv_d = v3(fabsf(d.x) >= 1e-6f ? rcp(d.x) : qnan, fabsf(d.y) >= 1e-6f ? rcp(d.y) : qnan,
                        fabsf(d.z) >= 1e-6f ? rcp(d.z) : qnan);
    if (PEEL) {
    // whole batches of four without guards (three scalar instructions per sphere, and the scalar pipe is the kernel's
    // second bound), then the remainder behind wave-uniform scalar branches: a padding entry costs no vector work
    int i = 0;
    for (; i + 4 <= sc.ns; i += 4) {
        const f4v s0 = sph[i], s1 = sph[i + 1], s2 = sph[i + 2], s3 = sph[i + 3];
        sphere_test<COUNT>(s0, i, o, d, tbest, ibest, n_roots);
        sphere_test<COUNT>(s1, i + 1, o, d, tbest, ibest, n_roots);
        sphere_test<COUNT>(s2, i + 2, o, d, tbest, ibest, n_roots);
        sphere_test<COUNT>(s3, i + 3, o, d, tbest, ibest, n_roots);
    }
    if (i < sc.ns) {
        const f4v s0 = sph[i], s1 = sph[i + 1], s2 = sph[i + 2];
        sphere_test<COUNT>(s0, i, o, d, tbest, ibest, n_roots);
        if (i + 1 < sc.ns) sphere_test<COUNT>(s1, i + 1, o, d, tbest, ibest, n_roots);
        if (i + 2 < sc.ns) sphere_test<COUNT>(s2, i + 2, o, d, tbest, ibest, n_roots);
    }
    } else {
    for (int i = 0; i < sc.ns; i += 4) {
        const f4v s0 = sph[i], s1 = sph[i + 1], s2 = sph[i + 2], s3 = sph[i + 3];
        // the guards are wave-uniform scalar branches: a padding entry costs no vector work
        sphere_test<COUNT>(s0, i, o, d, tbest, ibest, n_roots);
        if (i + 1 < sc.ns) sphere_test<COUNT>(s1, i + 1, o, d, tbest, ibest, n_roots);
        if (i + 2 < sc.ns) sphere_test<COUNT>(s2, i + 2, o, d, tbest, ibest, n_roots);
        if (i + 3 < sc.ns) sphere_test<COUNT>(s3, i + 3, o, d, tbest, ibest, n_roots);
    }
    }
    axis_planes(as_const(sc.axp), sc.nax, o.x, inv_d.x, tbest, ibest);
    axis_planes(as_const(sc.axp) + sc.ax_off_y, sc.nay, o.y, inv_d.y, tbest, ibest);
    axis_planes(as_const(sc.axp) + sc.ax_off_z, sc.naz, o.z, inv_d.z, tbest, ibest);
    for (int i = 0; i < sc.npg; i += 4) {
        const f4v p0 = pl[2 * i], p1 = pl[2 * i + 2], p2 = pl[2 * i + 4], p3 = pl[2 * i + 6];
        const f4v q0 = pl[2 * i + 1], q1 = pl[2 * i + 3], q2 = pl[2 * i + 5], q3 = pl[2 * i + 7];  // .x = bits(id)
        plane_test(p0, as_id(q0.x), o, d, tbest, ibest);
        if (i + 1 < sc.npg) plane_test(p1, as_id(q1.x), o, d, tbest, ibest);
        if (i + 2 < sc.npg) plane_test(p2, as_id(q2.x), o, d, tbest, ibest);
        if (i + 3 < sc.npg) plane_test(p3, as_id(q3.x), o, d, tbest, ibest);
    }
    if (PEEL) {
    int k = 0;
    for (; k + 2 <= sc.nt; k += 2) {
        const f4v a0 = tri[3 * k], a1 = tri[3 * k + 1], a2 = tri[3 * k + 2];
        const f4v b0 = tri[3 * k + 3], b1 = tri[3 * k + 4], b2 = tri[3 * k + 5];
        tri_test(a0, a1, a2, sc.ns + sc.np + k, o, d, tbest, ibest);
        tri_test(b0, b1, b2, sc.ns + sc.np + k + 1, o, d, tbest, ibest);
    }
    if (k < sc.nt) {
        const f4v a0 = tri[3 * k], a1 = tri[3 * k + 1], a2 = tri[3 * k + 2];
        tri_test(a0, a1, a2, sc.ns + sc.np + k, o, d, tbest, ibest);
    }
    } else {
    for (int i = 0; i < sc.nt; i += 2) {
        const f4v a0 = tri[3 * i], a1 = tri[3 * i + 1], a2 = tri[3 * i + 2];
        const f4v b0 = tri[3 * i + 3], b1 = tri[3 * i + 4], b2 = tri[3 * i + 5];
        tri_test(a0, a1, a2, sc.ns + sc.np + i, o, d, tbest, ibest);
        if (i + 1 < sc.nt) tri_test(b0, b1, b2, sc.ns + sc.np + i + 1, o, d, tbest, ibest);
    }
    }
    t_out = t_of_key(tbest);
}

// ---- lane-parallel closest hit --------------------------------------------------------------
// closest_hit() spends one instruction stream on 64 rays x all primitives; when only a few lanes
// of the wave still trace (the end of a launch: paths of up to max_depth bounces, one or two per
// wave) it costs the same ~25 instructions per primitive for those few rays, and the launch ends
// when the slowest such wave does.  Here the roles are swapped: each ray in turn is broadcast to
// the whole wave (v_readlane), lane i tests primitives i, i+64, ..., and a DPP min-reduction plus
// one ballot find the winner.  Per ray that is ~120 instructions for a Cornell box instead of
// ~460, and N/64 tests instead of N for large scenes.  Ties between primitives at exactly equal
// t go to the lowest id here and to the first in list order (spheres, axis planes, general
// planes, triangles) in closest_hit(); both are measure-zero events.
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fminf(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fminf(v, dpp_f<0x141>(v));  // row_half_mirror
    v = fminf(v, dpp_f<0x140>(v));  // row_mirror: every lane now holds the minimum of its row of 16
    return fminf(fminf(readlane_f(v, 0), readlane_f(v, 16)), fminf(readlane_f(v, 32), readlane_f(v, 48)));
}

// The first 64 primitives of each kind are staged in LDS by the block at kernel start (as many as the scene has:
// 432 B for the Cornell box, at most 5 KB): a ds_read costs ~100 cycles where a global load that hits L2 costs ~500,
// and in this mode the wave has no other work to hide it behind.  Primitives beyond 64 of a kind come from global
// memory.  The image is dynamic LDS, laid out by the host (PathArgs::lds_*): spheres | planes | triangle rows.
struct FewLds {
    const float4* sph;
    const float4* pla;
    const float4* tri;
};
extern __shared__ float4 pt_lds[];

__device__ __forceinline__ f4v to_f4v(float4 a) { f4v r; r.x = a.x; r.y = a.y; r.z = a.z; r.w = a.w; return r; }


// element `index` of a scene array through a 32-bit byte offset from the uniform base (global_load ... v_off, s[base]):
// the 64-bit per-lane address of `p[index]` costs the lane-parallel loops a register
__device__ __forceinline__ float4 at32(const float4* p, uint32_t index) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p) + (size_t)(index * 16u));
}
// The running minimum of one lane in the lane-parallel mode: selects, not a branch (one ray, a handful of busy lanes: an
// exec-masked region costs its three scalar instructions and a taken branch, with nothing to skip).
__device__ __forceinline__ void closer_sel(bool also, float t, int id, uint32_t& tb, int& ib) {
    const uint32_t k = t_key(t);
    const bool take = also && k < tb;
    tb = take ? k : tb;
    ib = take ? id : ib;
}
template <bool COUNT, bool FAST>
__device__ __forceinline__ void closest_hit_few(const DevSceneF& sc, const FewLds& lds, uint64_t rays, int lane, V3 o, V3 d,
                                                float& t_out, int& id_out, uint32_t& n_roots) {
    // FAST (the builds without light sampling; with it the build is at its register limit and spills):
    // The lane's own primitive of each kind among the first 64: its five LDS rows are read together, one wait for all
    // (clamped indices: a lane beyond the count reads a row that exists and its result is masked out).  Read per trip,
    // not once per launch: held across the main loop they would cost a wave of occupancy.
    const bool has_s = lane < sc.ns, has_p = lane < sc.np, has_t = lane < sc.nt;
    const int is = min(lane, max(sc.ns, 1) - 1), ip = min(lane, max(sc.np, 1) - 1), it = min(lane, max(sc.nt, 1) - 1);
    const f4v S = to_f4v(FAST && sc.ns > 0 ? lds.sph[is] : make_float4(0, 0, 0, -1.0f));
    const f4v P = to_f4v(FAST && sc.np > 0 ? lds.pla[ip] : make_float4(0, 0, 0, 0));
    f4v T2 = {0, 0, 0, 0};   // (the triangle's other two rows are read per ray, after the sphere and the plane are done with
    if (FAST && sc.nt > 0) T2 = to_f4v(lds.tri[3 * it + 2]);   //  their registers: all five rows at once take the build past 80 VGPRs)
    while (rays != 0ull) {
        const int L = __builtin_ctzll(rays);
        rays &= rays - 1ull;
        const V3 O = v3(readlane_f(o.x, L), readlane_f(o.y, L), readlane_f(o.z, L));
        const V3 D = v3(readlane_f(d.x, L), readlane_f(d.y, L), readlane_f(d.z, L));
        uint32_t tb = TB_INIT;  // biased (closer()); valid keys order like the floats they stand for
        int ib = -1;
        if constexpr (FAST) {   // sphere_test without its branches
            const V3 oc = v3(O.x - S.x, O.y - S.y, O.z - S.z);
            const float half_b = dot(oc, D);
            const float c = fmaf(oc.x, oc.x, fmaf(oc.y, oc.y, fmaf(oc.z, oc.z, -S.w)));
            const float disc = half_b * half_b - c;
            const bool real = has_s && disc >= 0.0f;
            if (COUNT) n_roots += real ? 1u : 0u;
            const float sq = fsqrt(fmaxf(disc, 0.0f));
            const float r0 = -half_b - sq, r1 = sq - half_b;
            closer_sel(real, r0 >= RAY_MIN_T ? r0 : r1, lane, tb, ib);
        } else {
            // the first 64 of each kind from LDS (re-read per ray: keeping them in registers across the
            // main loop costs a wave of occupancy), the rest from global memory
            if (lane < sc.ns) sphere_test<COUNT>(to_f4v(lds.sph[lane]), lane, O, D, tb, ib, n_roots);
        }
        for (uint32_t i = 64u + (uint32_t)lane; i - (uint32_t)lane < (uint32_t)sc.ns; i += 64u)
            if (i < (uint32_t)sc.ns) sphere_test<COUNT>(to_f4v(at32(sc.sph, i)), (int)i, O, D, tb, ib, n_roots);
        if constexpr (FAST) {   // plane_test
            const float denom = P.x * D.x + P.y * D.y + P.z * D.z;
            const float q = fmaf(P.x, O.x, fmaf(P.y, O.y, fmaf(P.z, O.z, P.w)));
            closer_sel(has_p && fabsf(denom) >= 1e-6f, q * rcp(-denom), sc.ns + lane, tb, ib);
        } else {
            if (lane < sc.np) plane_test(to_f4v(lds.pla[lane]), sc.ns + lane, O, D, tb, ib);
        }
        for (uint32_t i = 64u + (uint32_t)lane; i - (uint32_t)lane < (uint32_t)sc.np; i += 64u)
            if (i < (uint32_t)sc.np) plane_test(to_f4v(at32(sc.pla, i)), sc.ns + (int)i, O, D, tb, ib);
        if constexpr (FAST) {
          if (sc.nt > 0) {   // tri_test (wave-uniform guard: a scene without triangles skips it)
            const float oz = fmaf(T2.x, O.x, fmaf(T2.y, O.y, fmaf(T2.z, O.z, T2.w)));
            const float dz = T2.x * D.x + T2.y * D.y + T2.z * D.z;
            const float t = oz * rcp(-dz);
            const V3 h = v3(O.x + D.x * t, O.y + D.y * t, O.z + D.z * t);
            const f4v T0 = to_f4v(lds.tri[3 * it]), T1 = to_f4v(lds.tri[3 * it + 1]);
            const float u = fmaf(T0.x, h.x, fmaf(T0.y, h.y, fmaf(T0.z, h.z, T0.w)));
            const float v = fmaf(T1.x, h.x, fmaf(T1.y, h.y, fmaf(T1.z, h.z, T1.w)));
            closer_sel(has_t && u >= 0.0f && v >= 0.0f && u + v <= 1.0f, t, sc.ns + sc.np + lane, tb, ib);
          }
        } else {
            if (lane < sc.nt)
                tri_test(to_f4v(lds.tri[3 * lane]), to_f4v(lds.tri[3 * lane + 1]), to_f4v(lds.tri[3 * lane + 2]), sc.ns + sc.np + lane, O, D, tb, ib);
        }
        for (uint32_t i = 64u + (uint32_t)lane; i - (uint32_t)lane < (uint32_t)sc.nt; i += 64u)
            if (i < (uint32_t)sc.nt)
                tri_test(to_f4v(at32(sc.tri, 3u * i)), to_f4v(at32(sc.tri, 3u * i + 1u)), to_f4v(at32(sc.tri, 3u * i + 2u)), sc.ns + sc.np + (int)i, O, D, tb, ib);
        const float tl = t_of_key(tb);
        const float tmin = wave_min(tl);  // never NaN: every test stores only values that passed `<`
        const int winner = __builtin_ctzll(__ballot(tl == tmin));
        const int idw = __builtin_amdgcn_readlane(ib, winner);
        if (lane == L) { t_out = tmin; id_out = idw; }
    }
}

// Solid-angle pdf of picking one of nl sphere lights uniformly and sampling its cone from P.
// Returns 0 when P is inside the light.  omc (= 1 - cos of the cone's half angle) / wc / dc2 are by-products
// for the sampler.
__device__ __forceinline__ float light_pdf(float4 lt, int nl, V3 P, float& omc, V3& wc, float& dc2) {
    wc = v3(lt.x - P.x, lt.y - P.y, lt.z - P.z);
    dc2 = dot(wc, wc);
    omc = 0.0f;
    if (!(dc2 > lt.w)) return 0.0f;
    const float s2 = lt.w * rcp(dc2);
    const float cosmax = fsqrt(fmaxf(0.0f, 1.0f - s2));
    omc = s2 * rcp(1.0f + cosmax);  // 1 - cosmax without cancellation
    return rcp(6.283185307179586f * omc * (float)nl);
}

__device__ __forceinline__ uint32_t pack_rgba8_cuda(float r, float g, float b) {  // Window.cuh:25-29
    const uint32_t R = (uint32_t)fminf(fmaxf(r * 255.0f, 0.0f), 255.0f);
    const uint32_t G = (uint32_t)fminf(fmaxf(g * 255.0f, 0.0f), 255.0f);
    const uint32_t B = (uint32_t)fminf(fmaxf(b * 255.0f, 0.0f), 255.0f);
    return (R << 16) | (G << 8) | B;
}

// One step of the reference's running mean, renderKernel :297-304: c * t + prev * (1 - t), t = 1 / (frame_count + 1).
// One function for the path kernel's flush and the fold pass, with the contraction spelled out, so that the serial
// launch and the multi-frame launch round alike.
__device__ __forceinline__ float mean_step(float c, float prev, float t) { return __builtin_fmaf(c, t, prev * (1.0f - t)); }

__device__ __forceinline__ uint32_t lane_prefix(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The grid is exactly the set of waves that is resident at once (occupancy x CUs).  Every wave
// draws work from a global ticket counter until the frame is exhausted, so no wave slot idles
// while pixels are still queued (a static split loses up to a whole scheduling round to
// quantisation: 8100 waves on 7168 slots).  The unit of work is a 4x4 pixel block (16 lanes'
// worth): a wave whose pool is empty draws as many units as it has idle lanes (at least
// units_per_grab), so tickets stay fine-grained when a pixel carries many samples and a wave
// still fills all 64 lanes with one atomic when it carries one.  No unit is ever owned
// implicitly: a block that the hardware admits late (the occupancy query can be one block per CU
// high, MI355X_MICROARCH.md §Residency) must not hold pixels hostage until the end.
// One counter word sustains only ~88 returning atomics per microsecond chip-wide
// (MI355X_MICROARCH.md, row `dequeue`), so the counter is sharded 8 ways on separate 128-B
// lines: shard = blockIdx % 8 (blocks b and b+8 share an XCD under round-robin dispatch) serves
// the 8x8 tiles congruent to it mod 8.  Two counter sets alternate between launches; each launch
// zeroes the set of the next one (ordered by the kernel boundary), so no memset is enqueued.
//
// Kernel arguments.  Everything the intersection loops and the shading read stays in SGPRs for the
// whole launch.  The camera (20 dwords) and the display-side arguments are needed only where a
// lane starts a sample or finishes a pixel; held across the loop they pushed the kernel past the
// 102 SGPRs a wave has and the spills (v_readlane in every loop preheader) cost more than
// re-reading them: those are fetched from the kernarg segment, through the scalar cache, inside
// the regeneration block (`cold()` hides the pointer from the optimiser so the loads stay there).
struct KArgs {
    DevSceneF sc;
    DevCamF cam;
    PathArgs a;
};
__device__ __forceinline__ const PT_CONST KArgs* cold() {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const PT_CONST KArgs* k = (const PT_CONST KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma clang diagnostic pop
    asm volatile("" : "+s"(k));
    return k;
}

// Register budget.  The light-sampling build needs 85 VGPRs when left alone (5 waves/SIMD); asked for 6
// waves/SIMD it fits 80 with two spilled registers and runs 5 % faster.  The plain build sits at 71 VGPRs
// (7 waves/SIMD) and is left unconstrained.
#ifndef PT_NEE_WAVES
#define PT_NEE_WAVES 6
#endif
// (the plain build with the lane-parallel mode: round 2's needed 71 VGPRs and was left alone (asked for 7 waves it
//  spilled 31 SGPRs and ran 9 % slower).  Round 3's reads the lane's LDS rows together and takes 80-81; its 106 SGPRs
//  admit 6 waves/SIMD in any case (MI355X_MICROARCH.md, residency), so it is budgeted for 6: 77 VGPRs, nothing spilled)
#ifndef PT_FEW_WAVES
#define PT_FEW_WAVES 6
#endif
// (the light-sampling build without the lane-parallel mode: 76 VGPRs left alone; asked for 7 waves/SIMD it fits 72 with
//  three spilled registers and runs 3.6 % faster)
#ifndef PT_NEE_PLAIN_WAVES
#define PT_NEE_PLAIN_WAVES 7
#endif
#define PT_WAVES_PER_EU ((NEE && !COUNT) ? (FEW ? PT_NEE_WAVES : PT_NEE_PLAIN_WAVES) : ((FEW && !COUNT) ? PT_FEW_WAVES : 1))
// Threads per workgroup of the path kernel.  One wave per workgroup: the waves of a launch share nothing
// but the 5-KB LDS image of the scene, and a wave's slot is free for the next launch the moment it ends
// instead of when the slowest of four does (measured against 256: +2.5 % at 720p / 1 spp with three frames
// in flight, +2 % serial, +1 % at 4K; 128 is in between, 512 loses 6-14 %).  The host sizes grids in
// 256-thread units; the launcher converts.
#ifndef PT_BLOCK
#define PT_BLOCK 64
#endif
// SGPR budget.  Left alone every build takes all 102 SGPRs (106 with VCC and the like), and 97-112 SGPRs admit 6 waves
// per SIMD whatever the VGPRs say (MI355X_MICROARCH.md, residency: floor(800 / (alloc + 16))).  The plain build without
// the lane-parallel mode -- 63 VGPRs, the build of every overlapped launch -- is capped at 96: 19 spilled SGPRs instead
// of 9, 7 waves per SIMD instead of 6, and +2.4 % (1 024 frames at 720p), +3.2 % (4K), +3 % (glass_stress), +3.5 % (150
// spheres), +0.5...1.7 % (the driver's 20 frames) against the uncapped build in the same gpurun call (round 3; round 2
// had measured +-3 % for a cap on ALL builds: the lane-parallel builds lose 1.5-2.5 % under it, a cap of 80 costs serial
// frames 15 %, and the light-sampling build without the mode already takes 94).
#ifndef PT_PLAIN_SGPRS
#define PT_PLAIN_SGPRS 96
#endif
#define PT_OCC __attribute__((amdgpu_waves_per_eu(PT_WAVES_PER_EU)))
// (amdgpu_num_sgpr wants a literal, not an expression of the template parameters: the plain build is an explicit
//  specialisation of the kernel with its own attribute, below; both forms inline the one body)

// FEW: the build with the lane-parallel closest hit.  Its mere presence costs the common loop 8 VGPRs and
// 3-4 % (Cornell box, several samples per launch, overlapped launches), and it pays where waves drain alone
// (serial launches) or where the scene is large, so the host picks the build per launch (pt_api.cpp).
template <bool NEE, bool COUNT, bool FEW>
__device__ __forceinline__ void pt_path_body(const KArgs& kargs) {
    const DevSceneF& sc = kargs.sc;
    const PathArgs& a = kargs.a;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t shard = blockIdx.x & 7u;
    const uint32_t shard_units = ((a.num_tiles + 7u - shard) >> 3) * 4u;  // 4x4 blocks of the tiles = shard (mod 8), per frame
    uint32_t unit0 = 0, pool_size = 0, pool_next = 0;  // wave-uniform; unit0 counts within the shard
    bool exhausted = false;                             // wave-uniform
    if (blockIdx.x == 0 && threadIdx.x < 8u) a.next_counter[threadIdx.x * 32u] = 0u;
    const FewLds few_lds = {pt_lds, pt_lds + a.lds_off_pla, pt_lds + a.lds_off_tri};
    if (threadIdx.x < 64u) {
        const int i = (int)threadIdx.x;
        if (FEW) {
            if (i < sc.ns) pt_lds[i] = sc.sph[i];
            if (i < sc.np) pt_lds[a.lds_off_pla + i] = sc.pla[i];
            if (i < sc.nt) { float4* t = pt_lds + a.lds_off_tri + 3 * i; t[0] = sc.tri[3 * i]; t[1] = sc.tri[3 * i + 1]; t[2] = sc.tri[3 * i + 2]; }
        }
    }
    __syncthreads();

    // lane state
    // The lane's CURRENT pixel (sum, prev, samples still to finish) and its NEXT camera ray, prepared ahead of time
    // (see the prepare step at the end of the loop body).
    bool cur_valid = false;          // owns a pixel whose sum is not flushed yet
    uint32_t pixel = 0, cur_left = 0;  // pixel: index into a.accum (frame * plane_stride + y * width + x)
    bool nx_valid = false;           // a prepared ray is waiting
    uint32_t nx_pxy = 0;             // pixel of the prepared ray: x | y << xw | frame << jo (PathArgs)
    uint32_t nx_si = 0xFFFFFFFFu;    // samples of that pixel generated so far (>= count: the lane needs a new pixel)
    uint32_t nx_key = 0;
    V3 nx_d = v3(0, 0, 1);
    bool alive = false;      // a path is in flight
    V3 sum = v3(0, 0, 0);    // radiance sum of this pixel in this launch
    float4 prev = make_float4(0, 0, 0, 0);  // the pixel's accumulator, fetched when the pixel is taken
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1), beta = v3(1, 1, 1), L = v3(0, 0, 0);
    uint32_t key = 0, depth = 0;
    // NEE state
    bool shadow = false, cont = false;
    V3 pend_o = v3(0, 0, 0), pend_d = v3(0, 0, 1), contrib = v3(0, 0, 0);
    int expect = -1;
    float prev_pdf_b = -1.0f;
    // counters (instrumented build only)
    uint32_t c_path_rays = 0;   // path rays of the lane's current sample (shadow rays apart): the histogram's bin
    uint32_t c_samples = 0, c_rays = 0, c_hits = 0, c_roots = 0, c_b0 = 0, c_b1 = 0, c_b2 = 0, c_b3 = 0, c_b4 = 0,
             c_diff = 0, c_nee = 0, c_shadow = 0;
    const bool need_prev = a.overwrite == 0u && (a.acc_mode == 0u || a.s0 > 0u);
#ifdef PT_TRACE
    // timeline of this wave (100 MHz constant clock): start, frame exhausted, first lane-parallel trip, end
    uint64_t tr_start = wall_clock64(), tr_exh = 0, tr_few = 0;
    uint32_t tr_trips = 0, tr_few_trips = 0, tr_few_rays = 0, tr_lane_trips = 0;
    // shader-clock cycles this wave spent per section (s_memtime stamps; the stamps themselves cost ~10 %)
    uint64_t tc_regen = 0, tc_isect = 0, tc_shade = 0, tc_mark = clock64();
    uint64_t tf_regen = 0, tf_isect = 0, tf_shade = 0;  // the same, trips of the lane-parallel mode only
    bool tr_in_few = false;
#define PT_SECTION(acc) { const uint64_t now_ = clock64(); tc##acc += now_ - tc_mark; if (tr_in_few) tf##acc += now_ - tc_mark; tc_mark = now_; }
#else
#define PT_SECTION(acc)
#endif

    // Loop order: trace + shade first, regeneration last (the first trip finds no lane alive and goes
    // straight to the regeneration steps).  With regeneration first hipcc kept the lane state in two
    // register sets and copied one to the other on every trip (~45 v_mov, 78 VGPRs; the light-sampling
    // build 96 VGPRs with spills); in this order it needs 71 and 85 and 40-120 fewer instructions.
    for (;;) {
        {   // (no `if any lane is alive` around this: with none, every region below is skipped by its own
            //  mask, and one nesting level fewer is worth +0.5 %)
        // ---- (4) one ray: 64 rays per instruction stream, or lane-parallel when few lanes trace
        float t_hit = T_BEST_INIT; int id_hit = -1;
        const uint64_t alive_mask = __ballot(alive);
#ifdef PT_TRACE
        if (a.trace && lane == 0) {   // the trip's start, in the ring of the wave's last 256 trips
            uint32_t* ring = (uint32_t*)(a.trace + (size_t)(blockIdx.x * (PT_BLOCK / 64u) + (threadIdx.x >> 6)) * PT_TRACE_WORDS + 16u);
            ring[tr_trips & 255u] = ((uint32_t)(wall_clock64() - tr_start) << 7) | (uint32_t)__popcll(alive_mask);
        }
        tr_trips++; tr_lane_trips += (uint32_t)__popcll(alive_mask);
        tr_in_few = (uint32_t)__popcll(alive_mask) <= a.few_max;
        if (tr_in_few) { if (!tr_few) tr_few = wall_clock64(); tr_few_trips++; tr_few_rays += (uint32_t)__popcll(alive_mask); }
#endif
        if (FEW && (uint32_t)__popcll(alive_mask) <= a.few_max) {  // wave-uniform: all lanes take part
            closest_hit_few<COUNT, PT_FEWFAST != 0 && !NEE>(sc, few_lds, alive_mask, (int)lane, o, d, t_hit, id_hit, c_roots);
        } else if (alive) {
            closest_hit<COUNT, PT_PEEL != 0 && !NEE>(sc, o, d, t_hit, id_hit, c_roots);  // (light-sampling builds: the peeled loops cost 9 more spilled SGPRs, -1.5 %)
        }
        PT_SECTION(_isect)
        {
            // (miss and hit are two regions side by side, not an if/else inside `if (alive)`: one nesting
            //  level fewer for hipcc to copy the lane state around)
            const float t = t_hit; const int id = id_hit;
            if (COUNT) { if (alive) { c_rays++; if (id >= 0) c_hits++; if (!(NEE && shadow)) c_path_rays++; } }
            bool end_path = false;
            if (alive && id < 0) {
                // miss: black background (:208); an escaping shadow ray reached no light
                if (NEE && shadow) { shadow = false; if (cont) { o = pend_o; d = pend_d; } else end_path = true; }
                else end_path = true;
            }
            if (alive && id >= 0) {
                // ---- (5) hit record of the winner (:66-71, :95-100, :142-147; Intersection.cuh:29-32):
                //          three independent 16-B gathers from one 64-B line
                // (round 3: the same three rows staged in LDS for scenes of <= 64 primitives -- within +-1 % in every
                //  workload, serial launches included; the L1 serves 19 hot lines as well: dropped)
                const float4 g0 = sc.rec[4 * id], g1 = sc.rec[4 * id + 1], g2 = sc.rec[4 * id + 2];
                const V3 p = o + d * t;
                V3 n = id < sc.ns ? (p - xyz(g0)) * g0.w : xyz(g0);
                const bool front = dot(d, n) < 0.0f;
                if (!front) n = -n;
                const uint32_t mtype = __float_as_uint(g1.w);
                const V3 albedo = xyz(g1);
                const int light = (int)__float_as_uint(g2.w);

                if (NEE && shadow) {
                    // ---- shadow ray: the light is visible iff it is the closest hit, seen from outside
                    if (id == expect && front) L = L + contrib;
                    shadow = false;
                    if (cont) { o = pend_o; d = pend_d; } else end_path = true;
                } else {
                    // ---- final_color += throughput * emitted (:211; Emissive_t::emitted, Material.cuh:254-258)
                    if (mtype == MAT_EMISSIVE) {
                        if (front) {
                            V3 Le = (a.light_colour && light >= 0) ? xyz(sc.lights[3 * light + 2]) : xyz(sc.rec[4 * id + 3]);
                            float w = 1.0f;
                            if (NEE && prev_pdf_b >= 0.0f && light >= 0) {
                                const V3 P = o - d * RAY_OFFSET;  // apex = the previous hit point
                                float cm, dc2; V3 wc;
                                const float pl = light_pdf(sc.lights[3 * light], sc.nl, P, cm, wc, dc2);
                                w = prev_pdf_b * rcp(prev_pdf_b + pl);
                            }
                            L = L + beta * Le * w;
                        }
                        if (COUNT) c_b4++;
                        end_path = true;  // Emissive_t::scatter returns false
                    } else {
                        // ---- scatter.  The four BSDFs share what they have in common, so a wave that
                        // holds all of them (every wave, after the first bounce) runs it once and not
                        // four times: the two uniform draws, the mirror direction, the final
                        // normalisation (Ray_t's constructor, Ray.cuh:43-53), the origin offset and the
                        // throughput product.  Per lane the arithmetic is what the separate
                        // Material.cuh scatter functions do, in the same order.
                        const uint32_t slot = depth + 1u;
                        const float u0 = uni(pt_draw(key, slot, 0));
                        const float u1 = uni(pt_draw(key, slot, 1));
                        const V3 refl = d - n * (2.0f * dot(d, n));
                        V3 dir = refl;                // un-normalised scattered direction
                        V3 tint = albedo;             // throughput factor
                        float eps = RAY_OFFSET;       // Material.cuh:64,115,173; Glossy_t uses none (:204)
                        bool do_shadow = false;
                        if (mtype == MAT_LAMBERTIAN) {
                            if (COUNT) c_b0++;
                            if (NEE && sc.nl > 0) {
                                // next-event estimation with balance-heuristic MIS (north_star addition)
                                if (COUNT) c_nee++;
                                const uint32_t j = __umulhi(pt_draw(key, slot, 4), (uint32_t)sc.nl);
                                const float v1 = uni(pt_draw(key, slot, 5)), v2 = uni(pt_draw(key, slot, 6));
                                float omc, dc2; V3 wc;
                                const float pl = light_pdf(sc.lights[3 * j], sc.nl, p, omc, wc, dc2);
                                if (pl > 0.0f) {
                                    // uniform in the cone, written in the small quantity 1 - cos(theta) itself: a light 50
                                    // units away subtends 1 - cosmax ~ 3e-5, which `1 - cosmax` and `1 - cos^2` would keep
                                    // to three digits
                                    const float omct = v1 * omc;
                                    const float cost = 1.0f - omct;
                                    const float sint = fsqrt(omct * (2.0f - omct));
                                    const V3 w = wc * rsq(dc2);
                                    V3 uu, vv; onb(w, uu, vv);
                                    const V3 ldir = normalized(uu * (cos2pi(v2) * sint) + vv * (sin2pi(v2) * sint) + w * cost);
                                    const float cosN = dot(n, ldir);
                                    if (cosN > 0.0f) {
                                        const float pb = cosN * INV_PI;
                                        const float wgt = pb * rcp(pl + pb);
                                        const float4 l1 = sc.lights[3 * j + 1];
                                        const V3 Ls = a.light_colour ? xyz(sc.lights[3 * j + 2]) : xyz(l1);
                                        contrib = beta * albedo * Ls * wgt;
                                        expect = (int)__float_as_uint(l1.w);
                                        do_shadow = true;
                                        pend_o = p + ldir * RAY_OFFSET;  // shadow ray, same offset rule as a scattered ray
                                        pend_d = ldir;
                                    }
                                }
                            }
                            // Lambertian_t::scatter, Material.cuh:43-70: cosine-weighted direction about n
                            V3 uu, vv; onb(n, uu, vv);
                            const float r2s = fsqrt(u1);
                            dir = uu * (cos2pi(u0) * r2s) + vv * (sin2pi(u0) * r2s) + n * fsqrt(1.0f - u1);
                        } else if (mtype == MAT_DIELECTRIC) {
                            // Dielectric_t::scatter, Material.cuh:149-175
                            if (COUNT) c_b2++;
                            const float ior = g2.z;
                            const float eta = front ? rcp(ior) : ior;
                            const float cos_theta = fminf(-dot(d, n), 1.0f);
                            const float sin_theta = fsqrt(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
                            const bool cannot = eta * sin_theta > 1.0f;
                            float r0 = (1.0f - eta) * rcp(1.0f + eta); r0 *= r0;
                            const float m1 = 1.0f - cos_theta, m2 = m1 * m1;
                            const float sch = r0 + (1.0f - r0) * (m2 * m2 * m1);  // powf(1-cos, 5), :143
                            if (!(cannot || sch > u0)) {  // refract, :133-138 (else: the mirror direction)
                                const V3 perp = (d + n * cos_theta) * eta;
                                const V3 par = n * -fsqrt(fabsf(1.0f - dot(perp, perp)));
                                dir = perp + par;
                            }
                            tint = v3(1, 1, 1);
                        } else {
                            const float rough = g2.x;  // clamped to <= 1 for Metal at upload (Material.cuh:94)
                            V3 pv;
                            if (mtype == MAT_METAL) {
                                // Metal_t::scatter, Material.cuh:97-121: the perturbation, and its third draw, only for a
                                // rough metal (:107; the palette's two metals are mirrors: refl + pv * 0 == refl)
                                if (COUNT) c_b1++;
                                pv = v3(0, 0, 0);
                                if (rough > 0.0f) pv = v3(u0 * 2.0f - 1.0f, u1 * 2.0f - 1.0f, uni(pt_draw(key, slot, 2)) * 2.0f - 1.0f);
                            } else {
                                // Glossy_t::scatter, Material.cuh:191-213; Random_t::randomUnitVector, Random.cuh:36-45
                                if (COUNT) c_b3++;
                                const float z = -1.0f + 2.0f * u0;
                                const float r = fsqrt(fmaxf(0.0f, 1.0f - z * z));
                                pv = normalized(v3(r * cos2pi(u1), r * sin2pi(u1), z));
                                const float m = g2.y;
                                const V3 spec = m > 0.5f ? albedo : v3(1, 1, 1);
                                tint = v3(albedo.x * (1.0f - m) + spec.x * m, albedo.y * (1.0f - m) + spec.y * m,
                                          albedo.z * (1.0f - m) + spec.z * m);
                                eps = 0.0f;
                            }
                            dir = refl + pv * rough;
                        }
                        dir = normalized(dir);
                        float cosine = dot(n, dir);
                        if (mtype > MAT_DIELECTRIC && cosine < 0.0f) { dir = normalized(refl); cosine = dot(n, dir); }  // Glossy, :199-201
                        beta = beta * tint;  // throughput *= albedo * scatteringPdf / pdf (:229-230; the ratio is 1)
                        bool go = true;
                        if (mtype == MAT_LAMBERTIAN) {
                            go = cosine > 0.0f;  // pdf > 0 (the reference's 0*inf NaN case ends the path)
                            if (COUNT) { if (go) c_diff++; }
                            if (go && depth > a.rr_depth) {  // :233-239
                                const float pr = fmaxf(beta.x, fmaxf(beta.y, beta.z));
                                if (uni(pt_draw(key, slot, 3)) > pr) go = false;
                                else beta = beta * rcp(pr);
                            }
                        } else if (mtype == MAT_METAL) {
                            go = cosine > 0.0f;  // Metal_t::scatter returns dot(scattered, normal) > 0
                        }
                        o = p + dir * eps;
                        d = dir;
                        if (NEE) {
                            prev_pdf_b = mtype == MAT_LAMBERTIAN ? cosine * INV_PI : -1.0f;
                            if (do_shadow) {
                                // the shadow ray goes first; the scattered ray waits in pend_*
                                if (COUNT) c_shadow++;
                                const V3 so = pend_o, sd = pend_d;
                                pend_o = o; pend_d = d;
                                o = so; d = sd;
                                shadow = true;
                                cont = go && (depth + 1u < a.max_depth);
                                go = true;  // the shadow ray is traced regardless
                            }
                        }
                        if (!go) end_path = true;
                    }
                    // for (depth = 0; depth < max_depth; depth++), :206
                    depth++;
                    if (!(NEE && shadow) && depth >= a.max_depth) end_path = true;
                }
            }
            if (end_path) {
                sum = sum + L; alive = false; cur_left--;
                // (64 copies of the histogram, one per blockIdx % 64, summed by the host: one copy's 64 hot words serialise the atomics)
                if (COUNT) { if (a.hist) atomicAdd((unsigned long long*)&a.hist[(blockIdx.x & 63u) * 64u + min(c_path_rays, 63u)], 1ull); }
            }
        }
        PT_SECTION(_shade)
        }
        // ---- Regeneration, double-buffered.  A lane whose path ended wants a flush (rarely), and a new camera ray
        // (always): ~150 instructions that used to run on every trip at the 10-15 % of the lanes that had just
        // ended.  Now every lane holds its NEXT camera ray ready, so a path end costs a handful of moves (steps 1
        // and 3), and the expensive part -- pulling pixels from the pool and generating camera rays (step 2) --
        // runs for all lanes whose next ray is missing at once, when a.regen_min of them have gathered (or when a
        // lane would otherwise sit idle).  Nobody waits for the batch, so unlike a thresholded regeneration it costs
        // no idle lane-trips.  Samples of a pixel are still traced one after the other by one lane, in order: the
        // sums are bit-identical to the single-buffered loop's.
        const uint64_t dead_mask = __ballot(!alive);
        // (builds with the lane-parallel mode -- serial launches, where waves drain long paths alone: the block is skipped
        //  outright unless there is something to flush or to take (a dead lane that owns a pixel or holds a prepared ray)
        //  or something to prepare (tickets remain, or some lane's pixel has samples left to generate); in launches of
        //  many frames the test costs more than the few trips it shortens, -0.5 ... -1 %)
        const bool regen_work = !(FEW && PT_SKIPREGEN != 0) || !exhausted ||
                                __ballot((!alive && (cur_valid || nx_valid)) || (!nx_valid && nx_si < a.count)) != 0ull;
        if (dead_mask != 0ull && regen_work) {
        const PT_CONST KArgs* kc = cold();
        // ---- (1) a pixel whose last sample just ended is flushed: one 16-B store (its 16-B load was issued when the
        //          pixel was taken, so the latency is long gone)
        if (!alive && cur_valid && cur_left == 0u) {
            float4 acc;
            if (a.acc_mode == 0u) {  // PT_ACC_SUM (prev is zero in a launch that writes its own sums whole)
                acc = make_float4(prev.x + sum.x, prev.y + sum.y, prev.z + sum.z, prev.w + (float)a.count);
            } else {  // PT_ACC_RUNNING_MEAN, renderKernel :297-304 (count == 1, s0 = frame_count)
                acc = make_float4(sum.x, sum.y, sum.z, 1.0f);   // (a launch of several frames writes the frame's own value: the
                if (a.s0 > 0u && a.overwrite == 0u) {           //  fold pass applies the recurrence, pt_fold_kernel)
                    const float t = kc->a.mean_t;
                    acc.x = mean_step(sum.x, prev.x, t);
                    acc.y = mean_step(sum.y, prev.y, t);
                    acc.z = mean_step(sum.z, prev.z, t);
                }
            }
            a.accum[pixel] = acc;
            uint32_t* const rgba = kc->a.rgba;
            if (rgba) {  // fused convertToRGBAKernel (launches that overlap others display in a pass of their own)
                // every pixel normally holds the same count: its reciprocal comes from the host (same IEEE divide)
                const float inv = (a.acc_mode != 0u) ? 1.0f : (acc.w == kc->a.total_w ? kc->a.inv_total_w : 1.0f / acc.w);
                rgba[pixel] = pack_rgba8_cuda(acc.x * inv, acc.y * inv, acc.z * inv);
            }
            cur_valid = false;
        }
        // ---- (2) prepare next rays, in a batch
        // A lane wants one while it has none and either its pixel has samples left to generate or pixels remain.
        // Taking the NEXT PIXEL ahead of time withholds it from lanes that may fall idle at the end of the frame,
        // so a lane that still traces does that only in launches with many pixels per lane (a.prefetch_pixels).
        const bool want_next = !nx_valid && (nx_si < a.count || (!exhausted && (!alive || a.prefetch_pixels != 0u)));
        const uint32_t n_want = (uint32_t)__popcll(__ballot(want_next));
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(!alive && want_next));  // lanes that sit idle until prepared
        if (n_want >= a.regen_min || n_idle >= a.idle_min || (n_idle != 0u && dead_mask == ~0ull)) {
            // ---- (2a) lanes that have generated every sample of their pixel pull the next pixel from the wave's
            //           pool; an empty pool is refilled with one ticket from the global counter
            const bool want = want_next && nx_si >= a.count;
            const uint64_t pull_mask = __ballot(want);
            if (pull_mask != 0ull && !exhausted) {
                if (pool_next >= pool_size) {
                    const uint32_t total_units = shard_units * kc->a.frames;  // a launch of several frames: frame-major
                    // (round 3: tickets that shrink towards the end of the ticket space, to a fair share of what is left,
                    //  made the waves run out of work within 125 us of each other instead of 290 -- and the launch no
                    //  shorter: it ends when its longest path ends, 190-260 us after the last ticket.  Dropped.)
                    const uint32_t need = max(a.units_per_grab, ((uint32_t)__popcll(pull_mask) + 15u) >> 4);
                    uint32_t ticket = 0;
                    if (lane == 0) ticket = atomicAdd(a.work_counter + shard * 32u, need);
                    ticket = __builtin_amdgcn_readfirstlane(ticket);
                    if (ticket >= total_units) {
                        exhausted = true;
#ifdef PT_TRACE
                        tr_exh = wall_clock64();
#endif
                    } else {
                        unit0 = ticket;
                        pool_size = min(need, total_units - ticket) * 16u;
                        pool_next = 0;
                    }
                }
                if (!exhausted) {
                    const uint32_t cand = pool_next + lane_prefix(pull_mask);
                    if (want && cand < pool_size) {
                        uint32_t unit = unit0 + (cand >> 4), frame = 0u;
                        const uint32_t li = cand & 15u;
                        // (integer divisions by wave-uniform numbers: float quotient + one fix-up instead of 25 instructions;
                        //  the host keeps both dividends below 2^23)
                        if (kc->a.frames > 1u) {  // wave-uniform: which frame of the launch, which unit of that frame
                            frame = (uint32_t)(((float)unit + 0.5f) * (1.0f / (float)shard_units));
                            int u = (int)(unit - frame * shard_units);
                            if (u < 0) { u += (int)shard_units; frame--; }
                            if (u >= (int)shard_units) { u -= (int)shard_units; frame++; }
                            unit = (uint32_t)u;
                        }
                        const uint32_t q = unit & 3u;
                        const uint32_t tile = (unit >> 2) * 8u + shard;
                        uint32_t ty = (uint32_t)(((float)tile + 0.5f) * a.inv_tiles_x);
                        int tx = (int)(tile - ty * a.tiles_x);
                        if (tx < 0) { tx += (int)a.tiles_x; ty--; }
                        if (tx >= (int)a.tiles_x) { tx -= (int)a.tiles_x; ty++; }
                        // (units of 8 x 2 pixels -- whole 128-B lines of the plane instead of 64-B halves -- write 391 MB instead of
                        //  404 per 20-frame launch against 295 algorithmic, +-1 %: the excess is lanes of one line finishing
                        //  their paths many microseconds apart, not the unit's shape; round 3)
                        const uint32_t px = (uint32_t)tx * 8u + (q & 1u) * 4u + (li & 3u);
                        const uint32_t py = ty * 8u + (q >> 1) * 4u + (li >> 2);
                        if (px < a.width && py < a.height) {
                            nx_pxy = px | (py << kc->a.xw) | (frame << kc->a.jo);
                            nx_si = 0u;
                        }
                    }
                    pool_next = min(pool_next + (uint32_t)__popcll(pull_mask), pool_size);
                }
            }
            // ---- (2b) the camera ray of the pixel's next sample
            if (want_next && nx_si < a.count) {
                const uint32_t nx_x = __builtin_amdgcn_ubfe(nx_pxy, 0u, kc->a.xw), nx_y = __builtin_amdgcn_ubfe(nx_pxy, kc->a.xw, kc->a.yw);
                const uint32_t pixkey = pt_hash(a.seed_hash + (nx_y * a.width + nx_x));
                // frame f of the launch holds samples [s0 + f * frame_stride, s0 + f * frame_stride + count)
                nx_key = pt_hash(pixkey + (a.s0 + __builtin_amdgcn_ubfe(nx_pxy, kc->a.jo, kc->a.jw) * kc->a.frame_stride + nx_si));
                nx_si++;
                // renderKernel :286-291 + generateCameraRay :245-265:
                //   (2*(x + xi)/W - 1) * half_w  ==  (x + xi) * (2*half_w/W) - half_w
                const PT_CONST DevCamF& cam = kc->cam;
                const float xo = ((float)nx_x + uni(pt_draw(nx_key, 0, 0))) * cam.sx - cam.half_w;
                const float yo = ((float)nx_y + uni(pt_draw(nx_key, 0, 1))) * cam.sy - cam.half_h;
                nx_d = v3(cam.fx + cam.rx * xo + cam.ux * yo, cam.fy + cam.ry * xo + cam.uy * yo, cam.fz + cam.rz * xo + cam.uz * yo);
                if (!(cam.lens_radius > 0.0f)) nx_d = normalized(nx_d);  // (with a lens the direction is finished in step 3)
                nx_valid = true;
                if (COUNT) c_samples++;
            }
        }
        // ---- (3) lanes without a path take their prepared ray; the first ray of a pixel also takes the pixel
        if (!alive && nx_valid) {
            const PT_CONST DevCamF& cam = kc->cam;
            if (!cur_valid) {
                pixel = __builtin_amdgcn_ubfe(nx_pxy, kc->a.jo, kc->a.jw) * kc->a.plane_stride + __builtin_amdgcn_ubfe(nx_pxy, kc->a.xw, kc->a.yw) * a.width +
                        __builtin_amdgcn_ubfe(nx_pxy, 0u, kc->a.xw);
                if (need_prev) prev = a.accum[pixel];
                sum = v3(0, 0, 0);
                cur_left = a.count;
                cur_valid = true;
            }
            key = nx_key;
            o = v3(cam.ox, cam.oy, cam.oz);
            d = nx_d;
            const float lens_radius = cam.lens_radius;
            if (lens_radius > 0.0f) {  // wave-uniform
                // Depth of field, Camera_t::getRayDevice (Camera.cu:31-47): the virtual screen is at focus_dist
                // (Camera.cu:15-19), the origin moves on the lens disk (randomInUnitDisk, :25-29) in the
                // camera's right/up plane and the ray aims at the point the pinhole ray reaches on that screen.
                const float r = fsqrt(uni(pt_draw(key, 0, 2))) * lens_radius;
                const float th = uni(pt_draw(key, 0, 3));
                const float lx = r * cos2pi(th), ly = r * sin2pi(th);
                const V3 org = v3(o.x + cam.rx * lx + cam.ux * ly, o.y + cam.ry * lx + cam.uy * ly, o.z + cam.rz * lx + cam.uz * ly);
                const float fd = cam.focus_dist;
                d = normalized(v3(o.x + d.x * fd - org.x, o.y + d.y * fd - org.y, o.z + d.z * fd - org.z));
                o = org;
            }
            beta = v3(1, 1, 1);
            L = v3(0, 0, 0);
            depth = 0;
            alive = true;
            nx_valid = false;
            if (COUNT) c_path_rays = 0;
            if (NEE) { shadow = false; prev_pdf_b = -1.0f; }
        }
        }  // some lane has no path
        PT_SECTION(_regen)
        // done when nothing is in flight, prepared or unflushed, and the frame is exhausted
        if (__ballot(alive || nx_valid || cur_valid) == 0ull && exhausted) break;
    }

#ifdef PT_TRACE
    if (a.trace && lane == 0) {
        uint64_t* tr = a.trace + (size_t)(blockIdx.x * (PT_BLOCK / 64u) + (threadIdx.x >> 6)) * PT_TRACE_WORDS;
        tr[0] = tr_start; tr[1] = tr_exh; tr[2] = tr_few; tr[3] = wall_clock64();
        tr[4] = tr_trips; tr[5] = tr_few_trips; tr[6] = tr_few_rays; tr[7] = tr_lane_trips;
        tr[8] = tc_regen; tr[9] = tc_isect; tr[10] = tc_shade; tr[11] = tf_regen; tr[12] = tf_isect; tr[13] = tf_shade;
    }
#endif
    if (COUNT && a.counters) {
        uint32_t c[12] = {c_samples, c_rays, c_hits, c_roots, c_b0, c_b1, c_b2, c_b3, c_b4, c_diff, c_nee, c_shadow};
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            uint32_t v = c[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0 && v) atomicAdd((unsigned long long*)&a.counters[k], (unsigned long long)v);
        }
    }
}

template <bool NEE, bool COUNT, bool FEW>
__global__ __launch_bounds__(PT_BLOCK) PT_OCC void pt_path_kernel(const KArgs kargs) { pt_path_body<NEE, COUNT, FEW>(kargs); }
#if PT_PLAIN_SGPRS > 0
template <>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(1), amdgpu_num_sgpr(PT_PLAIN_SGPRS))) void
pt_path_kernel<false, false, false>(const KArgs kargs) { pt_path_body<false, false, false>(kargs); }
#endif

// ------------------------------------------------------------------------------------------
// Resolve pass: accumulation buffer -> display buffer (convertToRGBAKernel, Window.cuh:16-30;
// the CPU app's accumulate + gamma loop, main.cpp:223-237; Camera::saveImage, Camera.h:162-180).
// One pixel per lane: a 16-B coalesced load, a 4-B (or 16-B / 3-B) store.
__global__ __launch_bounds__(256) void pt_resolve_kernel(const float4* __restrict__ accum, void* __restrict__ out,
                                                         uint32_t npix, uint32_t acc_mode, uint32_t mode) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= npix) return;
    const float4 acc = accum[p];
    float inv = 1.0f;
    if (acc_mode == 0u) inv = acc.w > 0.0f ? 1.0f / acc.w : 0.0f;
    const float r = acc.x * inv, g = acc.y * inv, b = acc.z * inv;
    if (mode == 2u) {  // PT_RESOLVE_FLOAT_MEAN
        ((float4*)out)[p] = make_float4(r, g, b, 1.0f);
    } else if (mode == 0u) {  // PT_RESOLVE_RGBA8_CUDA
        ((uint32_t*)out)[p] = pack_rgba8_cuda(r, g, b);
    } else if (mode == 1u) {  // PT_RESOLVE_RGBA8_CPU (main.cpp:229-234 + WindowX11.h:130-139)
        // std::pow(std::clamp(c, 0.0, 1.0), 1.0/2.2) in double as the reference evaluates it, then the window's
        // truncating *255 pack.  The pass is HBM-bound: three double pows per pixel ride along.
        uint32_t c8[3];
        const float c[3] = {r, g, b};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = pow(fmin(fmax((double)c[k], 0.0), 1.0), 1.0 / 2.2);
            c8[k] = (uint32_t)(uint8_t)(v * 255.0);
        }
        ((uint32_t*)out)[p] = (c8[0] << 16) | (c8[1] << 8) | c8[2];
    } else {  // PT_RESOLVE_RGB8_PPM: ColorDBL::gamma(2.2), clamp, *255 + 0.5 truncated (Camera.h:172-176, Color.h:45-51,86-91)
        uint8_t* o = (uint8_t*)out + 3ull * p;
        const float c[3] = {r, g, b};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double gm = pow((double)c[k], 1.0 / 2.2);  // a negative mean gives NaN, which the clamp sends to 0 like std::clamp's comparisons
            o[k] = (uint8_t)(fmin(fmax(gm, 0.0), 1.0) * 255.0 + 0.5);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Frames in flight: the pass behind every overlapped launch.  The launch wrote the sums of each of its frames whole
// into its chain's buffer (one plane per frame); this adds them to the accumulation buffer ONE FRAME AFTER THE OTHER
// and, when asked, makes the display of every frame from the sums as they stand after it (convertToRGBAKernel,
// Window.cuh:16-30).  The passes of a context run in launch order, so the buffer grows exactly as under serial
// launches -- acc += frame k, k = 0, 1, 2, ... -- and the picture made for frame k is the mean of samples 0..k, as in
// the reference's frame loop (main.cu:284-303): the newest goes to the display buffer, the ones before it to the
// history planes (pt_download_display_history).  HBM-bound: per pixel 16 B read per frame, 16 B read + 16 B written
// per pass, 4 B written per displayed frame, beside path launches that are VALU-bound and leave HBM idle.
// MEAN: the frames hold one sample each and the buffer the reference's running mean (PT_ACC_RUNNING_MEAN): frame f of
// the launch is the caller's frame_count n = s0 + f * stride, which overwrites when n == 0 and otherwise enters with
// weight t = 1 / (n + 1) -- the same float division the host makes for a serial launch (PathArgs::mean_t).
template <bool MEAN>
__device__ __forceinline__ void fold_frame(float4& a, const float4 q, uint32_t* dst, uint32_t n) {
    if (MEAN) {
        if (n == 0u) { a.x = q.x; a.y = q.y; a.z = q.z; }
        else {
            const float t = 1.0f / (float)(n + 1u);
            a.x = mean_step(q.x, a.x, t); a.y = mean_step(q.y, a.y, t); a.z = mean_step(q.z, a.z, t);
        }
        a.w = 1.0f;
        if (dst) *dst = pack_rgba8_cuda(a.x, a.y, a.z);
    } else {
        a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
        if (dst) {
            const float inv = a.w > 0.0f ? 1.0f / a.w : 0.0f;
            *dst = pack_rgba8_cuda(a.x * inv, a.y * inv, a.z * inv);
        }
    }
}
template <bool MEAN>
__global__ __launch_bounds__(256) void pt_fold_kernel(float4* __restrict__ accum, const float4* __restrict__ planes, uint32_t plane_stride,
                                                      uint32_t frames, uint32_t* __restrict__ newest, uint32_t* __restrict__ history,
                                                      uint32_t npix, uint32_t s0, uint32_t stride) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= npix) return;
    float4 a = accum[p];
    const float4* q = planes + p;
    // display of frame f: the last one into the display buffer, the others into history plane f (newest == nullptr: none)
    uint32_t* h = (newest && history) ? history + p : nullptr;  // (no history planes when a launch holds one frame)
    uint32_t f = 0, n = s0;
    for (; f + 4u < frames; f += 4u) {  // four loads in flight; the last frame is always left to the tail below
        const float4 q0 = q[0], q1 = q[plane_stride], q2 = q[2u * (size_t)plane_stride], q3 = q[3u * (size_t)plane_stride];
        fold_frame<MEAN>(a, q0, h, n); fold_frame<MEAN>(a, q1, h ? h + npix : nullptr, n + stride);
        fold_frame<MEAN>(a, q2, h ? h + 2u * (size_t)npix : nullptr, n + 2u * stride); fold_frame<MEAN>(a, q3, h ? h + 3u * (size_t)npix : nullptr, n + 3u * stride);
        q += 4u * (size_t)plane_stride;
        n += 4u * stride;
        if (h) h += 4u * (size_t)npix;
    }
    for (; f < frames; ++f) {
        fold_frame<MEAN>(a, *q, f + 1u == frames ? (newest ? newest + p : nullptr) : h, n);
        q += plane_stride;
        n += stride;
        if (h) h += npix;
    }
    accum[p] = a;
}

// ------------------------------------------------------------------------------------------
// FP32 VALU ceiling: 16 independent FMA chains per lane, `iters` rounds.  packed != 0 runs the
// chains as float2 so hipcc emits v_pk_fma_f32.  FLOPs = lanes * iters * 16 * 2.
typedef float float2v __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ __launch_bounds__(256) void pt_valu_peak_kernel(float* sink, uint32_t iters) {
    const float s = (float)(threadIdx.x & 7u) * 1e-3f + 0.999f;
    const float b = 1e-3f;
    if (PACKED) {
        float2v acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = float2v{(float)k, (float)k + 0.5f};
        const float2v sv = {s, s + 1e-4f}, bv = {b, b};
        for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_elementwise_fma(acc[k], sv, bv);
        }
        float r = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += acc[k].x + acc[k].y;
        if (r == 12345.678f) sink[0] = r;
    } else {
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = (float)k;
        for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(acc[k], s, b);
        }
        float r = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += acc[k];
        if (r == 12345.678f) sink[0] = r;
    }
}

}  // namespace

namespace {
using PathKernel = void (*)(const KArgs);
PathKernel path_kernel(bool nee, bool count, bool few) {
    static const PathKernel table[8] = {
        pt_path_kernel<false, false, false>, pt_path_kernel<false, false, true>, pt_path_kernel<false, true, false>, pt_path_kernel<false, true, true>,
        pt_path_kernel<true, false, false>,  pt_path_kernel<true, false, true>,  pt_path_kernel<true, true, false>,  pt_path_kernel<true, true, true>};
    return table[(nee ? 4 : 0) + (count ? 2 : 0) + (few ? 1 : 0)];
}
}  // namespace

hipError_t pt_launch_path(const DevSceneF& sc, const DevCamF& cam, const PathArgs& a, bool nee, bool count, bool few,
                          uint32_t blocks, hipStream_t stream) {
    const dim3 grid((blocks * 256u + PT_BLOCK - 1u) / PT_BLOCK), block(PT_BLOCK);
    KArgs k;
    k.sc = sc; k.cam = cam; k.a = a;
    k.a.width = cam.width; k.a.height = cam.height;
    // dynamic LDS: the lane-parallel mode's image of the scene, laid out here (at most 64 + 64 + 3 * 64 rows = 5 KB)
    const uint32_t ns = (uint32_t)sc.ns < 64u ? (uint32_t)sc.ns : 64u, np = (uint32_t)sc.np < 64u ? (uint32_t)sc.np : 64u, nt = (uint32_t)sc.nt < 64u ? (uint32_t)sc.nt : 64u;
    const uint32_t f4s = few ? ns + np + 3u * nt : 0u;
    k.a.lds_off_pla = ns; k.a.lds_off_tri = ns + np;
    hipLaunchKernelGGL(path_kernel(nee, count, few), grid, block, f4s * 16u, stream, k);
    return hipGetLastError();
}

hipError_t pt_path_resident_blocks(bool nee, bool count, bool few, int device, uint32_t* blocks) {
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, path_kernel(nee, count, few), PT_BLOCK, PT_LDS_MAX_BYTES);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    // (the query does not know the SGPR side of residency: with 82-96 SGPRs it answers 8 waves per SIMD where the hardware
    //  admits 7, MI355X_MICROARCH.md; the capped plain build is that case)
    constexpr int plain_waves = PT_PLAIN_SGPRS > 0 ? (800 / ((PT_PLAIN_SGPRS + 15) / 16 * 16 + 16) > 8 ? 8 : 800 / ((PT_PLAIN_SGPRS + 15) / 16 * 16 + 16)) : 8;
    if (PT_PLAIN_SGPRS > 0 && !nee && !count && !few && per_cu > 4 * plain_waves) per_cu = 4 * plain_waves;   // 96 -> 7 per SIMD, 80 -> 8
    if (per_cu < 1) per_cu = 1;
    *blocks = (uint32_t)per_cu * (uint32_t)prop.multiProcessorCount * PT_BLOCK / 256u;
    return hipSuccess;
}

hipError_t pt_launch_resolve(const float4* accum, void* out, uint32_t npix, uint32_t acc_mode, uint32_t mode,
                             hipStream_t stream) {
    hipLaunchKernelGGL(pt_resolve_kernel, dim3((npix + 255) / 256), dim3(256), 0, stream, accum, out, npix, acc_mode, mode);
    return hipGetLastError();
}

hipError_t pt_launch_fold(float4* accum, const float4* planes, uint32_t plane_stride, uint32_t frames, uint32_t* newest,
                          uint32_t* history, uint32_t npix, uint32_t acc_mode, uint32_t s0, uint32_t stride, hipStream_t stream) {
    if (acc_mode != 0u) hipLaunchKernelGGL(pt_fold_kernel<true>, dim3((npix + 255u) / 256u), dim3(256), 0, stream, accum, planes, plane_stride, frames, newest, history, npix, s0, stride);
    else hipLaunchKernelGGL(pt_fold_kernel<false>, dim3((npix + 255u) / 256u), dim3(256), 0, stream, accum, planes, plane_stride, frames, newest, history, npix, s0, stride);
    return hipGetLastError();
}

hipError_t pt_launch_valu_peak(float* sink, uint32_t packed, uint32_t iters, uint32_t blocks, hipStream_t stream) {
    if (packed) hipLaunchKernelGGL((pt_valu_peak_kernel<true>), dim3(blocks), dim3(256), 0, stream, sink, iters);
    else hipLaunchKernelGGL((pt_valu_peak_kernel<false>), dim3(blocks), dim3(256), 0, stream, sink, iters);
    return hipGetLastError();
}
