/*
 * pt_oracle.h — CPU oracle for the per-pixel render path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.  The
 * product (path_tracer_amd, libpt_amd.so) never links, loads or calls it.
 *
 * It is a plain-C restatement of the reference's algorithms:
 *   - `direct`  : cpu_pathtracer/src/main.cpp:14-53 (renderRegion) with
 *                 Scene.h:52-112, ImplicitObject.h:57-172, Camera.h:65-85,124-142,
 *                 Ray.h:43-58,171-194, Intersection.h:16-19, Vec3.h (FP64).
 *   - `path`    : cuda_pathtracer/src/raytracer_kernel.cu:44-184,202-305 with
 *                 Material.cuh:36-259, Random.cuh:22-55, Ray.cuh:31-53,133-153,
 *                 Intersection.cuh:29-32, Camera.cu:3-23, in float (the reference's
 *                 arithmetic type) and in double (the converged-image referee).
 * The random streams of the reference (curand XORWOW; libstdc++ mt19937) are NOT
 * reproduced: north_star replaces both by a counter-based generator keyed on
 * (pixel, sample, bounce), specified in DESIGN.md and restated here.
 *
 * PARITY PIN: the reference CPU program cannot be built in this image
 * (cpu_pathtracer/include/Vec3.h:5 needs <format>, absent from g++ 11.4, and no
 * stand-in header may be written), and the reference holds no tests or golden
 * vectors.  The known answers are (1) the two primary-hit histograms recorded
 * in BASELINE.md §2 / SURVEY.md §8c: tests/test_oracle_pins.py checks the
 * `direct`-side restatement (camera + Scene::intersect) against both; (2) the one
 * output of the CUDA program the reference holds, images/cuda.png (a screenshot of
 * its default scene): the `path` restatement reproduces it to a few 8-bit levels
 * (fixture tests/golden/reference_cuda_screenshot_480x270.npz).  That is an
 * image-level pin (8 bits, unknown sample count); (3) the one output of the CPU program
 * it holds, images/cpu.png (a grab at one sample per pixel; tests/cpu_screenshot.py,
 * fixture tests/golden/reference_cpu_screenshot_711x1270.npz): the `direct`
 * restatement puts every object where the grab shows it, gives the grab's light term
 * pixel by pixel on both coloured walls and its bytes wherever a pixel shows two
 * unsaturated channels of one light sample.  Which `direct` light samples come out
 * visible (FP rounding of the reference's -ffast-math build) and the
 * individual formulas of the `path` integrator remain "parity unpinned" beyond these.
 */
#ifndef PT_ORACLE_H_
#define PT_ORACLE_H_

#include "../include/pt.h" /* POD input types only (PtSceneSoA, PtCamera, PtRenderParams, PtCounters) */

#ifdef __cplusplus
extern "C" {
#endif

/* Counter RNG: exposed so tests can pin known values. */
uint32_t pto_hash(uint32_t x);
uint32_t pto_path_key(uint32_t seed, uint32_t pixel, uint32_t sample);
uint32_t pto_draw(uint32_t key, uint32_t slot, uint32_t dim);
/* The specified sin / cos / acos of the `direct` light sample (pt_oracle.c): out3 = {sin x, cos x, acos y}. */
int pto_fmath(double x, double y, double* out3);

/* Primary-hit object / material ids through pixel centres, FP64. */
int pto_primary_ids(const PtSceneSoA* scene, const PtCamera* cam, uint32_t* obj_ids, uint32_t* mat_ids);

/* The same with the scene's rectangles tested by Rectangle::intersect (cpu_pathtracer/include/Polygon.h:103-153,
 * cuda_pathtracer/include/Polygon.cuh:130-168: plane hit, then alpha / beta in [0, 1]) in place of the triangles the
 * host flattens them to: rect_o_u_v holds origin, u, v (9 doubles) per rectangle, rect_obj / rect_mat its object and
 * material ids; triangles of the scene whose object id is in rect_obj are skipped. */
int pto_primary_ids_rect(const PtSceneSoA* scene, const PtCamera* cam, uint32_t n_rect, const double* rect_o_u_v,
                         const uint32_t* rect_obj, const uint32_t* rect_mat, uint32_t* obj_ids, uint32_t* mat_ids);

/* `direct` integrator (renderRegion): adds samples [s0, s0+n) of every pixel of
 * rows [y0, y1) to accum (float4 per pixel: sum r,g,b and sample count). */
int pto_render_direct(const PtSceneSoA* scene, const PtCamera* cam, uint32_t y0, uint32_t y1,
                      uint32_t s0, uint32_t n, const PtRenderParams* params, float* accum);
/* Same, returning the per-sample components for pixel (x,y), sample s:
 * out[0]=primary obj id (as double, -1 miss), out[1]=ambient+background term,
 * out[2..4]=emission term, out[5..7]=stochastic direct term. */
int pto_direct_components(const PtSceneSoA* scene, const PtCamera* cam, uint32_t x, uint32_t y,
                          uint32_t s, const PtRenderParams* params, double* out8);

/* `path` integrator (renderKernel + traceRay).  precision: 32 = float, 64 = double.
 * acc_mode as PtRenderParams.  counters may be NULL. */
int pto_render_path(const PtSceneSoA* scene, const PtCamera* cam, uint32_t y0, uint32_t y1,
                    uint32_t s0, uint32_t n, const PtRenderParams* params, int precision,
                    float* accum, PtCounters* counters);
/* Samples [s0, s0+n) of every pixel by path length: hist[k], k < 64, counts the paths that traced k rays (iterations of
 * traceRay's loop, raytracer_kernel.cu:206-208; shadow rays apart), the last bin every longer one. */
int pto_path_length_histogram(const PtSceneSoA* scene, const PtCamera* cam, uint32_t s0, uint32_t n,
                              const PtRenderParams* params, int precision, uint64_t* hist64);
/* Radiance of one sample of one pixel, double[3] (for targeted tests). */
int pto_path_sample(const PtSceneSoA* scene, const PtCamera* cam, uint32_t x, uint32_t y, uint32_t s,
                    const PtRenderParams* params, int precision, double* rgb);
/* Per-pixel mean and variance of the per-sample radiance over samples [s0,s0+n)
 * (double precision build): mean[3*npix], var[3*npix]. */
int pto_path_stats(const PtSceneSoA* scene, const PtCamera* cam, uint32_t s0, uint32_t n,
                   const PtRenderParams* params, double* mean, double* var);

/* Resolve (convertToRGBAKernel / CPU accumulate+gamma / PPM).  out sized per mode. */
int pto_resolve(const float* accum, uint32_t npix, uint32_t acc_mode, uint32_t mode, void* out);

#ifdef __cplusplus
}
#endif
#endif
