/*
 * pt.h — C ABI of the MI355X-native per-pixel render path.
 *
 * This is the drop-in boundary for the L3→L2 line of the reference
 * (Qervas/path_tracer): every symbol main.cu / main.cpp call across that line
 * has a counterpart here.  Plain pointers and sizes only; no C++ / torch types;
 * no exception crosses the ABI.  Every entry point returns 0 on success and a
 * negative PT_ERR_* code otherwise; pt_last_error() returns a thread-local
 * human-readable message for the last failing call, and "" when the thread's
 * last call succeeded (every entry point but pt_last_error clears it on entry).
 *
 * Citations are `path:line` under the reference tree (cuda_pathtracer/… or
 * cpu_pathtracer/…).
 *
 * Threading: one context per GPU.  A context is not thread-safe; different
 * contexts may be driven from different threads.  All device work of a context
 * is enqueued on the context's stream (its own, or the one given to
 * pt_set_stream); nothing synchronises with the host unless documented.  With
 * pt_set_frames_in_flight(n > 1) path launches run on internal streams that fork
 * from and join that stream (see there).
 */
#ifndef PT_H_
#define PT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- errors -- */
#define PT_OK 0
#define PT_ERR_INVALID (-1)   /* bad argument / inconsistent sizes            */
#define PT_ERR_HIP (-2)       /* a HIP runtime call failed                     */
#define PT_ERR_STATE (-3)     /* call order (no scene / camera / frame yet)    */
#define PT_ERR_CAPACITY (-4)  /* request exceeds a documented maximum          */

/* ------------------------------------------------------------- materials -- */
/* Tags of the reference's five polymorphic materials
 * (cuda_pathtracer/include/Material.cuh:36,82,125,178,235). */
#define PT_MAT_LAMBERTIAN 0u
#define PT_MAT_METAL 1u
#define PT_MAT_DIELECTRIC 2u
#define PT_MAT_GLOSSY 3u
#define PT_MAT_EMISSIVE 4u

/* ----------------------------------------------------------------- scene -- */
/*
 * Flattened scene, structure-of-arrays.  Replaces the GPUSphere / GPUPlane /
 * GPUTriangle AoS upload and the eight device-side material objects
 * (cuda_pathtracer/src/main.cu:214-278, include/raytracer_kernel.cuh:21-48,
 * include/MaterialFactory.cuh:41-75).  Array order within each primitive class
 * is insertion order; the kernels test all spheres, then all planes, then all
 * triangles with a strict `<` (raytracer_kernel.cu:157-181).
 *
 * All real-valued inputs are double so that the cpu_pathtracer scene
 * (cpu_pathtracer/src/main.cpp:63-131, double literals) is carried exactly; the
 * library keeps a double copy for the FP64 kernels (primary ids, `direct`
 * integrator) and a float copy for the FP32 `path` kernel.
 *
 * `*_obj` is the object's insertion index over ALL objects of the scene (the
 * "object id" reported by pt_primary_ids); `*_mat` indexes the material table.
 * `sph_emissive[i]` != 0 marks a sphere on which makeEmissive() was called
 * (ImplicitObject.cuh:31-35 / ImplicitObject.h:30-34); sph_em_* is then
 * colour*strength.  The caller keeps ownership of every array; they are copied
 * during pt_upload_scene.
 */
typedef struct PtSceneSoA {
    uint32_t num_spheres;
    uint32_t num_planes;
    uint32_t num_triangles;
    uint32_t num_materials;

    /* spheres: Sphere_t(center, radius, material), ImplicitObject.cuh:71 */
    const double *sph_cx, *sph_cy, *sph_cz, *sph_r;
    const uint32_t *sph_mat, *sph_obj, *sph_emissive;
    const double *sph_em_r, *sph_em_g, *sph_em_b;

    /* planes: Plane_t(point, normal, material), ImplicitObject.cuh:162
     * (normal is normalised by the library, as the constructor does). */
    const double *pl_px, *pl_py, *pl_pz, *pl_nx, *pl_ny, *pl_nz;
    const uint32_t *pl_mat, *pl_obj;

    /* triangles: Triangle_t(v0, v1, v2, material), Polygon.cuh:48
     * (geometric normal cross(v1-v0, v2-v0) normalised by the library). */
    const double *tri_v0x, *tri_v0y, *tri_v0z;
    const double *tri_v1x, *tri_v1y, *tri_v1z;
    const double *tri_v2x, *tri_v2y, *tri_v2z;
    const uint32_t *tri_mat, *tri_obj;

    /* materials: Lambertian(albedo) / Metal(albedo, roughness) /
     * Dielectric(ior) / Glossy(albedo, roughness, metallic) /
     * Emissive(colour, strength) — Material.cuh:41,92,147,185,241.
     * mat_em_* = colour*strength of an Emissive material. */
    const uint32_t *mat_type;
    const double *mat_albedo_r, *mat_albedo_g, *mat_albedo_b;
    const double *mat_roughness, *mat_metallic, *mat_ior;
    const double *mat_em_r, *mat_em_g, *mat_em_b;
} PtSceneSoA;

/* Capacity of one scene (the reference's __constant__ arrays hold 32/32/128,
 * raytracer_kernel.cu:12-18; BASELINE config 5 needs 4096 spheres). */
#define PT_MAX_SPHERES 65536u
#define PT_MAX_PLANES 4096u
#define PT_MAX_TRIANGLES 65536u
#define PT_MAX_MATERIALS 65536u

/* ---------------------------------------------------------------- camera -- */
/* Which ray-generation formula the kernels use. */
#define PT_CAMERA_CUDA 0u /* generateCameraRay, raytracer_kernel.cu:245-265,287-288 */
#define PT_CAMERA_CPU 1u  /* Camera::getRayForPixel, cpu_pathtracer/include/Camera.h:124-142
                             (including its vertical-offset behaviour) */

/* Camera(origin, target, up, Settings{fov, aspect_ratio, aperture, focus_dist,
 * width, height, use_dof}) — Camera.cuh:12-20,39 / Camera.h:18-26,54.
 * The library derives the look-at basis exactly as Camera_t::initialize
 * (Camera.cu:3-23, float) or Camera::initialize (Camera.h:65-85, double).
 * `up` is used as given when `normalize_up` is 0 (Camera::lookAt(from,to,up),
 * Camera.h:251-256) and normalised first otherwise (the constructors). */
typedef struct PtCamera {
    double origin[3];
    double target[3];
    double up[3];
    double fov_deg;
    double aspect_ratio;
    double aperture;
    double focus_dist;
    uint32_t width;
    uint32_t height;
    uint32_t use_dof;      /* 1 = thin-lens rays (Camera_t::getRayDevice, Camera.cu:31-47) in the path integrator:
                              lens radius aperture/2, screen at focus_dist.  The reference's render loops never
                              set it; the direct integrator and the id pass ignore it. */
    uint32_t model;        /* PT_CAMERA_CUDA | PT_CAMERA_CPU */
    uint32_t normalize_up; /* 1 = constructor behaviour */
} PtCamera;

/* ---------------------------------------------------------------- render -- */
#define PT_INTEGRATOR_PATH 0u   /* traceRay, raytracer_kernel.cu:202-243 (FP32)   */
#define PT_INTEGRATOR_DIRECT 1u /* renderRegion, cpu_pathtracer/src/main.cpp:14-53 (FP64) */

#define PT_FLAG_NEE_MIS 1u      /* next-event estimation + balance-heuristic MIS (north_star addition) */
#define PT_FLAG_LIGHT_COLOUR 2u /* lights emit their makeEmissive colour*strength instead of the shared
                                   Emissive material's (reference behaviour) */
#define PT_FLAG_COUNTERS 4u     /* count rays / bounces for pt_counters (instrumented kernel) */

#define PT_ACC_SUM 0u          /* buffer holds (sum_r, sum_g, sum_b, n_samples) */
#define PT_ACC_RUNNING_MEAN 1u /* renderKernel's in-place mean, alpha = 1 (raytracer_kernel.cu:297-304) */

typedef struct PtRenderParams {
    uint32_t integrator;    /* PT_INTEGRATOR_*                                         */
    uint32_t max_depth;     /* 50 in the reference (raytracer_kernel.cu:294)           */
    uint32_t rr_depth;      /* Russian roulette when depth > rr_depth; reference: 3    */
    uint32_t seed;          /* counter-RNG seed (replaces curand_init, :267-272)       */
    uint32_t flags;         /* PT_FLAG_*                                               */
    uint32_t acc_mode;      /* PT_ACC_*                                                */
    uint32_t fused_rgba8;   /* != 0: the render launch also writes the display buffer
                               (convertToRGBAKernel, Window.cuh:16-30) for the pixels it touched */
    uint32_t pixels_per_wave; /* tuning knob, 0 = library default                      */
} PtRenderParams;

/* Event counts of the instrumented kernel; identical on CPU oracle and GPU
 * because both key the RNG on (pixel, sample, bounce).  Feed the fixed FLOP
 * table of SURVEY §8(d). */
typedef struct PtCounters {
    uint64_t samples;         /* camera paths started                                  */
    uint64_t rays;            /* intersectScene calls, shadow rays included            */
    uint64_t hits;            /* rays that hit something (winning-hit records built)   */
    uint64_t sphere_roots;    /* ray-sphere tests with discriminant >= 0               */
    uint64_t bounces[5];      /* scatter evaluations per PT_MAT_* of the hit material  */
    uint64_t diffuse_updates; /* non-specular throughput/RR updates (:229-238)         */
    uint64_t nee_samples;     /* light samples drawn (PT_FLAG_NEE_MIS)                 */
    uint64_t shadow_rays;     /* of `rays`, how many were shadow rays                  */
} PtCounters;

#define PT_RESOLVE_RGBA8_CUDA 0u /* clamp(c*255,0,255) truncated, 0x00RRGGBB (Window.cuh:25-29)         */
#define PT_RESOLVE_RGBA8_CPU 1u  /* clamp, pow(1/2.2), *255 truncated, 0x00RRGGBB
                                    (main.cpp:228-236 + WindowX11.h:130-139)                            */
#define PT_RESOLVE_FLOAT_MEAN 2u /* float4 (mean_r, mean_g, mean_b, 1)                                   */
#define PT_RESOLVE_RGB8_PPM 3u   /* 3 bytes/pixel, gamma 2.2, +0.5 rounding (Camera.h:162-180, Color.h:86-91) */

typedef struct PtContext PtContext;

/* -- lifetime ------------------------------------------------------------- */
/* Replaces initializeCuda()/cudaSetDevice(0) (main.cu:28-40). */
int pt_create(int device_id, PtContext** out_ctx);
int pt_destroy(PtContext* ctx);
const char* pt_last_error(void);
/* Library/ABI version, for the binding to check. */
uint32_t pt_abi_version(void);

/* Use an existing HIP stream (e.g. torch's current stream) for all work of the
 * context.  NULL is HIP's null (legacy default) stream, which is what torch's
 * default stream is.  pt_use_own_stream() goes back to the context's own
 * non-blocking stream.  Work the context already queued on the stream it
 * leaves stays ordered before whatever is queued on the new one (an event wait,
 * no host synchronisation). */
int pt_set_stream(PtContext* ctx, void* hip_stream);
int pt_use_own_stream(PtContext* ctx);

/* -- scene / camera -------------------------------------------------------- */
/* Replaces MaterialFactory::createMaterialsOnDevice (MaterialFactory.cuh:41-75),
 * the flatten loop and the cudaMemcpyToSymbol uploads (main.cu:214-278). */
int pt_upload_scene(PtContext* ctx, const PtSceneSoA* scene);
/* Replaces initializeGPUData(camera, scene) (raytracer_kernel.cu:308-321). Also
 * sets the frame size to camera->width x camera->height (allocating the
 * accumulation and display buffers unless one is bound). */
int pt_set_camera(PtContext* ctx, const PtCamera* camera);
/* Replaces the cudaMalloc of d_output (main.cu:55-57) / Window_t's RGBA buffer. */
int pt_resize(PtContext* ctx, uint32_t width, uint32_t height);
/* Render into a caller-owned device buffer of width*height float4 instead of
 * the context's own (NULL unbinds).  This is how a torch tensor becomes the
 * accumulation buffer for the RCCL reduce.  Work on the buffer is asynchronous
 * like everything else: unbinding does not wait for it, so pt_sync() (or a wait
 * on the context's stream) comes before the caller frees or reuses the memory. */
int pt_bind_accum(PtContext* ctx, void* dev_float4);
/* Zero the accumulation buffer and the sample count — the CPU app's reset on
 * camera movement (cpu_pathtracer/src/main.cpp:201-205). */
int pt_reset(PtContext* ctx);

/* -- render ---------------------------------------------------------------- */
/* Adds samples [sample_begin, sample_begin+sample_count) to every pixel;
 * generalises launchRenderKernel's frame_count (one sample per launch).
 * Asynchronous on the context's stream. */
int pt_render(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count,
              const PtRenderParams* params);
/* pt_render for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for hosts that must not
 * call hipMalloc while they render — code under HIP graph capture, tight serving loops, harnesses that forbid it.
 * Adds samples [sample_begin, sample_begin + sample_count) of the path integrator to `accum_float4`, a caller-owned
 * device buffer of width*height float4 (in/out, 16-byte aligned; width x height must be the camera's), as ONE
 * serial launch on the context's stream.  The call makes no hipMalloc / hipFree / stream / event and no occupancy
 * query: the kernel's only memory beyond its inputs and outputs, its ticket counters, lives in a 4 KB
 * file-scope __device__ array (one copy per GPU, held until the process ends) whose address pt_create resolved.
 * Scene and camera are the context's (uploaded at set-up); with fused_rgba8 the display goes to the context's
 * display buffer.  Same kernel and same launch shape as a pt_render call that is not cut into parts, so for such a
 * call the buffer is bit-identical to pt_render's.
 * Limits: at most PT_NOALLOC_MAX_PIXELS pixels and PT_NOALLOC_MAX_SAMPLES samples per call (PT_ERR_CAPACITY
 * beyond); path integrator only; PT_ERR_STATE while frames are in flight (pt_flush first).
 * Every call on a GPU — from any context — shares that GPU's array and clears the part it uses on its stream:
 * calls that overlap in time on one GPU race for the counters (undefined image); serialise them on one stream. */
#define PT_NOALLOC_MAX_PIXELS (1u << 26) /* 8192 x 8192 */
#define PT_NOALLOC_MAX_SAMPLES 4096u
int pt_render_noalloc(PtContext* ctx, void* accum_float4, uint32_t width, uint32_t height,
                      uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* params);
/* Frames in flight (1..8, default 1).  The reference's frame loop (main.cu:284-303)
 * serialises launch and display, one frame at a time; there is no counterpart.
 * With n > 1 a path call -- sums, or the reference's running mean (PT_ACC_RUNNING_MEAN,
 * one sample per call) -- is a FRAME that the library may hold back and overlap with
 * others:
 *  - Consecutive calls with the same parameters and sample count whose sample
 *    ranges are equally spaced -- each starting where the one before it ended (the
 *    reference's loop is exactly that), or a fixed distance further on (a rank of a
 *    sample-split render takes its slice of every step) -- are collected -- up to 64 of them, fewer on frames beyond 2 Mpixel
 *    (2^27 pixel-frames per launch) -- and
 *    leave as ONE launch of the path kernel when that many have gathered or when
 *    anything needs their result.  The launch hands out (frame, pixel) pairs from one
 *    ticket space: one ramp and one tail instead of one per frame, and 64 x as many
 *    work items per lane as a frame has (2.3 pixels per resident lane at 1280x720).
 *  - Launches go round-robin to n internal streams, so launch k+1 fills the
 *    compute units that launch k leaves idle while its last long paths finish.
 *  - Each launch writes the sums of each of its frames into a plane of a buffer of
 *    its chain, and a fold pass behind it adds them to the accumulation buffer one
 *    frame after the other (and, with fused_rgba8, makes every frame's display from
 *    the sums as they stand after it).  The fold passes run in launch order, so
 *    the buffer receives acc += frame k -- or, for the running mean, acc = frame k * t
 *    + acc * (1 - t) with t = 1 / (frame_count + 1), frame_count == 0 overwriting,
 *    raytracer_kernel.cu:297-304 -- for k = 0, 1, 2, ... exactly as under serial
 *    launches: the image is bit-identical to the n = 1 image, and the display made
 *    for frame k is the mean of samples 0..k as in the reference's loop.  The newest
 *    display is in the display buffer (pt_readback); the ones the same fold pass
 *    made before it are kept until the next fold pass (pt_readback_history).
 * Every call that reads or replaces the accumulation buffer first launches what
 * is held back and waits (on the context's stream, not on the host) for the fold
 * passes still queued (pt_resolve, pt_readback*, pt_upload_accum, pt_reset,
 * pt_sync, pt_counters, pt_bind_accum, pt_set_stream, pt_upload_scene,
 * pt_set_camera, pt_resize); a caller that reads pt_accum_device_ptr() or a bound
 * buffer itself, or that waits on its own stream, calls pt_flush() first, which is
 * asynchronous on the context's stream.  A caller that flushes after every frame
 * gets serial launches.  A frame that is held back has only been validated: a
 * failure of its launch (an allocation of the chains' buffers, say) is reported by
 * the call that launches it -- a later pt_render, pt_flush or any of the calls
 * above -- and pt_destroy drops frames that were never launched.
 * n > 3 needs more hardware queues than HIP's default of 4 per process:
 * GPU_MAX_HW_QUEUES >= n + 4 in the environment before HIP initialises (16 for
 * n = 8).  Chains that share a queue serialise (with the default, 4 chains are
 * slower than 3), so the request is GRANTED only up to what the variable, as
 * the library finds it at pt_create, leaves room for: 3 chains when it is unset
 * or <= 7, GPU_MAX_HW_QUEUES - 4 beyond that.  The call still returns PT_OK;
 * pt_frames_in_flight() returns the number in effect.  Memory: per chain that is
 * used, one plane of the frame's size (16 B per pixel) for every frame a launch may
 * hold (944 MB at 1280x720, at most 2 GiB), plus 4 B per pixel and frame for the
 * display history (one per context); allocated on the chain's first launch, whole
 * (so that no later launch of the frame loop reallocates), freed when the frame is
 * resized.  When the device has no room for them the frames are rendered by
 * serial launches instead: same sums, same displays, no history.
 *
 * Large calls.  Within a frame a pixel's samples belong to one lane, so one launch
 * with many samples per pixel over few pixels per lane (2.3 at 1280x720) ends with
 * most lanes idle.  On frames of at most 2^22 pixels a PT_ACC_SUM path call for 4
 * samples or more is therefore rendered as FRAMES of 1..4 samples (32 frames for
 * up to 128 samples, 4 samples per frame beyond) of multi-frame launches: with
 * n > 1 they join the open burst like any other frame; with n = 1 the call is a
 * burst of its own on three chains (GPU_MAX_HW_QUEUES - 4 of them when the
 * environment grants more, up to 8; each chain's planes sized by what the call puts
 * on it: 16 planes for 16 samples), folded (and, with fused_rgba8, displayed)
 * before pt_render returns, so nothing stays in flight.  The cut depends only on
 * the call's arguments and the frame size: the image is reproducible, the same
 * bit for bit for every n, and equals the single launch's up to float summation
 * order (the frames' sums are added one after the other instead of sample by
 * sample; with one sample per frame that IS the reference's progressive order,
 * main.cu:284-303).  (Environment variable PT_AMD_CHUNK=0, read at pt_create,
 * keeps every call one launch.) */
int pt_set_frames_in_flight(PtContext* ctx, uint32_t n);
uint32_t pt_frames_in_flight(PtContext* ctx);
/* Launches of the path kernel this context has made so far (frames that are held
 * back are not launched yet; a frame rendered ahead counts when a call takes it,
 * pt_render_ahead_stats counts the others): lets a caller relate frames to
 * launches when it reads a profile. */
uint64_t pt_path_launches(PtContext* ctx);
int pt_flush(PtContext* ctx);
/* Render-ahead, for the loop that launches one frame at a time and shows it before it launches the next (the
 * reference's own: launchRenderKernel, updateScreen, main.cu:284-303) -- one frame in flight, where a 1280x720 frame
 * of one sample spends more than half of its launch draining the scene's longest paths.  Once two calls in a row have
 * continued one another (same parameters, same sample count, equally spaced sample ranges: pt_render or
 * launchRenderKernel alike), the library also launches the frame that the NEXT call will ask for if the loop goes
 * on, on a stream of its own into a plane of its own; that launch fills the chip while the current frame's tail
 * drains and while the host shows it.  If the next call is that frame it only adds the plane to the accumulation
 * buffer (or applies the running mean's step) and makes the display, as a frame in flight is folded: buffer and
 * display are the serial loop's bit for bit.  If it is anything else -- another sample range, other parameters, a new
 * camera or scene -- the plane is dropped and the call renders as usual; render-ahead resumes when two calls
 * continue one another again.  The cost: one frame of GPU work wasted when the loop stops, and four planes of 16 B
 * per pixel (allocated when it first happens; a device without room for them renders serially).  On by default for
 * every call that is one launch of its own -- fewer than 4 samples, or any number on a frame of more than 2^22 pixels
 * (see "Large calls") -- on frames of up to 2^24 pixels, without event counters; off with frames in flight (n > 1),
 * which do better for callers that can let the library hold frames back.  pt_render_noalloc never renders ahead.
 * `on` = 0 turns it off (strictly one launch per call, the previous call's kernel ended before the next begins).
 * pt_render_ahead_stats: frames launched ahead so far, and how many of them a call then used.
 * No counterpart in the reference. */
int pt_set_render_ahead(PtContext* ctx, uint32_t on);
int pt_render_ahead_stats(PtContext* ctx, uint64_t* launched, uint64_t* used);
/* Resolve the accumulation buffer into the context's display buffer
 * (convertToRGBAKernel / the CPU accumulate+gamma loop).  Asynchronous. */
int pt_resolve(PtContext* ctx, uint32_t mode);
/* The same conversion for pixels [first_pixel, first_pixel + pixel_count) of the accumulation buffer (row-major
 * pixel index) into a CALLER-OWNED device buffer (pixel_count elements of the mode's size: 4 B, 16 B for
 * PT_RESOLVE_FLOAT_MEAN, 3 B for PT_RESOLVE_RGB8_PPM).  Multi-GPU hosts that combine their buffers with a
 * reduce-scatter (each rank then owns the sums of 1/N of the pixels) resolve their own slice with this
 * and gather the 4-byte pixels, instead of sending every float to one GPU.  The context's display buffer and the
 * display history are not touched.  No counterpart in the reference (single GPU, main.cu:30).  Asynchronous. */
int pt_resolve_range(PtContext* ctx, uint32_t mode, uint32_t first_pixel, uint32_t pixel_count, void* dev_out);
/* Copy the display buffer of the last pt_resolve / fused render to the host
 * (Window_t::updateScreen's cudaMemcpy, Window.cuh:160).  Synchronises. */
int pt_readback(PtContext* ctx, void* host_dst, size_t bytes);
/* Frames in flight: the displays (RGBA8, as pt_readback returns them) that the last
 * fold pass made BEFORE its newest one, which is in the display buffer.
 * pt_history_frames() returns how many are held (0 .. 63; 0 after anything else
 * wrote the display: pt_resolve, a serial launch with fused_rgba8, pt_reset);
 * frames_back = 1 is the frame before the newest.  Both first launch and fold
 * what is in flight.  No counterpart in the reference, whose loop shows each
 * frame before it launches the next (Window_t::updateScreen, Window.cuh:160). */
uint32_t pt_history_frames(PtContext* ctx);
int pt_readback_history(PtContext* ctx, uint32_t frames_back, void* host_dst, size_t bytes);
/* Copy the raw accumulation buffer (width*height float4) to the host. Synchronises. */
int pt_readback_accum(PtContext* ctx, float* host_dst, size_t bytes);
/* Upload a raw accumulation buffer (resume from a checkpoint). */
int pt_upload_accum(PtContext* ctx, const float* host_src, size_t bytes);
int pt_sync(PtContext* ctx);

/* Multi-GPU: every rank holds the whole frame and renders its own sample indices;
 * this adds the accumulation buffers (sums and counts) of all ranks of an existing RCCL
 * communicator in place — ncclReduce to `root`, or ncclAllReduce when root < 0 — on the
 * context's stream, after folding the frames in flight.  `nccl_comm` is the caller's
 * ncclComm_t (one process per GPU, created by the host with ncclCommInitRank).  The library
 * does not link RCCL: it resolves ncclReduce / ncclAllReduce from the copy already loaded in the
 * process (the one that made the communicator), falling back to librccl.so.1.  Asynchronous.
 * (The reference is single-GPU, main.cu:30; no counterpart.) */
int pt_reduce(PtContext* ctx, void* nccl_comm, int root);

/* The cheaper combine of SURVEY §8e for the same hosts: ncclReduceScatter in place, after which THIS rank's slice of
 * the buffer -- pixels [*first_pixel, *first_pixel + *pixel_count) = the rank-th of `world` equal slices, row-major --
 * holds the sums and counts of all ranks (the other pixels keep this rank's own); follow with
 * pt_resolve_range on that slice and gather the 4-byte pixels.  `rank` / `world` are the communicator's; the pixel
 * count must be divisible by `world` (PT_ERR_INVALID otherwise: use pt_reduce).  Asynchronous.  No counterpart. */
int pt_reduce_scatter(PtContext* ctx, void* nccl_comm, uint32_t rank, uint32_t world, uint32_t* first_pixel, uint32_t* pixel_count);

/* Device pointers for callers that drive collectives themselves. */
void* pt_accum_device_ptr(PtContext* ctx);
void* pt_display_device_ptr(PtContext* ctx);

/* -- checks / measurement --------------------------------------------------- */
/* Primary-hit object id and material id per pixel through the pixel centre, in
 * FP64 with the formulas of Scene::intersect (cpu_pathtracer/include/Scene.h:52-72)
 * and the camera model set by pt_set_camera.  0xFFFFFFFF = miss.  Either output
 * may be NULL.  Synchronises. */
int pt_primary_ids(PtContext* ctx, uint32_t* host_obj_ids, uint32_t* host_mat_ids);
/* Counters accumulated by renders run with PT_FLAG_COUNTERS since the last
 * pt_reset_counters.  Synchronises. */
int pt_counters(PtContext* ctx, PtCounters* out);
int pt_reset_counters(PtContext* ctx);
/* Samples by path length, accumulated with the counters (PT_FLAG_COUNTERS; cleared by pt_reset_counters):
 * out_bins[k], k < PT_HISTOGRAM_BINS, is the number of camera paths that traced k rays -- one per iteration of
 * traceRay's loop (raytracer_kernel.cu:206-208), shadow rays apart; the last bin also holds every longer path.
 * Sum of bins = PtCounters.samples, sum of k * bins[k] = rays - shadow_rays (max_depth < PT_HISTOGRAM_BINS).
 * This is the "bounce histogram" of SURVEY §8(d).4: the divergence the lanes of a wave see.  Synchronises. */
#define PT_HISTOGRAM_BINS 64u
int pt_path_length_histogram(PtContext* ctx, uint64_t* out_bins);
/* FP32 VALU ceiling micro-benchmark: runs `iters` dependent-chain FMA rounds on
 * every lane of a full-chip grid; packed != 0 uses v_pk_fma_f32.  Returns the
 * achieved TFLOP/s in *out_tflops.  Synchronises. */
int pt_valu_peak(PtContext* ctx, uint32_t packed, uint32_t iters, double* out_tflops);

/* -- reference call shape ---------------------------------------------------- */
/* extern "C" void launchRenderKernel(float4* output, uint32_t width, uint32_t height,
 *                                    uint32_t frame_count, dim3 grid, dim3 block)
 * (raytracer_kernel.cuh:51, raytracer_kernel.cu:325-327) on the context made
 * current by pt_make_current(): one more sample per pixel, running mean in
 * `output` (device, caller-owned); frame_count == 0 overwrites.  grid and block
 * are taken BY VALUE as the reference passes its dim3 (three uint32_t: PtDim3 has
 * the layout and the x86-64 calling convention of CUDA's and HIP's dim3), and are
 * ignored: the launch shape is the library's.  pt_host.hpp gives C++ hosts a
 * `dim3` with the (x = 1, y = 1, z = 1) constructor, so the reference's call site
 * (main.cu:208-212,287-288) compiles as it stands.
 * Stream: the reference's loop works on the legacy default stream throughout
 * (renderKernel, then convertToRGBAKernel on the same buffer).  A context whose
 * stream nobody chose (pt_set_stream / pt_use_own_stream) moves to the null stream
 * on its first call through this function, so the caller's own kernels and copies
 * on `output` stay ordered with the launch as they were.
 * Frames in flight: after pt_set_frames_in_flight(n > 1) consecutive calls that name
 * the same `output` are frames like pt_render's -- held back, collected into one
 * launch, the running mean applied by the fold pass in frame order, bit-identical
 * to one launch per call.  `output` then stays bound to the context until the next
 * entry point other than this one, which first folds what is held back into it; a
 * caller that reads `output` itself calls pt_flush() before (a loop that shows
 * every frame therefore gets serial launches, as with n = 1, where nothing changes).
 * The function returns void like the reference's; its error channel is
 * pt_last_error(), which is empty after a launch that was enqueued (every entry
 * point clears the thread's message on entry), so
 * `if (*pt_last_error())` replaces CUDA_CHECK(cudaGetLastError()) (main.cu:289). */
typedef struct PtDim3 { uint32_t x, y, z; } PtDim3;
int pt_make_current(PtContext* ctx);
void launchRenderKernel(void* output_float4, uint32_t width, uint32_t height, uint32_t frame_count,
                        PtDim3 grid, PtDim3 block);

/* -- interactive camera (host only, no GPU needed) ---------------------------- */
/* One iteration of the CPU app's input handling (cpu_pathtracer/src/main.cpp:157-195):
 * yaw += dx * 0.002; pitch = clamp(pitch - dy * 0.002, -pi/2 + 0.1, pi/2 - 0.1);
 * forward = (cos p cos y, sin p, cos p sin y); W/S along forward, A/D along
 * normalize(forward x (0,1,0)), space / left-ctrl along +-Y, 0.1 per frame; then
 * camera.lookAt(pos, pos + forward, (0,1,0)) (un-normalised `up`, Camera.h:251-256).
 * Updates camera->origin/target/up/normalize_up and *yaw / *pitch.  Returns 1 when
 * the app would reset its accumulation (any mouse motion or movement key,
 * main.cpp:188-199: follow with pt_set_camera + pt_reset), 0 when not, < 0 on error.
 * The X11 event pump that produces (dx, dy, keys) is out of scope; a scripted
 * sequence replaces it (tools/pt_headless --script). */
#define PT_KEY_W 1u
#define PT_KEY_S 2u
#define PT_KEY_A 4u
#define PT_KEY_D 8u
#define PT_KEY_SPACE 16u
#define PT_KEY_CTRL 32u
int pt_fly_camera_step(PtCamera* camera, double* yaw, double* pitch, double mouse_dx, double mouse_dy, uint32_t keys);

/* -- built-in scenes (host only, no GPU needed) ------------------------------ */
/* Fills *scene and *camera with pointers into storage owned by the library
 * (valid until the next call with the same name from the same thread).
 * Names: "cpu_default" (main.cpp:63-131), "cuda_default" (main.cu:64-200),
 * "glass_stress", "random4096" (SURVEY §8d configs 4 and 5). */
int pt_builtin_scene(const char* name, uint32_t width, uint32_t height,
                     PtSceneSoA* scene, PtCamera* camera);

#ifdef __cplusplus
}
#endif
#endif /* PT_H_ */
