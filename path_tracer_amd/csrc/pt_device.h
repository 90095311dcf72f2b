// pt_device.h — device-side data layout shared by the C-ABI host code and the HIP kernels.
//
// Layout in HBM (see DESIGN.md §Data layout).  Everything the intersection loops read is
// indexed by a wave-uniform loop counter, so hipcc turns those reads into scalar loads
// (s_load_dwordx4 through the scalar data cache — CDNA's counterpart of __constant__ memory)
// and the operands live in SGPRs.  Per-lane gathers (the winning primitive's normal and
// material) are ordinary vector loads served by L1/L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- FP32 scene for the `path` kernel -------------------------------------------------
// Loop operands (sph, pl, tri) are padded with never-hit primitives to whole batches (4 spheres,
// 4 planes, 2 triangles) so each loop trip is one s_load_dwordx16 + one wait.
struct DevSceneF {
    const float4* sph;     // [ns4]    cx, cy, cz, r*r
    const float4* pl;      // [2*npg4] general planes: (unit normal, -(point . normal)) | (bits(primitive id), 0, 0, 0)
    const float4* axp;     // axis-aligned planes, one list per axis (x | y | z), two planes per float4:
                           //          p_k, bits(primitive id), p_k, bits(primitive id); NaN-padded
    const float4* pla;     // [np]     every plane in insertion order, general form (lane-parallel test)
    const float4* tri;     // [3*nt2]  rows of the world -> (u, v, height) affine map of the triangle
    const float4* rec;     // [4*(ns+np+nt)] winner record, one 64-B line per primitive:
                           //   0: sphere (cx,cy,cz,1/r) | plane, triangle (unit normal, 0)
                           //   1: albedo rgb, bits(material type)
                           //   2: roughness (clamped for Metal), metallic, ior, bits(light index | -1)
                           //   3: material emission rgb (colour*strength), 0
    const float4* lights;  // [3*nl]   (cx,cy,cz,r*r) | (shared-material emission, bits(sphere index))
                           //          | (makeEmissive colour*strength of that sphere, 0)
    int ns, np, nt, nl;     // np = all planes (primitive ids ns .. ns+np-1 in insertion order)
    int npg;                // general planes
    int nax, nay, naz;      // axis-aligned planes per axis
    int ax_off_y, ax_off_z; // float4 offsets of the y and z lists inside axp
};

struct DevCamF {
    float ox, oy, oz;
    float fx, fy, fz;  // forward = -w
    float rx, ry, rz;  // right   =  u
    float ux, uy, uz;  // up      =  v
    float half_w, half_h;
    float sx, sy;      // 2*half_w/width, 2*half_h/height
    float lens_radius; // aperture / 2 when use_dof (Camera.cu:20), else 0 = pinhole
    float focus_dist;
    uint32_t width, height;
};

struct PathArgs {
    float4* accum;
    uint32_t* rgba;          // display buffer (fused resolve) or nullptr
    uint64_t* counters;      // 12 x u64 (PtCounters order) or nullptr
    uint64_t* hist;          // 64 copies x 64 x u64: samples by the number of path rays they traced (instrumented builds) or nullptr
    uint32_t* work_counter;  // this launch's 8 ticket counters on separate 128-B lines (word 32*s), zero at start
    uint32_t* next_counter;  // the chain's other set: zeroed by this launch for the chain's next one
    float total_w, inv_total_w;  // (float)(s0 + count) and its reciprocal, for the fused display pack
    uint32_t s0, count;      // sample range of this launch
    uint32_t max_depth, rr_depth, seed_hash;
    uint32_t light_colour;   // PT_FLAG_LIGHT_COLOUR
    uint32_t acc_mode;       // PT_ACC_*
    uint32_t overwrite;      // != 0 (PT_ACC_SUM only): the buffer receives this launch's (sums, count), not buffer + them
    float mean_t;            // 1/(s0+1) for PT_ACC_RUNNING_MEAN
    uint32_t tiles_x, tiles_y;  // 8x8 pixel tiles
    float inv_tiles_x;
    uint32_t units_per_grab;    // minimum 4x4-pixel units per ticket
    uint32_t num_tiles;
    uint32_t width, height;     // frame size (filled by pt_launch_path from the camera)
    uint32_t few_max;           // lane-parallel intersection when at most this many lanes of a wave trace
    uint32_t regen_min;         // camera rays are prepared ahead in a batch once this many lanes lack one ...
    uint32_t idle_min;          // ... or once this many lanes sit idle for lack of one (or no lane traces)
    uint32_t prefetch_pixels;   // != 0: lanes take their next pixel while they still trace (launches with many pixels per lane)
    // A launch of several FRAMES (frames in flight, pt_api.cpp): frame f holds samples [s0 + f*frame_stride, ... + count)
    // of every pixel (frame_stride = count for one process; count * ranks when the samples of a step are split among
    // GPUs) and writes its sums to accum + f * plane_stride.  The work items are (frame, pixel) pairs from one
    // ticket space, frame-major.  A lane keeps (x, y, frame) of its next ray packed in one register: x in bits
    // [0, xw), y in [xw, xw + yw), frame in [jo, jo + jw); 16 / 16 / - for one frame; for several, as many bits as width and
    // height need and the rest for the frame.
    uint32_t frames, plane_stride, frame_stride;
    uint32_t xw, yw, jo, jw;
    // dynamic LDS image of the lane-parallel builds (laid out by pt_launch_path, in float4 units): spheres at 0, planes, triangle rows
    uint32_t lds_off_pla, lds_off_tri;
    uint64_t* trace;            // per-wave timeline, 16 x u64 per wave (PT_TRACE kernel builds only; else nullptr)
};

// ---- FP64 scene for the id pass and the `direct` integrator ---------------------------
struct DevSceneD {
    const double* sph;   // [ns*4]  cx, cy, cz, r
    const double* pl;    // [np*6]  px,py,pz, nx,ny,nz (unit)
    const double* tri;   // [nt*9]  v0, v1, v2
    const double* sph_em;     // [ns*3]
    const double* mat_albedo; // [nm*3]
    const uint32_t* sph_info; // [ns*3] mat, obj, emissive
    const uint32_t* pl_info;  // [np*2] mat, obj
    const uint32_t* tri_info; // [nt*2] mat, obj
    int ns, np, nt;
};

struct DevCamD {
    uint32_t model, width, height, pad;
    double origin[3];
    // PT_CAMERA_CPU: Camera::initialize products
    double top_left[3], horizontal[3], vertical[3];
    // PT_CAMERA_CUDA: float basis widened
    double fwd[3], right[3], up[3];
    double half_w, half_h;
};

// ---- static scratch of pt_render_noalloc (pt_path_kernel.hip) ---------------------------
// One array per GPU, 4 KB: two ticket-counter sets of the path kernel (8 shards x 128 B each).  The two limits are
// the entry point's stated maxima (include/pt.h); the counters themselves do not grow with the frame.
constexpr uint32_t PT_SCRATCH_BYTES = 4096;
// Resolves (once per device, with that device current) and returns the device address of this GPU's copy.
hipError_t pt_scratch_setup(int device, uint32_t** out);

// Per-wave record of the PT_TRACE kernel builds (tools/gpu_trace.py), in 8-byte words: 16 of totals, then a ring of the
// wave's last 256 trips as 32-bit entries (time since the wave's start in 10-ns ticks << 7 | lanes alive).
constexpr uint32_t PT_TRACE_WORDS = 16u + 128u;

// Most dynamic LDS a path-kernel workgroup (one wave) takes: 28 waves per CU x 5.5 KB stay within the CU's 160 KB, so
// LDS never costs a wave of occupancy.  The occupancy query is made with this figure.
constexpr uint32_t PT_LDS_MAX_BYTES = 5632;

// ---- launchers implemented in the .hip translation units ------------------------------
hipError_t pt_launch_path(const DevSceneF& sc, const DevCamF& cam, const PathArgs& a, bool nee, bool count, bool few,
                          uint32_t blocks, hipStream_t stream);
// Resident 256-thread blocks per CU of each path-kernel variant (occupancy query) x CU count.
hipError_t pt_path_resident_blocks(bool nee, bool count, bool few, int device, uint32_t* blocks);
hipError_t pt_launch_resolve(const float4* accum, void* out, uint32_t npix, uint32_t acc_mode, uint32_t mode,
                             hipStream_t stream);
// frames in flight: adds the sums of the `frames` frames of one overlapped launch (planes of its chain's buffer,
// plane_stride pixels apart) to the accumulation buffer, frame after frame; with newest != nullptr it makes every
// frame's display on the way: the last frame's into `newest`, frame f < frames - 1 into history + f * npix
constexpr uint32_t PT_MAX_CHAINS = 8;
// acc_mode = PT_ACC_*: sums are added; the running mean takes frame f (the caller's frame_count s0 + f * stride) with weight 1 / (frame_count + 1)
hipError_t pt_launch_fold(float4* accum, const float4* planes, uint32_t plane_stride, uint32_t frames, uint32_t* newest,
                          uint32_t* history, uint32_t npix, uint32_t acc_mode, uint32_t s0, uint32_t stride, hipStream_t stream);
hipError_t pt_launch_valu_peak(float* sink, uint32_t packed, uint32_t iters, uint32_t blocks, hipStream_t stream);
hipError_t pt_launch_primary_ids(const DevSceneD& sc, const DevCamD& cam, uint32_t* obj, uint32_t* mat,
                                 hipStream_t stream);
// counters != nullptr: the instrumented build, which adds samples / rays / hits / light samples / shadow rays to the PtCounters block
hipError_t pt_launch_direct(const DevSceneD& sc, const DevCamD& cam, float4* accum, uint32_t s0, uint32_t count,
                            uint32_t seed, uint64_t* counters, hipStream_t stream);
