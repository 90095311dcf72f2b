// pt_api.cpp — the C ABI of include/pt.h: contexts, scene/camera upload, render/resolve
// launches.  Host code only (compiled by hipcc for the HIP runtime API); the kernels live in
// pt_path_kernel.hip and pt_f64_kernels.hip.
//
// Replaces, on the reference side: initializeCuda / cudaMalloc of d_output (main.cu:28-57),
// MaterialFactory::createMaterialsOnDevice (MaterialFactory.cuh:41-75), the flatten +
// cudaMemcpyToSymbol upload (main.cu:214-278), initializeGPUData (raytracer_kernel.cu:308-321),
// launchRenderKernel (:325-327) and Window_t::updateScreen's device half (Window.cuh:147-165).
#include "../../include/pt.h"
#include "pt_device.h"
#include "pt_error.h"

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <new>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

thread_local std::string g_err;
thread_local PtContext* g_current = nullptr;

#define fail pt_fail

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

int pt_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
void pt_clear_error() { g_err.clear(); }

// PT_AMD_HOSTPROF=1: host time spent in the HIP calls of the overlapped launch path, printed at pt_destroy
struct HostProf {
    bool on = false;
    double t[6] = {}; uint64_t n[6] = {};
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
};
#define PROF(ctx, slot, expr) do { if ((ctx)->prof.on) { const double t0_ = HostProf::now(); HIP_TRY(expr); (ctx)->prof.t[slot] += HostProf::now() - t0_; (ctx)->prof.n[slot]++; } else HIP_TRY(expr); } while (0)

struct PtContext {
    HostProf prof;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // scene
    void* scene_arena = nullptr;
    bool have_scene = false;
    DevSceneF sf{};
    DevSceneD sd{};
    // camera / frame
    bool have_camera = false;
    DevCamF cf{};
    DevCamD cd{};
    uint32_t width = 0, height = 0;
    float4* accum_own = nullptr;
    float4* accum = nullptr;  // own or bound
    bool accum_bound = false;
    void* display = nullptr;  // 16 B / pixel: large enough for every resolve mode
    size_t display_bytes = 0;  // bytes the last resolve produced
    uint32_t last_acc_mode = PT_ACC_SUM;
    uint64_t* counters = nullptr;  // 12 x u64, then 64 copies of the 64 x u64 path-length histogram (one allocation)
    uint32_t* ids = nullptr;       // 2 * npix, allocated on demand
    float* sink = nullptr;
    uint32_t* work_counter = nullptr;  // ticket counters of the path kernel: 8 chains x 2 alternating sets x 8 shards x 128 B
    uint32_t work_parity[PT_MAX_CHAINS] = {};  // per chain: which of its two sets the next launch uses
    uint32_t* scratch = nullptr;               // this GPU's copy of the static scratch array (pt_render_noalloc), PT_SCRATCH_BYTES
    // frames in flight (pt_set_frames_in_flight): launch k of a burst goes to chain k % flight, which
    // has its own stream and its own accumulator; chain 0's accumulator is the context's (or the bound one)
    uint32_t flight = 1;
    bool own_burst = false;      // the chains are in use for a call that the library cut into frames by itself (render_in_parts on a default context)
    bool stream_chosen = false;  // pt_set_stream / pt_use_own_stream was called (launchRenderKernel otherwise moves to the null stream)
    // launchRenderKernel renders into the caller's `output`; with frames in flight the binding stays across consecutive
    // calls on the same buffer (so that they can be collected into one launch) and is undone by the next other entry point
    bool shim_active = false;
    float4* shim_saved = nullptr;
    bool shim_saved_bound = false;
    hipStream_t chain_stream[PT_MAX_CHAINS] = {};
    hipEvent_t chain_done[PT_MAX_CHAINS] = {};
    hipEvent_t fork_event = nullptr;
    float4* chain_acc[PT_MAX_CHAINS] = {};  // per chain: the sums of the chain's launch in flight (written whole by it, folded
                                            // into the accumulation buffer by the pass behind it)
    size_t chain_cap[PT_MAX_CHAINS] = {};   // per chain: planes (frames) its buffer holds; grown on demand up to frames_per_launch_max()
    hipEvent_t fold_done[PT_MAX_CHAINS] = {};  // per chain: the fold pass behind its last launch (and, through the chain of
                                               // waits, every earlier fold pass of the burst) is done
    bool display_current = false;           // the display buffer holds the mean of everything folded so far
    uint32_t folds_queued = 0;              // fold passes of the open burst enqueued so far (launches 0 .. folds_queued-1)
    bool fold_rgba[PT_MAX_CHAINS] = {};     // per chain: the launch whose fold pass is not enqueued yet asked for a display
    uint32_t chain_frames[PT_MAX_CHAINS] = {};  // per chain: frames of the launch whose fold pass is not enqueued yet
    uint32_t chain_s0[PT_MAX_CHAINS] = {}, chain_stride[PT_MAX_CHAINS] = {}, chain_mode[PT_MAX_CHAINS] = {};  // ... its first frame's sample
                                       // index, the distance between its frames, its PT_ACC_* mode (the running mean's weights)
    uint32_t* history = nullptr;       // displays of the frames before the newest of the last fold pass, one plane each
    size_t history_cap = 0;            // planes `history` holds; grown on demand
    uint32_t history_frames = 0;       // planes of `history` that hold a picture (frame history_frames-1 is the newest but one)
    // Frames collected for the next overlapped launch (pt_render calls that were accepted but not launched yet)
    struct Pending { uint32_t begin = 0, per_frame = 0, stride = 0, frames = 0; PtRenderParams prm{}; } pending;
    // Render-ahead (one frame in flight, the reference's loop: launch, show, launch): once two calls in a row have
    // continued one another -- same parameters, same sample count, equally spaced -- the frame that the NEXT call will
    // ask for if the loop goes on is launched as well, on one of two streams of its own, into a plane of its own.  If
    // the next call is that frame, its sums are there (or on their way) and the call only folds them in, exactly as a
    // frame in flight is folded; if it is anything else the plane is dropped.  The image is the serial loop's bit for
    // bit; what changes is that frame k + 1 fills the chip while frame k's last long paths drain and the host shows it.
    bool ahead_on = true;              // pt_set_render_ahead
    struct Ahead { bool valid = false; uint32_t begin = 0, count = 0, chain = 0, plane = 0; PtRenderParams prm{}; } ahead;
    struct Last { bool valid = false; uint32_t begin = 0, count = 0, stride = 0, streak = 0; PtRenderParams prm{}; } last;
    uint32_t ahead_next = 0;           // which of the two render-ahead chains the next such launch takes
    bool ahead_launched_on[2] = {false, false};    // per render-ahead chain: chain_done holds the end of a launch ahead
    // Each render-ahead chain has two planes and takes them in turn, so the kernel of frame k + 2 (same chain as frame k)
    // need not wait for the pass that folds frame k -- only for the one that folded frame k - 2 out of the plane it is
    // about to write, which is long done (a chain's kernels follow one another 36 us closer; the other chain's kernel fills most of that gap
    // anyway: 0.142 -> 0.138 ms per 720p frame).
    uint32_t ahead_plane[2] = {0, 0};                        // per chain: the plane its next launch writes
    hipEvent_t ahead_fold_ev[2][2] = {};                     // per chain and plane: the pass that last read the plane
    bool ahead_fold_pending[2][2] = {};                      // ... is recorded and not waited for yet
    bool ahead_fork[2] = {true, true}; // per render-ahead chain: its next launch is ordered behind the context's stream (uploads)
    uint64_t ahead_launched = 0, ahead_used = 0;   // pt_render_ahead_stats
    // A call that the library cuts into frames by itself (own_burst) is a burst of its own, joined before the call returns.
    // Consecutive such calls take the chains in turn (own_base rotates) and their kernels do not wait for the context's
    // stream -- which waits for the previous call's fold passes -- unless something they read was enqueued there since
    // (chain_needs_fork: scene, camera, frame size; a frame rendered ahead on a borrowed stream): a call's kernel fills
    // the chip while the previous call's last paths drain (720p in calls of 16 samples: 9.4 -> 10.6 Gsamples/s, glass_stress
    // 7.5 -> 8.1; with the context's stream among the chains 10.1 and 7.8).  The fold
    // passes keep their order: the first of a burst waits for the context's stream as it was when the burst began.
    uint32_t own_base = 0;
    bool chain_needs_fork[PT_MAX_CHAINS] = {true, true, true, true, true, true, true, true};
    bool ahead_dirty = false;          // a frame was rendered ahead since drop_ahead last ran
    uint32_t burst = 0;                // launches since the chains were last joined
    uint64_t path_launches = 0;        // launches of the path kernel since pt_create (pt_path_launches)
    bool burst_displayed = false;      // some launch of the burst asked for a display
    uint32_t compute_units = 256;
    uint32_t burst_chains = 3;         // most chains worth running at once: 3 on HIP's default 4 hardware queues, else
                                       // GPU_MAX_HW_QUEUES - 4 (up to 8); the cap of pt_set_frames_in_flight and the
                                       // size of a burst that pt_render opens by itself
    uint64_t* trace = nullptr;         // PT_AMD_TRACE=<file>: per-wave timeline of the last launch (PT_TRACE kernel builds)
    size_t trace_waves = 0;
    uint32_t resident_blocks[8] = {};  // per kernel build (nee, count, few)
    // tuning / tracing knobs for tools/gpu_ab.py and tools/gpu_trace.py, read once at pt_create
    // (PT_AMD_FEW_MAX, PT_AMD_MAX_BLOCKS, PT_AMD_GRID_PCT, PT_AMD_FRAMES, PT_AMD_CHUNK, PT_AMD_REGEN_MIN, PT_AMD_IDLE_MIN, PT_AMD_PREFETCH, PT_AMD_TRACE); -1 / empty = not set
    bool knob_own_chain0 = false;         // PT_AMD_OWN_CHAIN0=1 (A/B): the library's own bursts use chain 0 too
    bool knob_fail_chain_alloc = false;   // PT_AMD_FAIL_CHAIN_ALLOC=1 (tests): the chains' buffers cannot be allocated
    int knob_few_max = -1, knob_max_blocks = -1, knob_chunk = -1, knob_regen_min = -1, knob_idle_min = -1, knob_prefetch = -1, knob_grid_pct = -1, knob_frames = -1;
    std::string knob_trace;
};

namespace {

// First statement of every entry point that takes a context: clears the thread's error message (so it is
// empty after a call that succeeds) and makes the context's device current.
int join_chains(PtContext* ctx);
int queue_next_fold(PtContext* ctx);
int launch_pending(PtContext* ctx, bool joining = false);

// launchRenderKernel left the caller's `output` bound (frames in flight): fold what is held back into it and give the
// context its own buffer back.
int end_shim_binding(PtContext* ctx) {
    if (!ctx->shim_active) return PT_OK;
    const int rc = join_chains(ctx);
    ctx->accum = ctx->shim_saved; ctx->accum_bound = ctx->shim_saved_bound;
    ctx->shim_active = false;
    return rc;
}

// Render-ahead: forget the frame launched ahead (the scene, the camera or the frame size is about to change, or frames
// in flight take over).  What the context's stream does next is ordered behind the kernels still running ahead.
// The two chains (streams, planes, ticket counters) render-ahead uses are chains 1 and 2 of the frames in flight, which
// are never active at the same time: with chain 0 = the context's stream that is the three streams HIP's default of four
// hardware queues keeps apart (streams of their own shared a queue with the context's stream in a process that had used
// frames in flight before, and the frame rendered ahead then ran behind the pass it was to overlap).
constexpr uint32_t AHEAD_CHAIN_0 = 1u;
// `uploads`: the caller is about to change what kernels read (scene, camera, frame): the next kernel on every chain goes
// behind the context's stream.
int drop_ahead(PtContext* ctx, bool uploads = true) {
    for (uint32_t c = AHEAD_CHAIN_0; c < AHEAD_CHAIN_0 + 2u; ++c)
        if (ctx->chain_done[c]) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->chain_done[c], 0));   // (never recorded: no wait)
    ctx->ahead.valid = false;
    ctx->last.valid = false; ctx->last.streak = 0;
    ctx->ahead_fork[0] = ctx->ahead_fork[1] = true;
    if (uploads || ctx->ahead_dirty)
        for (bool& f : ctx->chain_needs_fork) f = true;
    ctx->ahead_dirty = false;
    return PT_OK;
}
// Chain of launch k of the open burst.
// (a burst of the library's own keeps off chain 0, the context's stream: a kernel there is behind the previous call's fold
//  passes whatever it waits for)
inline uint32_t chain_period(const PtContext* ctx) {   // launches after which the burst comes back to a chain
    return (ctx->own_burst && ctx->flight > 2u && !ctx->knob_own_chain0) ? ctx->flight - 1u : ctx->flight;
}
inline uint32_t chain_of(const PtContext* ctx, uint32_t k) {
    if (ctx->own_burst && ctx->flight > 2u && !ctx->knob_own_chain0) return 1u + (k + ctx->own_base) % (ctx->flight - 1u);
    return (k + (ctx->own_burst ? ctx->own_base : 0u)) % ctx->flight;
}

int use(PtContext* ctx) {
    pt_clear_error();
    if (!ctx) return fail(PT_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return end_shim_binding(ctx);
}

void free_chain_buffers(PtContext* ctx) {  // (hipFree synchronises the device: nothing queued still uses them)
    for (uint32_t c = 0; c < PT_MAX_CHAINS; ++c) {
        if (ctx->chain_acc[c]) { (void)hipFree(ctx->chain_acc[c]); ctx->chain_acc[c] = nullptr; }
        ctx->chain_cap[c] = 0;
    }
    if (ctx->history) { (void)hipFree(ctx->history); ctx->history = nullptr; }
    ctx->history_cap = 0;
    ctx->history_frames = 0;
}

int alloc_frame(PtContext* ctx, uint32_t w, uint32_t h) {
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return fail(PT_ERR_INVALID, "bad frame size %ux%u", w, h);
    if (w == ctx->width && h == ctx->height && ctx->display) return PT_OK;
    if (int rc = join_chains(ctx)) return rc;
    if (int rc = drop_ahead(ctx)) return rc;
    const size_t npix = (size_t)w * h;
    // the new buffers first, the swap only when both exist: a failed allocation leaves the old frame intact
    float4* accum_new = nullptr;
    void* display_new = nullptr;
    hipError_t e = hipMalloc((void**)&accum_new, npix * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&display_new, npix * 16);
    if (e == hipSuccess) e = hipMemsetAsync(accum_new, 0, npix * sizeof(float4), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(display_new, 0, npix * 16, ctx->stream);
    if (e != hipSuccess) {
        if (accum_new) (void)hipFree(accum_new);
        if (display_new) (void)hipFree(display_new);
        return fail(PT_ERR_HIP, "allocating a %ux%u frame failed: %s", w, h, hipGetErrorString(e));
    }
    // hipFree synchronises the device, so nothing queued still uses the old buffers when they go
    if (ctx->accum_own) (void)hipFree(ctx->accum_own);
    if (ctx->display) (void)hipFree(ctx->display);
    if (ctx->ids) { (void)hipFree(ctx->ids); ctx->ids = nullptr; }
    free_chain_buffers(ctx);  // sized for the old frame; the display history too (pt_history_frames() is 0 from here on)
    ctx->accum_own = accum_new;
    ctx->display = display_new;
    ctx->width = w;
    ctx->height = h;
    ctx->display_bytes = 0;
    ctx->accum_bound = false;  // a bound buffer was sized for the old frame
    ctx->accum = ctx->accum_own;
    return PT_OK;
}

constexpr size_t WORK_SET_WORDS = 256;  // 8 shards x 32 words (128 B apart)

// Frames in flight: enqueues the fold pass of the oldest launch of the burst that has none yet.  It runs behind its
// launch on the chain's own stream (no stream of its own: HIP's default is 4 hardware queues per process, and streams
// that share one serialise) and after the fold pass of the launch before it, whichever chain that ran on: the
// accumulation buffer receives the launches in launch order.  That cross-stream wait is the expensive call of the
// frame loop (hipStreamWaitEvent on an event still pending on another stream: 26 us of host time against 3 us for a
// kernel launch), so a fold pass is enqueued as late as it may be: when its chain is about to launch again, or when
// the burst is joined.  The first launches of a burst therefore leave the host 3 us apart and fill the chip at once.
int queue_next_fold(PtContext* ctx) {
    const uint32_t k = ctx->folds_queued, chain = chain_of(ctx, k);
    hipStream_t stream = ctx->chain_stream[chain];
    const uint32_t npix = ctx->width * ctx->height, frames = ctx->chain_frames[chain];
    if (k > 0) PROF(ctx, 2, hipStreamWaitEvent(stream, ctx->fold_done[chain_of(ctx, k - 1u)], 0));
    else if (stream != ctx->stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->fork_event, 0));   // the buffer as the context's stream left it when the burst began
    PROF(ctx, 3, pt_launch_fold(ctx->accum, ctx->chain_acc[chain], npix, frames, ctx->fold_rgba[chain] ? (uint32_t*)ctx->display : nullptr,
                                ctx->history, npix, ctx->chain_mode[chain], ctx->chain_s0[chain], ctx->chain_stride[chain], stream));
    PROF(ctx, 4, hipEventRecord(ctx->fold_done[chain], stream));
    ctx->display_current = ctx->fold_rgba[chain];
    if (ctx->fold_rgba[chain]) ctx->history_frames = frames - 1u;
    ctx->folds_queued++;
    return PT_OK;
}

// Frames in flight: makes the context's stream wait for the last fold pass (each waits for the one before it, so
// that one is behind all others).  A no-op when no burst is open.  Nothing is left to add up: every
// launch's sums are already in the accumulation buffer, in launch order.
int join_chains(PtContext* ctx) {
    if (int rc = launch_pending(ctx, true)) return rc;
    if (ctx->burst == 0) return PT_OK;
    while (ctx->folds_queued < ctx->burst)
        if (int rc = queue_next_fold(ctx)) return rc;
    const uint32_t last = chain_of(ctx, ctx->burst - 1u);
    if (ctx->chain_stream[last] != ctx->stream) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->fold_done[last], 0));  // (chain 0 is the context's stream itself)
    if (ctx->own_burst) ctx->own_base = (ctx->own_base + ctx->burst) % (ctx->flight * (ctx->flight - 1u));   // (a multiple of both moduli of chain_of)
    ctx->burst = 0;
    ctx->folds_queued = 0;
    ctx->ahead_fork[0] = ctx->ahead_fork[1] = true;   // render-ahead shares chains 1 and 2: its next launches go behind the fold passes
    if (ctx->burst_displayed && !ctx->display_current) {
        // some launch of the burst asked for a display and a later one did not: redo it from the sums
        HIP_TRY(pt_launch_resolve(ctx->accum, ctx->display, ctx->width * ctx->height, ctx->last_acc_mode, PT_RESOLVE_RGBA8_CUDA, ctx->stream));
        ctx->history_frames = 0;  // (the display no longer continues the history planes)
    }
    ctx->burst_displayed = false;
    return PT_OK;
}

// Work already queued on the old stream (frame clears, launches, resolves) stays ordered before
// whatever the caller queues on the new one.
int switch_stream(PtContext* ctx, hipStream_t to) {
    if (int rc = join_chains(ctx)) return rc;
    if (to == ctx->stream) return PT_OK;
    if (!ctx->fork_event) HIP_TRY(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctx->fork_event, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(to, ctx->fork_event, 0));
    ctx->stream = to;
    return PT_OK;
}

// Bits that hold 0 .. n-1.
uint32_t bits_for(uint32_t n) { uint32_t b = 1; while (b < 32u && (1ull << b) < n) ++b; return b; }

// Most frames one overlapped launch may hold: at most 64, at most 2 GiB of sums per chain (2^27 pixel-frames, which
// also keeps the kernel's ticket arithmetic below 2^23), and no more than fit beside x and y in the one register a
// lane keeps (x, y, frame) of its next ray in (any frame up to 8192 x 8192 leaves six bits or more).
uint32_t frames_per_launch_max(const PtContext* ctx) {
    if (ctx->width == 0 || ctx->height == 0) return 1u;
    const uint32_t used = bits_for(ctx->width) + bits_for(ctx->height);
    if (used > 31u) return 1u;
    const uint64_t npix = (uint64_t)ctx->width * ctx->height;
    uint64_t fit = (1ull << 27) / npix;
    if (fit > 64u) fit = 64u;   // (1 024 frames at 720p: 32 per launch 11.6 Gsamples/s, 64 12.0, 96-128 12.0)
    if (fit > (1ull << (32u - used))) fit = 1ull << (32u - used);
    if (ctx->knob_frames > 0 && fit > (uint64_t)ctx->knob_frames) fit = (uint64_t)ctx->knob_frames;  // PT_AMD_FRAMES
    return fit < 1u ? 1u : (uint32_t)fit;
}

// Streams and events of the chains.
int prepare_chains(PtContext* ctx) {
    if (!ctx->fork_event) HIP_TRY(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
    // chain 0 runs on the context's current stream (its own, or the caller's: pt_set_stream joins the chains before it
    // changes): one stream fewer to share the process's hardware queues (GPU_MAX_HW_QUEUES, 4 by default; two chains on
    // one queue serialise), and a burst of a single launch -- 20 frames are one -- starts and ends without a hop
    // between queues (two such hops cost 60 us of the 1.96 ms the driver's 20 frames took)
    // (the null stream is not used as a chain: it synchronises with every blocking stream of the process)
    ctx->chain_stream[0] = ctx->stream ? ctx->stream : ctx->own_stream;
    for (uint32_t c = 0; c < ctx->flight; ++c) {
        if (!ctx->chain_stream[c]) HIP_TRY(hipStreamCreateWithFlags(&ctx->chain_stream[c], hipStreamNonBlocking));
        if (!ctx->chain_done[c]) HIP_TRY(hipEventCreateWithFlags(&ctx->chain_done[c], hipEventDisableTiming));
        if (!ctx->fold_done[c]) HIP_TRY(hipEventCreateWithFlags(&ctx->fold_done[c], hipEventDisableTiming));
    }
    return PT_OK;
}

// The buffers one overlapped launch needs: `frames` planes of sums on its chain and, when it displays, frames - 1
// planes of display history.  Allocated on the chain's first launch, never for chains that are not used.  Size: a
// caller that asked for frames in flight (pt_set_frames_in_flight) has opted into their memory and gets the most a
// launch may hold at once, so that no launch of the frame loop ever reallocates (a 295-MB hipFree + hipMalloc inside
// the driver's 20 timed frames cost 0.8 ms of 2); a call that the library cuts into frames on its own (a default
// context, render_in_parts) takes what that call holds -- 16 planes for 16 samples, not 64 on every chain -- and
// grows on demand.  Buffers only grow, so a fold pass that is enqueued later finds the history at least as large as
// it needs.  The chain's previous fold pass is already enqueued when this runs (launch_now), and hipFree
// synchronises the device, so nothing still uses what is replaced.
// Returns PT_NO_MEMORY (> 0, internal) when the device has no room: the caller then renders the frames serially.
constexpr int PT_NO_MEMORY = 1;
int ensure_chain_buffers(PtContext* ctx, uint32_t chain, uint32_t frames, bool displayed, bool exact = false) {   // exact: the render-ahead chains hold one frame
    const size_t npix = (size_t)ctx->width * ctx->height;
    if (ctx->knob_fail_chain_alloc) return PT_NO_MEMORY;
    const uint32_t most = frames_per_launch_max(ctx);
    if (ctx->chain_cap[chain] < frames) {
        if (ctx->chain_acc[chain]) { HIP_TRY(hipFree(ctx->chain_acc[chain])); ctx->chain_acc[chain] = nullptr; ctx->chain_cap[chain] = 0; }
        uint32_t planes = (ctx->own_burst || exact || most < frames) ? frames : most;
        hipError_t e = hipMalloc((void**)&ctx->chain_acc[chain], (size_t)planes * npix * sizeof(float4));
        if (e != hipSuccess && planes > frames) { (void)hipGetLastError(); planes = frames; e = hipMalloc((void**)&ctx->chain_acc[chain], (size_t)planes * npix * sizeof(float4)); }
        if (e != hipSuccess) { (void)hipGetLastError(); ctx->chain_acc[chain] = nullptr; return PT_NO_MEMORY; }
        ctx->chain_cap[chain] = planes;   // written whole by every launch: no clear
    }
    if (displayed && frames > 1u && ctx->history_cap < frames - 1u) {
        if (ctx->history) { HIP_TRY(hipFree(ctx->history)); ctx->history = nullptr; ctx->history_cap = 0; ctx->history_frames = 0; }
        uint32_t planes = (ctx->own_burst || most < frames) ? frames - 1u : most - 1u;
        hipError_t e = hipMalloc((void**)&ctx->history, (size_t)planes * npix * sizeof(uint32_t));
        if (e != hipSuccess && planes > frames - 1u) { (void)hipGetLastError(); planes = frames - 1u; e = hipMalloc((void**)&ctx->history, (size_t)planes * npix * sizeof(uint32_t)); }
        if (e != hipSuccess) { (void)hipGetLastError(); ctx->history = nullptr; return PT_NO_MEMORY; }
        ctx->history_cap = planes;
    }
    return PT_OK;
}

struct F3 { float x, y, z; };
inline F3 fnorm(F3 a) {  // Vec3.cuh:49-61, float
    float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    if (l > 0) { float il = 1 / l; a.x *= il; a.y *= il; a.z *= il; }
    return a;
}
inline F3 fcross(F3 u, F3 v) { return F3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }

struct D3h { double x, y, z; };
inline D3h dnorm(D3h a) {  // Vec3.h:48-63
    double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (l > 0) { double il = 1 / l; a.x *= il; a.y *= il; a.z *= il; }
    return a;
}
inline D3h dcross(D3h u, D3h v) { return D3h{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
inline D3h dmul(D3h a, double t) { return D3h{a.x * t, a.y * t, a.z * t}; }
inline D3h dadd(D3h a, D3h b) { return D3h{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3h dsub(D3h a, D3h b) { return D3h{a.x - b.x, a.y - b.y, a.z - b.z}; }

inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline uint32_t hash32(uint32_t x) {  // counter-RNG finaliser, DESIGN.md §RNG
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}


// Frames in flight hold sums and the reference's running mean alike: a launch writes every frame's own value into a
// plane, and the fold pass applies the recurrence -- acc += frame, or acc = frame * t + acc * (1 - t) with
// t = 1 / (frame_count + 1), raytracer_kernel.cu:297-304 -- frame after frame in launch order.
bool can_overlap(const PtContext* /*ctx*/, const PtRenderParams* /*prm*/) {
    return true;
}

// Samples per frame when a call for several samples is cut into frames of a multi-frame launch (0: never cut).
// A pixel's samples within a frame belong to one lane; frames of one or two samples make (frame, pixel) items small
// enough to balance the lanes at the end of the launch, and 32 frames fill a launch.  Measured at 1280x720, one
// call, Gsamples/s by samples per frame (tools/chunk_sweep.py):  16 spp: 1: 9.6, 2: 9.4, 4: 8.9, 8: 7.9, whole: 5.8;
// 64 spp: 1-3: 11.8, 8: 11.2, 32: 8.6;  256 spp: 1: 12.2, 2-8: 12.5, 32: 11.7, whole: 6.3.
uint32_t part_size(const PtContext* ctx, uint32_t sample_count) {
    if (ctx->knob_chunk >= 0) return (uint32_t)ctx->knob_chunk;
    const uint32_t c = (sample_count + 31u) / 32u;
    return c < 1u ? 1u : (c > 4u ? 4u : c);
}

// A pixel belongs to one lane for the whole launch, so one launch with many samples per pixel and few pixels per
// lane (2.3 at 720p) ends with most lanes idle while the last pixels finish.  On frames of at most 2^22 pixels a
// PT_ACC_SUM path call for several samples is therefore rendered as frames of 1..4 samples (4K launches hold 21
// pixels per lane and are left whole), from 4 samples on (below that one serial launch is as fast).  With frames in
// flight those frames join the open burst like any other; a call made without is a burst of its own.  The cut
// depends on the call's arguments and the frame size only -- not on the frames in flight -- so the image is
// the same bit for bit either way.
bool wants_parts(const PtContext* ctx, uint32_t sample_count, const PtRenderParams* prm) {
    if (prm->acc_mode != PT_ACC_SUM || (uint64_t)ctx->width * ctx->height > (1ull << 22)) return false;
    const uint32_t part = part_size(ctx, sample_count);
    return part > 0 && sample_count > part && sample_count >= 4u;
}

int launch_path(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* prm, float4* into = nullptr);
int launch_serial(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* prm);

// With frames in flight the launches join the open burst; without, the call is a burst of its own on
// `burst_chains` chains, folded before it returns, and a fused display is made from the folded sums.  The
// partition depends only on the call's arguments: reproducible, and equal to the single launch up to float
// summation order.
int render_in_parts(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* prm) {
    const bool pipelined = ctx->flight > 1;
    const uint32_t part = part_size(ctx, sample_count);
    const uint32_t parts = (sample_count + part - 1u) / part;
    const uint32_t user_flight = ctx->flight;
    PtRenderParams sub = *prm;
    if (!pipelined) {
        if (int rc = drop_ahead(ctx, false)) return rc;   // the burst takes the chains that render-ahead uses
        ctx->flight = ctx->burst_chains;
        ctx->own_burst = true;
        sub.fused_rgba8 = 0;
    }
    uint32_t begin = sample_begin, left = sample_count;
    int rc = PT_OK;
    for (uint32_t k = 0; k < parts && rc == PT_OK; ++k) {
        const uint32_t n = (left + (parts - k) - 1u) / (parts - k);  // as equal as integers allow
        rc = launch_path(ctx, begin, n, &sub);
        begin += n; left -= n;
    }
    if (!pipelined) {
        const int rj = join_chains(ctx);
        ctx->flight = user_flight;
        ctx->own_burst = false;
        if (rc == PT_OK) rc = rj;
        if (rc == PT_OK && prm->fused_rgba8) rc = pt_resolve(ctx, PT_RESOLVE_RGBA8_CUDA);
    }
    return rc;
}

// ---- Launch tuning table ------------------------------------------------------------------------------------------
// Every fitted constant of launch_now() in one place, with the measurement it came from.  The sweeps are reproducible:
// tools/heuristics_sweep.py (few_max, regen_min, idle_min, grid, prefetch on scenes of 19 / 150 / 1 000 / 4 097
// primitives, serial and in flight; profiles/r2_heuristics_sweep.json, profiles/r3_heuristics_sweep.json),
// tools/chunk_sweep.py (samples per frame of a large call), tools/bench_sweep.sh (chains, frames per launch; it also runs
// the 150- and 1 000-primitive scenes, so a kernel change that moves an optimum shows as a number).  The guard
// tests/test_y_heuristics_gpu.py fails when a default falls 20 % behind the best setting of a small sweep (round 3,
// medians: every default within 4 % of the best at all four scene sizes).
namespace tune {
// Cost of one pass of the 64-ray intersection loop, in VALU instructions per primitive (static counts of `make isa`,
// tools/isa_sections.py: a sphere 11 + 12 for the root block, an axis plane 6, a general plane 13, a triangle 26-35 with
// its share of the batch overhead).  Only ratios matter: the sum is compared with the lane-parallel form's cost per ray.
constexpr uint32_t COST_SPHERE = 25, COST_AXIS_PLANE = 7, COST_PLANE = 13, COST_TRIANGLE = 35;
// Cost per ray of the lane-parallel closest hit: broadcast + reduction, one test per 64 primitives of a kind, and the
// unhidden wait for every 64 primitives that come from global memory instead of LDS (the first two such waits cost more
// than the later ones, which overlap).  Fitted on the few_max sweep: best threshold 3 / 14 / 28 / 30 rays at 19 / 150 /
// 1 000 / 4 097 primitives; a model without the waits gave 3 / 27 / 48 / 48 and lost 25 % at 150 primitives.
constexpr uint32_t FEW_BASE = 40, FEW_PER_64_SPHERES = 27, FEW_PER_64_PLANES = 15, FEW_PER_64_TRIANGLES = 37;
constexpr uint32_t FEW_GLOBAL_WAIT_FIRST = 60, FEW_GLOBAL_WAIT_NEXT = 25;
constexpr uint32_t FEW_MAX_CAP = 32;            // beyond half a wave the 64-ray loop always wins
// The build with the lane-parallel mode costs the common loop 3-4 %: it is left out of launches that overlap others or
// give every lane >= 100 samples (4K / 16 spp: -3 % with it) unless the scene is large enough for a threshold of >= 6 rays
// (serial 720p / 1 spp: +15 % with it; 150+ primitives: +30...70 % everywhere).
constexpr uint32_t FEW_LONG_LAUNCH_SAMPLES_PER_LANE = 100, FEW_MIN_THRESHOLD_WHEN_OVERLAPPED = 6;
constexpr uint32_t LANES_PER_CU_AT_6_WAVES = 6u * 256u;
// Grid of overlapped launches.  The chains' grids together are ~1.35x the resident capacity (27 / 20 of it, divided among
// the chains); a launch with >= 14 samples per lane gets as many waves as that allows; a sample of a large scene counts as
// serial_cost / 400 Cornell samples (random4096 at 720p: 8 x 259 blocks 168 Msamples/s, 4 x 518 226).  Measured at 720p /
// 1 spp, Gsamples/s: 4 x 518 blocks 8.5, 6 x 345 9.3, 8 x 259 9.5, 8 x 514 8.6; 4K / 16 spp: 8 x 1536 11.0, 8 x 400 8.6.
constexpr uint32_t GRID_FILL_NUM = 27, GRID_FILL_DEN = 20, GRID_SAMPLES_PER_LANE = 14, GRID_COST_OF_A_CORNELL_SAMPLE = 400;
// A launch of many frames: a quarter more waves than are resident (the surplus starts as slots come free and finds
// tickets left): 20 frames at 720p +1.5 %, 1 024 frames +0.4 %; 150 % the same, 75 % -4 %.
constexpr uint32_t MULTI_FRAME_SURPLUS_DIV = 4;
// Tickets: aim at >= 8 per wave of the grid, at most 16 / 8 / 4 units of 4x4 pixels each (launches of many frames / of one
// or two samples / others).  720p, 1 spp, 8 chains: 64 pixels per ticket 10.0, 128-256 10.15, 384+ 9.96 Gsamples/s.
constexpr uint32_t TICKETS_PER_WAVE = 8, GRAB_MOST_MULTI_FRAME = 16, GRAB_MOST_FEW_SAMPLES = 8, GRAB_MOST = 4;
// Regeneration: the batch runs once 32 lanes lack a prepared ray (4K / 16 spp: 24-48 within 1 %, 8 loses 5 %) or idle_min
// lanes sit idle; in one-sample launches idle_min = min(12, 4800 / serial_cost): 12 idle lanes wait for the batch on the
// Cornell box (+2 %, serial +5 %), none on 4 096 spheres (12 lanes idle through that loop lose 6 %).
constexpr uint32_t REGEN_MIN = 32, IDLE_MIN_MOST = 12, IDLE_MIN_COST = 4800;
// Taking the next PIXEL ahead of time only with >= 8 (frame, pixel) pairs per lane, or in overlapped launches of several
// samples on small scenes (720p, 16 spp, 8 in flight: +3 %; serial 720p / 1 spp: -13 %; random4096: -7...-19 %).
constexpr uint32_t PREFETCH_PAIRS_PER_LANE = 8, PREFETCH_SMALL_SCENE_COST = 800;
}  // namespace tune

// One launch of the path kernel (arguments validated by pt_render): `frames` frames of sample_count samples each,
// frame f = samples [sample_begin + f * frame_stride, sample_begin + f * frame_stride + sample_count).
//  * pipelined: a launch of the open burst on the next chain; it writes one plane of sums per frame into the chain's
//    buffer and the fold pass behind it adds them to the accumulation buffer in order.
//  * otherwise (frames == 1): a serial launch on the context's stream that adds to the accumulation buffer itself.
//    `into` != nullptr is pt_render_noalloc's launch: into the caller's buffer, with the ticket counters in the device's
//    static scratch array; nothing on that path allocates, queries or creates anything.
//  * ahead_chain >= 0 (frames == 1, not pipelined): the render-ahead launch of a frame nobody has asked for yet, on that
//    chain's stream into that chain's plane, like a pipelined launch but outside the burst's book-keeping: no fold pass
//    is scheduled (the call that turns out to want the frame enqueues it, launch_path), chain_done[chain] marks its end.
int launch_now(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, uint32_t frames, uint32_t frame_stride,
               const PtRenderParams* prm, float4* into, bool pipelined, int ahead_chain = -1) {
    const bool noalloc = into != nullptr;
    const bool ahead = ahead_chain >= 0;
    const bool overlapped = pipelined || ahead;                       // shares the chip with another launch
    const uint32_t overlap = pipelined ? ctx->flight : (ahead ? 2u : 1u);
    uint32_t chain = 0;
    hipStream_t stream = ctx->stream;
    if (ahead) {
        chain = (uint32_t)ahead_chain;
        if (!ctx->fork_event) HIP_TRY(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
        // Streams: no more than three with work on them, the number HIP's default of four hardware queues keeps apart (a
        // fourth shares a queue with one of them, and a frame rendered ahead on the queue of the context's stream runs
        // behind the pass it was to overlap: 0.25 ms per frame instead of 0.14).  A context on its own stream uses chains 1
        // and 2; one that runs on the caller's stream leaves its own stream idle, and that takes chain 2's place.
        const bool borrow_own = chain == AHEAD_CHAIN_0 + 1u && ctx->own_stream && ctx->stream != ctx->own_stream;
        if (!borrow_own && !ctx->chain_stream[chain]) HIP_TRY(hipStreamCreateWithFlags(&ctx->chain_stream[chain], hipStreamNonBlocking));
        if (!ctx->chain_done[chain]) HIP_TRY(hipEventCreateWithFlags(&ctx->chain_done[chain], hipEventDisableTiming));
        stream = borrow_own ? ctx->own_stream : ctx->chain_stream[chain];
        // the first such launch after a change of scene, camera or frame goes behind what the context's stream holds (the
        // uploads among it); the later ones must NOT wait for that stream -- it holds the previous frame's fold pass, which
        // waits for the previous frame's kernel, the very kernel this one is to overlap
        // behind the chain's previous launch ahead, whatever stream that ran on (a frame that was dropped has no pass behind
        // it to wait for, and the context may have changed streams since: two launches on one plane and one set of ticket
        // counters must not overlap)
        if (ctx->ahead_launched_on[chain - AHEAD_CHAIN_0]) HIP_TRY(hipStreamWaitEvent(stream, ctx->chain_done[chain], 0));
        // the plane is written again: not before the pass that took the chain's previous frame out of it has run
        const uint32_t ci = chain - AHEAD_CHAIN_0, plane = ctx->ahead_plane[ci];
        if (ctx->ahead_fold_pending[ci][plane]) {
            HIP_TRY(hipStreamWaitEvent(stream, ctx->ahead_fold_ev[ci][plane], 0));
            ctx->ahead_fold_pending[ci][plane] = false;
        }
        if (ctx->ahead_fork[chain - AHEAD_CHAIN_0]) {
            HIP_TRY(hipEventRecord(ctx->fork_event, ctx->stream));
            HIP_TRY(hipStreamWaitEvent(stream, ctx->fork_event, 0));
            ctx->ahead_fork[chain - AHEAD_CHAIN_0] = false;
        }
        if (int rc = ensure_chain_buffers(ctx, chain, 2u, false, true)) return rc;   // two planes (PT_NO_MEMORY: the caller simply does not render ahead)
    }
    if (pipelined) {
        if (ctx->burst == 0) {
            if (int rc = prepare_chains(ctx)) return rc;
            HIP_TRY(hipEventRecord(ctx->fork_event, ctx->stream));
        }
        chain = chain_of(ctx, ctx->burst);
        stream = ctx->chain_stream[chain];
        // (a chain's launches and fold passes alternate on its stream: its buffer is folded before it is written again)
        // first launch of this chain in the burst: behind the context's stream -- always in a caller's burst, only when
        // something the kernel reads has changed in a burst that the library made of one call (PtContext::own_base)
        if (ctx->burst < chain_period(ctx) && stream != ctx->stream && (!ctx->own_burst || ctx->chain_needs_fork[chain])) {
            PROF(ctx, 0, hipStreamWaitEvent(stream, ctx->fork_event, 0));
            ctx->chain_needs_fork[chain] = false;
        }
        // (a chain launches again: the pass that folds its previous launch goes first -- `period` launches back)
        const uint32_t period = chain_period(ctx);
        while (ctx->burst >= period && ctx->folds_queued + period <= ctx->burst)
            if (int rc = queue_next_fold(ctx)) return rc;
        if (int rc = ensure_chain_buffers(ctx, chain, frames, prm->fused_rgba8 != 0)) return rc;  // (PT_NO_MEMORY: launch_pending renders serially)
    }
    PathArgs a{};
    // A launch that overlaps others writes its own sums (sum_r, sum_g, sum_b, count) whole into its chain's buffer;
    // the fold pass behind it adds them to the accumulation buffer and makes the display (pt_fold_kernel).
    a.accum = noalloc ? into : (overlapped ? ctx->chain_acc[chain] + (ahead ? (size_t)ctx->ahead_plane[chain - AHEAD_CHAIN_0] * ctx->width * ctx->height : 0u) : ctx->accum);
    a.overwrite = overlapped ? 1u : 0u;
    a.rgba = (prm->fused_rgba8 && !overlapped) ? (uint32_t*)ctx->display : nullptr;
    if (a.rgba) ctx->history_frames = 0;  // (the display no longer continues the history planes)
    a.counters = (prm->flags & PT_FLAG_COUNTERS) ? ctx->counters : nullptr;
    a.hist = a.counters ? ctx->counters + 12 : nullptr;
    a.s0 = sample_begin; a.count = sample_count;
    a.frames = frames; a.plane_stride = ctx->width * ctx->height; a.frame_stride = frame_stride;
    // (x, y, frame) of a lane's next ray in one register: as many bits as each needs (a frame is at most 2^28 pixels, so
    // x and y fit in 30 bits however long and thin it is); no bits for the frame number of a one-frame launch
    a.xw = bits_for(ctx->width); a.yw = bits_for(ctx->height); a.jo = a.xw + a.yw; a.jw = frames > 1u ? 32u - a.jo : 0u;
    a.max_depth = prm->max_depth; a.rr_depth = prm->rr_depth; a.seed_hash = hash32(prm->seed);
    a.light_colour = (prm->flags & PT_FLAG_LIGHT_COLOUR) ? 1u : 0u;
    a.acc_mode = prm->acc_mode;
    a.mean_t = 1.0f / (float)(sample_begin + 1u);  // :299
    a.tiles_x = (ctx->width + 7u) / 8u; a.tiles_y = (ctx->height + 7u) / 8u;
    a.num_tiles = a.tiles_x * a.tiles_y;
    const bool nee = (prm->flags & PT_FLAG_NEE_MIS) != 0, cnt = (prm->flags & PT_FLAG_COUNTERS) != 0;
    // Lane-parallel intersection for waves with few rays pays while (rays in flight) x (per-ray cost) < cost of
    // one 64-ray pass.  The kernel build that contains it costs the common loop 2-3 %, so it is used where
    // waves drain alone (short launches that do not overlap others) or the scene is large (threshold >= 6 rays).
    const DevSceneF& f = ctx->sf;
    const uint32_t serial_cost = tune::COST_SPHERE * (uint32_t)f.ns + tune::COST_AXIS_PLANE * (uint32_t)(f.nax + f.nay + f.naz) +
                                 tune::COST_PLANE * (uint32_t)f.npg + tune::COST_TRIANGLE * (uint32_t)f.nt;
    {
        // per ray in the lane-parallel form: the broadcast and the reduction, one test per 64 primitives of a kind, and
        // the wait for every 64 beyond the first of a kind, which come from global memory instead of LDS with no other
        // work to hide the load behind (fitted on tools/heuristics_sweep.py: best threshold 3 / 14 / 28 / 30 rays at
        // 19 / 150 / 1000 / 4097 primitives; the first model, without the waits, gave 3 / 27 / 48 / 48 and lost 25 %
        // at 150 primitives, where 24 rays already cost more than one pass of the serial loop)
        const uint32_t cs = ((uint32_t)f.ns + 63u) / 64u, cp = ((uint32_t)f.np + 63u) / 64u, ct = ((uint32_t)f.nt + 63u) / 64u;
        const uint32_t from_global = (cs > 1u ? cs - 1u : 0u) + (cp > 1u ? cp - 1u : 0u) + (ct > 1u ? ct - 1u : 0u);
        const uint32_t per_ray = tune::FEW_BASE + tune::FEW_PER_64_SPHERES * cs + tune::FEW_PER_64_PLANES * cp + tune::FEW_PER_64_TRIANGLES * ct +
                                 tune::FEW_GLOBAL_WAIT_FIRST * (from_global < 2u ? from_global : 2u) + tune::FEW_GLOBAL_WAIT_NEXT * (from_global > 2u ? from_global - 2u : 0u);
        const uint32_t few = serial_cost / per_ray;
        a.few_max = few > tune::FEW_MAX_CAP ? tune::FEW_MAX_CAP : few;
        // (serial launches: 720p, 1 spp +15 % with it; 4K, 16 spp, 340 samples per resident lane, -3 %)
        const uint64_t per_lane = ((uint64_t)ctx->width * ctx->height * sample_count * frames) / ((uint64_t)ctx->compute_units * tune::LANES_PER_CU_AT_6_WAVES);
        if ((overlapped || per_lane >= tune::FEW_LONG_LAUNCH_SAMPLES_PER_LANE) && a.few_max < tune::FEW_MIN_THRESHOLD_WHEN_OVERLAPPED) a.few_max = 0u;
        if (ctx->knob_few_max >= 0) a.few_max = (uint32_t)ctx->knob_few_max;
    }
    const bool few = a.few_max > 0u;
    uint32_t& resident = ctx->resident_blocks[(nee ? 4 : 0) + (cnt ? 2 : 0) + (few ? 1 : 0)];
    if (resident == 0) {  // (pt_create fills the table; this is the fall-back if its query failed)
        if (noalloc) return fail(PT_ERR_STATE, "the occupancy of this kernel build was not established at pt_create");
        HIP_TRY(pt_path_resident_blocks(nee, cnt, few, ctx->device, &resident));
    }
    a.inv_tiles_x = 1.0f / (float)a.tiles_x;
    a.total_w = (float)(sample_begin + sample_count); a.inv_total_w = 1.0f / a.total_w;
    // Grid: no more waves than are resident at once.  Overlapping launches fill the chip together, so each
    // can be smaller: fewer waves means more pixels per lane (lanes that regenerate instead of idling until
    // the launch ends).  The chains' grids together are ~1.35x the resident capacity; a launch with enough
    // work to keep its lanes regenerating for long (>= 14 samples per lane) gets as many waves as that allows.
    // (measured at 720p, 1 spp, Gsamples/s: 4 x 518 blocks 8.5, 6 x 345 9.3, 8 x 259 9.5, 8 x 514 8.6;
    //  4K, 16 spp: 8 x 1536 11.0, 8 x 400 8.6)
    // (Long launches on more chains than the ~4 the command processor keeps active at once underfill the chip
    //  with small grids: `random4096` at 720p, 8 x 259 blocks 168 Msamples/s, 4 x 518 226.  A sample of a
    //  large scene therefore counts as many: its brute-force cost relative to the Cornell box's.)
    uint32_t cap = resident;
    if (overlapped) {
        const uint64_t weight = serial_cost > tune::GRID_COST_OF_A_CORNELL_SAMPLE ? serial_cost / tune::GRID_COST_OF_A_CORNELL_SAMPLE : 1u;
        const uint64_t by_work = ((uint64_t)ctx->width * ctx->height * sample_count * frames * weight) / (256u * tune::GRID_SAMPLES_PER_LANE);
        // (the factor is against 6 waves/SIMD: a build that fits 7 has the same optimum in blocks)
        const uint32_t six = ctx->compute_units * 6u;
        const uint32_t by_fill = ((resident < six ? resident : six) * tune::GRID_FILL_NUM) / (tune::GRID_FILL_DEN * overlap);
        const uint32_t want = by_work > by_fill ? (by_work > resident ? resident : (uint32_t)by_work) : by_fill;
        if (want < cap) cap = want;
        // (a launch of many frames: a quarter more waves than are resident -- the surplus starts as slots come free and
        //  finds tickets left: 20 frames at 720p +1.5 %, 1024 frames +0.4 %; 150 % the same, 75 % -4 %)
        if (frames > 1u && cap == resident) cap = resident + resident / tune::MULTI_FRAME_SURPLUS_DIV;
    }
    // work unit = 4x4 pixels; a ticket is at least units_per_grab of them.  Default: aim at >= 8
    // tickets per wave of the grid, but never more than one 8x8 tile per ticket -- two for launches of one
    // or two samples, where the atomics count (720p, 1 spp, 8 chains: 64 pixels per ticket 10.0, 128-256
    // 10.15, 384 and more 9.96 Gsamples/s; 4K, 16 spp: 64 pixels 11.73, 128 11.63).
    const uint32_t units = a.num_tiles * 4u * frames;
    uint32_t upg;
    if (prm->pixels_per_wave) upg = (prm->pixels_per_wave + 15u) / 16u;
    else {
        const uint32_t most = frames > 1u ? tune::GRAB_MOST_MULTI_FRAME : (sample_count <= 2u ? tune::GRAB_MOST_FEW_SAMPLES : tune::GRAB_MOST);
        upg = units / (cap * 4u * tune::TICKETS_PER_WAVE);
        upg = upg > most ? most : upg;
    }
    a.units_per_grab = upg < 1u ? 1u : (upg > 65536u ? 65536u : upg);
    uint32_t blocks = (units / a.units_per_grab + 3u) / 4u;  // no more waves than tickets ...
    if (blocks > cap) blocks = cap;                          // ... and no more than the cap
    // Regeneration (pt_path_kernel.hip, end of the loop body).  Camera rays are prepared one ahead, in batches: the
    // batch runs once 32 lanes of a wave lack a prepared ray (4K, 16 spp: 24-48 all within 1 %, 8 loses 5 %).  Taking the
    // next PIXEL ahead of time is right only when a lane has many pixels to go through: at 720p a resident lane gets 2.3
    // pixels per launch, and a pixel held in reserve by a lane that is busy with a long path is a pixel an idle lane
    // cannot take (-13 % on serial 720p / 1 spp launches, -19 % on `random4096`).  Without it a lane that ends a
    // pixel waits for its next one; in one-sample launches of small scenes every path end is such a wait, and the
    // pull + camera ray (~150 instructions) then waits for 12 idle lanes (+2 %, serial +5 %) -- unless a trip of
    // waiting costs more than that (large scenes: 12 lanes idle through a 4096-sphere loop lose 6 %).
    a.regen_min = tune::REGEN_MIN;
    a.idle_min = 1u;
    if (sample_count == 1u) { const uint32_t r = tune::IDLE_MIN_COST / (serial_cost ? serial_cost : 1u); a.idle_min = r > tune::IDLE_MIN_MOST ? tune::IDLE_MIN_MOST : r; }
    // (launches that overlap others lose less to a withheld pixel -- another launch's lanes fill in: 720p, 16 spp, 8 in
    //  flight +3 % with it, `random4096` -7 %)
    a.prefetch_pixels = ((uint64_t)ctx->width * ctx->height * frames >= (uint64_t)cap * 256u * tune::PREFETCH_PAIRS_PER_LANE ||
                         (overlapped && sample_count > 1u && serial_cost <= tune::PREFETCH_SMALL_SCENE_COST)) ? 1u : 0u;
    if (ctx->knob_regen_min >= 0) a.regen_min = (uint32_t)ctx->knob_regen_min;
    if (ctx->knob_idle_min >= 0) a.idle_min = (uint32_t)ctx->knob_idle_min;
    if (ctx->knob_prefetch >= 0) a.prefetch_pixels = (uint32_t)ctx->knob_prefetch;
    if (a.regen_min < 1u) a.regen_min = 1u;
    if (a.idle_min < 1u) a.idle_min = 1u;
    if (ctx->knob_max_blocks > 0 && blocks > (uint32_t)ctx->knob_max_blocks) blocks = (uint32_t)ctx->knob_max_blocks;
    if (ctx->knob_grid_pct > 0) blocks = (uint32_t)((uint64_t)blocks * (uint32_t)ctx->knob_grid_pct / 100u);  // PT_AMD_GRID_PCT: the sweep of tests/ and tools/
    if (blocks < 8u) blocks = 8u;                            // every shard needs a block to serve its tiles
    // launches of one chain are ordered on its stream (chain 0 also serves unpipelined launches, which
    // join_chains() orders after it): each zeroes the set its successor on the same chain will use
    a.work_counter = ctx->work_counter + WORK_SET_WORDS * (2u * chain + ctx->work_parity[chain]);
    a.next_counter = ctx->work_counter + WORK_SET_WORDS * (2u * chain + (ctx->work_parity[chain] ^ 1u));
    if (noalloc) {
        // The static array is shared by every context on this GPU and keeps the last call's contents: this call's
        // counter set is cleared here, on the stream, and the "next" set the kernel zeroes is the array's second
        // half, which nobody relies on.  (A memset of >= 256 B on the launch's own stream is seen as zero by the
        // kernel's atomics: MI355X_MICROARCH.md, zeroing across a kernel boundary.)
        a.work_counter = ctx->scratch;
        a.next_counter = ctx->scratch + WORK_SET_WORDS;
        HIP_TRY(hipMemsetAsync(a.work_counter, 0, WORK_SET_WORDS * sizeof(uint32_t), stream));
    }
    const char* trace_file = (ctx->knob_trace.empty() || noalloc || ahead) ? nullptr : ctx->knob_trace.c_str();  // kernel-timeline experiments (tools/gpu_trace.py)
    if (trace_file) {
        const size_t waves = (size_t)blocks * 4u;
        if (ctx->trace_waves < waves) {
            if (ctx->trace) HIP_TRY(hipFree(ctx->trace));
            HIP_TRY(hipMalloc((void**)&ctx->trace, waves * PT_TRACE_WORDS * 8u));
            ctx->trace_waves = waves;
        }
        HIP_TRY(hipMemsetAsync(ctx->trace, 0, waves * PT_TRACE_WORDS * 8u, stream));
        a.trace = ctx->trace;
    } else a.trace = nullptr;
    PROF(ctx, 1, pt_launch_path(ctx->sf, ctx->cf, a, nee, cnt, few, blocks, stream));
    if (ahead) {   // counted as a launch when a call takes the frame (launch_path); nothing else of the context's state moves
        HIP_TRY(hipEventRecord(ctx->chain_done[chain], stream));
        ctx->ahead_launched_on[chain - AHEAD_CHAIN_0] = true;
        ctx->ahead_dirty = true;
        ctx->ahead_plane[chain - AHEAD_CHAIN_0] ^= 1u;   // (launch_serial notes which plane this frame went to before the next launch moves on)
        ctx->work_parity[chain] ^= 1u;
        ctx->ahead_launched++;
        return PT_OK;
    }
    ctx->path_launches++;
    if (pipelined) {
        ctx->chain_frames[chain] = frames;
        ctx->chain_s0[chain] = sample_begin; ctx->chain_stride[chain] = frame_stride; ctx->chain_mode[chain] = prm->acc_mode;
        ctx->fold_rgba[chain] = prm->fused_rgba8 != 0;   // its fold pass is enqueued later (queue_next_fold)
        if (prm->fused_rgba8) ctx->burst_displayed = true;
        ctx->burst++;
    }
    if (trace_file) {
        std::vector<uint64_t> host((size_t)blocks * 4u * PT_TRACE_WORDS);
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(host.data(), ctx->trace, host.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(trace_file, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    }
    if (!noalloc) ctx->work_parity[chain] ^= 1u;
    ctx->last_acc_mode = prm->acc_mode;
    if (prm->fused_rgba8) ctx->display_bytes = (size_t)ctx->width * ctx->height * 4;
    return PT_OK;
}

// Frames in flight: launches the frames collected so far as one launch of the burst.
int launch_pending(PtContext* ctx, bool joining) {
    if (ctx->pending.frames == 0) return PT_OK;
    const PtContext::Pending p = ctx->pending;
    ctx->pending.frames = 0;
    // a single frame with nothing to overlap and nothing to follow (the caller flushes after every frame): the serial
    // launch does the same sums without a fold pass behind it
    if (joining && p.frames == 1u && ctx->burst == 0) return launch_now(ctx, p.begin, p.per_frame, 1u, p.per_frame, &p.prm, nullptr, false);
    const int rc = launch_now(ctx, p.begin, p.per_frame, p.frames, p.frames > 1u ? p.stride : p.per_frame, &p.prm, nullptr, true);
    if (rc != PT_NO_MEMORY) return rc;
    // no room for the planes of an overlapped launch: fold what is in flight, then one serial launch per frame, which
    // needs no buffer beyond the accumulation buffer and gives the same sums and the same pictures
    if (int rj = join_chains(ctx)) return rj;
    for (uint32_t f = 0; f < p.frames; ++f)
        if (int rs = launch_now(ctx, p.begin + f * (p.frames > 1u ? p.stride : 0u), p.per_frame, 1u, p.per_frame, &p.prm, nullptr, false)) return rs;
    return PT_OK;
}

// A serial launch (one frame in flight) with render-ahead around it (PtContext::ahead): for every call that is one launch
// of its own and has no event counters to keep.  The gain is largest where the serial launch is worst -- few samples on
// few pixels: 720p / 1 spp 0.21 -> 0.14 ms, 1440p / 1 spp +33 % -- and still there at 3840x2160: 1 / 4 / 16 samples per call
// +19 / +14 / +5 % (tools/ahead_4k.py).  It starts once two calls in a row have continued one another and stops with the
// first that does not.
int launch_serial(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* prm) {
    const uint32_t npix = ctx->width * ctx->height;
    const bool eligible = ctx->ahead_on && !ctx->own_burst && ctx->knob_trace.empty() && !ctx->knob_fail_chain_alloc &&
                          !(prm->flags & PT_FLAG_COUNTERS) &&
                          // calls that launch_path sends here as ONE launch: < 4 samples, or any count on frames of more than 2^22
                          // pixels (wants_parts); not beyond 2^24 pixels, where the four planes would pass 1 GiB
                          (sample_count < 4u || (uint64_t)npix > (1ull << 22)) && (uint64_t)npix <= (1ull << 24);
    PtContext::Ahead& ah = ctx->ahead;
    PtContext::Last& last = ctx->last;
    bool taken = false;
    if (ah.valid) {
        ah.valid = false;   // one way or the other, the plane is spoken for
        if (eligible && ah.begin == sample_begin && ah.count == sample_count && memcmp(&ah.prm, prm, sizeof(PtRenderParams)) == 0) {
            // the frame was launched ahead: behind its kernel, the pass that a frame in flight gets -- acc += plane (or the
            // running mean's step) and the frame's display -- on the context's stream
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->chain_done[ah.chain], 0));
            const uint32_t ci = ah.chain - AHEAD_CHAIN_0;
            HIP_TRY(pt_launch_fold(ctx->accum, ctx->chain_acc[ah.chain] + (size_t)ah.plane * npix, npix, 1u, prm->fused_rgba8 ? (uint32_t*)ctx->display : nullptr, nullptr, npix,
                                   prm->acc_mode, sample_begin, sample_count, ctx->stream));
            if (!ctx->ahead_fold_ev[ci][ah.plane]) HIP_TRY(hipEventCreateWithFlags(&ctx->ahead_fold_ev[ci][ah.plane], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ctx->ahead_fold_ev[ci][ah.plane], ctx->stream));
            ctx->ahead_fold_pending[ci][ah.plane] = true;
            ctx->path_launches++;
            ctx->ahead_used++;
            ctx->last_acc_mode = prm->acc_mode;
            if (prm->fused_rgba8) { ctx->display_bytes = (size_t)npix * 4; ctx->history_frames = 0; }
            taken = true;
        }
    }
    if (!taken)
        if (int rc = launch_now(ctx, sample_begin, sample_count, 1u, sample_count, prm, nullptr, false)) return rc;
    // does this call continue the one before it?
    const bool continues = last.valid && last.count == sample_count && memcmp(&last.prm, prm, sizeof(PtRenderParams)) == 0 &&
                           sample_begin >= last.begin + last.count && (last.streak == 0u || sample_begin - last.begin == last.stride);
    if (continues) { last.stride = sample_begin - last.begin; last.streak++; }
    else { last.streak = 0u; last.stride = 0u; last.count = sample_count; last.prm = *prm; }
    last.valid = true; last.begin = sample_begin;
    if (eligible && last.streak >= 1u && (uint64_t)sample_begin + last.stride + sample_count <= 0xFFFFFFFFull) {
        const uint32_t chain = AHEAD_CHAIN_0 + (ctx->ahead_next & 1u);
        const uint32_t plane = ctx->ahead_plane[chain - AHEAD_CHAIN_0];   // the plane this launch writes (it moves on to the other one)
        const int rc = launch_now(ctx, sample_begin + last.stride, sample_count, 1u, sample_count, prm, nullptr, false, (int)chain);
        if (rc == PT_OK) { ah.valid = true; ah.plane = plane; ah.begin = sample_begin + last.stride; ah.count = sample_count; ah.chain = chain; ah.prm = *prm; ctx->ahead_next++; }
        else if (rc != PT_NO_MEMORY) return rc;
        else ctx->ahead_on = false;   // no room for a plane: this context renders serially from here on
    }
    return PT_OK;
}

// One pt_render call (or one part of a large one).  With frames in flight the call is a FRAME of the next overlapped
// launch: consecutive calls with the same parameters and sample count, each continuing where the one before it ended
// (the reference's frame loop, main.cu:284-303, is exactly that), are collected and leave as one launch when
// frames_per_launch_max() of them have gathered or anything needs their result (join_chains).  One launch over many
// frames has one ramp and one tail instead of one per frame, hands out (frame, pixel) pairs instead of pixels -- at
// 1280x720 a resident lane gets 2.3 pixels per frame -- and needs one fold pass, which still adds the frames one by one.
int launch_path(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* prm, float4* into) {
    if (into) return launch_now(ctx, sample_begin, sample_count, 1u, sample_count, prm, into, false);  // (pt_render_noalloc has checked that no burst is open)
    if (!(ctx->flight > 1 && can_overlap(ctx, prm))) {
        if (int rc = join_chains(ctx)) return rc;
        return launch_serial(ctx, sample_begin, sample_count, prm);
    }
    // The frames of a launch are equally long and equally far apart: the second frame fixes the distance (the sample
    // count for one process; a multiple of it when the samples of every step are divided among GPUs, each rank
    // rendering the r-th slice of every step, path_tracer_amd/distributed.py).
    PtContext::Pending& p = ctx->pending;
    if (p.frames > 0) {
        bool fits = sample_count == p.per_frame && memcmp(prm, &p.prm, sizeof(PtRenderParams)) == 0;
        if (fits && p.frames == 1u) {
            fits = sample_begin >= p.begin + p.per_frame;   // (no overlap; a wrapped range is not collected either)
            if (fits) p.stride = sample_begin - p.begin;
        } else if (fits) {
            fits = sample_begin == p.begin + p.frames * p.stride;
        }
        if (!fits)
            if (int rc = launch_pending(ctx)) return rc;
    }
    if (p.frames == 0) { p.begin = sample_begin; p.per_frame = sample_count; p.stride = sample_count; p.prm = *prm; }
    p.frames++;
    ctx->last_acc_mode = prm->acc_mode;
    if (prm->fused_rgba8) ctx->display_bytes = (size_t)ctx->width * ctx->height * 4;
    if (p.frames >= frames_per_launch_max(ctx)) return launch_pending(ctx);
    return PT_OK;
}

}  // namespace

extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }
uint32_t pt_abi_version(void) { return 2u; }

int pt_create(int device_id, PtContext** out) {
    pt_clear_error();
    if (!out) return fail(PT_ERR_INVALID, "out_ctx is null");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(PT_ERR_INVALID, "device %d out of range (%d devices)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    PtContext* ctx = new (std::nothrow) PtContext();
    if (!ctx) return fail(PT_ERR_INVALID, "out of host memory");
    ctx->device = device_id;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->compute_units = (uint32_t)prop.multiProcessorCount;
    }
    if (const char* e = getenv("PT_AMD_HOSTPROF")) ctx->prof.on = atoi(e) != 0;
    if (const char* e = getenv("PT_AMD_FEW_MAX")) ctx->knob_few_max = atoi(e);
    if (const char* e = getenv("PT_AMD_MAX_BLOCKS")) ctx->knob_max_blocks = atoi(e);
    if (const char* e = getenv("PT_AMD_TRACE")) ctx->knob_trace = e;
    if (const char* e = getenv("PT_AMD_OWN_CHAIN0")) ctx->knob_own_chain0 = atoi(e) != 0;
    if (const char* e = getenv("PT_AMD_RENDER_AHEAD")) ctx->ahead_on = atoi(e) != 0;   // (pt_set_render_ahead's default, for A/B runs)
    if (const char* e = getenv("PT_AMD_CHUNK")) ctx->knob_chunk = atoi(e);
    if (const char* e = getenv("PT_AMD_REGEN_MIN")) ctx->knob_regen_min = atoi(e);
    if (const char* e = getenv("PT_AMD_IDLE_MIN")) ctx->knob_idle_min = atoi(e);
    if (const char* e = getenv("PT_AMD_PREFETCH")) ctx->knob_prefetch = atoi(e);
    if (const char* e = getenv("PT_AMD_GRID_PCT")) ctx->knob_grid_pct = atoi(e);
    if (const char* e = getenv("PT_AMD_FRAMES")) ctx->knob_frames = atoi(e);
    if (const char* e = getenv("PT_AMD_FAIL_CHAIN_ALLOC")) ctx->knob_fail_chain_alloc = atoi(e) != 0;
    if (const char* e = getenv("GPU_MAX_HW_QUEUES")) {  // HIP reads the same variable when it initialises
        const int q = atoi(e) - 4;
        ctx->burst_chains = q < 3 ? 3u : (q > (int)PT_MAX_CHAINS ? PT_MAX_CHAINS : (uint32_t)q);
    }
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(PT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    ctx->stream = ctx->own_stream;
    e = hipMalloc((void**)&ctx->counters, (12 + 64 * PT_HISTOGRAM_BINS) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemset(ctx->counters, 0, (12 + 64 * PT_HISTOGRAM_BINS) * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->sink, 256);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->work_counter, 2 * PT_MAX_CHAINS * WORK_SET_WORDS * 4);
    if (e == hipSuccess) e = hipMemset(ctx->work_counter, 0, 2 * PT_MAX_CHAINS * WORK_SET_WORDS * 4);
    if (e != hipSuccess) { pt_destroy(ctx); return fail(PT_ERR_HIP, "context allocation failed: %s", hipGetErrorString(e)); }
    // set-up for pt_render_noalloc, which may not allocate or query: the address of this GPU's copy of the static
    // scratch array (resolving it also loads the code object on the device) and the occupancy of every kernel build
    e = pt_scratch_setup(device_id, &ctx->scratch);
    if (e != hipSuccess) { pt_destroy(ctx); return fail(PT_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device_id, hipGetErrorString(e)); }
    for (int v = 0; v < 8; ++v)
        if (pt_path_resident_blocks((v & 4) != 0, (v & 2) != 0, (v & 1) != 0, device_id, &ctx->resident_blocks[v]) != hipSuccess) ctx->resident_blocks[v] = 0;
    *out = ctx;
    return PT_OK;
}

int pt_destroy(PtContext* ctx) {
    pt_clear_error();
    if (!ctx) return PT_OK;
    if (ctx->prof.on) {
        static const char* const names[5] = {"wait fork event", "launch path kernel", "wait previous fold", "launch fold pass", "record fold event"};
        for (int k = 0; k < 5; ++k)
            if (ctx->prof.n[k]) fprintf(stderr, "[pt hostprof] %-20s %8llu calls  %7.2f us each\n", names[k], (unsigned long long)ctx->prof.n[k], ctx->prof.t[k] / (double)ctx->prof.n[k]);
    }
    (void)hipSetDevice(ctx->device);
    for (uint32_t c = 0; c < PT_MAX_CHAINS; ++c) {
        if (c > 0 && ctx->chain_stream[c]) { (void)hipStreamSynchronize(ctx->chain_stream[c]); (void)hipStreamDestroy(ctx->chain_stream[c]); }
        if (ctx->chain_done[c]) (void)hipEventDestroy(ctx->chain_done[c]);
        if (ctx->fold_done[c]) (void)hipEventDestroy(ctx->fold_done[c]);
        if (ctx->chain_acc[c]) (void)hipFree(ctx->chain_acc[c]);
        if (c == 0 && ctx->history) (void)hipFree(ctx->history);
    }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    for (auto& per_chain : ctx->ahead_fold_ev) for (hipEvent_t& e : per_chain) if (e) (void)hipEventDestroy(e);
    if (ctx->trace) (void)hipFree(ctx->trace);
    if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); }
    if (ctx->scene_arena) (void)hipFree(ctx->scene_arena);
    if (ctx->accum_own) (void)hipFree(ctx->accum_own);
    if (ctx->display) (void)hipFree(ctx->display);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->ids) (void)hipFree(ctx->ids);
    if (ctx->sink) (void)hipFree(ctx->sink);
    if (ctx->work_counter) (void)hipFree(ctx->work_counter);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (g_current == ctx) g_current = nullptr;
    delete ctx;
    return PT_OK;
}

int pt_set_stream(PtContext* ctx, void* hip_stream) {
    if (int rc = use(ctx)) return rc;
    ctx->stream_chosen = true;
    return switch_stream(ctx, (hipStream_t)hip_stream);
}

int pt_use_own_stream(PtContext* ctx) {
    if (int rc = use(ctx)) return rc;
    ctx->stream_chosen = true;
    return switch_stream(ctx, ctx->own_stream);
}

int pt_upload_scene(PtContext* ctx, const PtSceneSoA* s) {
    if (int rc = use(ctx)) return rc;
    if (int rc = join_chains(ctx)) return rc;  // the stream is synchronised below, before the old scene is freed
    if (int rc = drop_ahead(ctx)) return rc;
    if (!s) return fail(PT_ERR_INVALID, "scene is null");
    const uint32_t ns = s->num_spheres, np = s->num_planes, nt = s->num_triangles, nm = s->num_materials;
    if (ns > PT_MAX_SPHERES || np > PT_MAX_PLANES || nt > PT_MAX_TRIANGLES || nm > PT_MAX_MATERIALS)
        return fail(PT_ERR_CAPACITY, "scene exceeds capacity (%u spheres / %u planes / %u triangles / %u materials)", ns, np, nt, nm);
    if (nm == 0 || ns + np + nt == 0) return fail(PT_ERR_INVALID, "scene needs at least one primitive and one material");
#define NEED(cnt, ptr) if ((cnt) && !(s->ptr)) return fail(PT_ERR_INVALID, "scene array " #ptr " is null")
    NEED(ns, sph_cx); NEED(ns, sph_cy); NEED(ns, sph_cz); NEED(ns, sph_r); NEED(ns, sph_mat); NEED(ns, sph_obj);
    NEED(ns, sph_emissive); NEED(ns, sph_em_r); NEED(ns, sph_em_g); NEED(ns, sph_em_b);
    NEED(np, pl_px); NEED(np, pl_py); NEED(np, pl_pz); NEED(np, pl_nx); NEED(np, pl_ny); NEED(np, pl_nz); NEED(np, pl_mat); NEED(np, pl_obj);
    NEED(nt, tri_v0x); NEED(nt, tri_v0y); NEED(nt, tri_v0z); NEED(nt, tri_v1x); NEED(nt, tri_v1y); NEED(nt, tri_v1z);
    NEED(nt, tri_v2x); NEED(nt, tri_v2y); NEED(nt, tri_v2z); NEED(nt, tri_mat); NEED(nt, tri_obj);
    NEED(nm, mat_type); NEED(nm, mat_albedo_r); NEED(nm, mat_albedo_g); NEED(nm, mat_albedo_b); NEED(nm, mat_roughness);
    NEED(nm, mat_metallic); NEED(nm, mat_ior); NEED(nm, mat_em_r); NEED(nm, mat_em_g); NEED(nm, mat_em_b);
#undef NEED
    for (uint32_t i = 0; i < nm; ++i)
        if (s->mat_type[i] > PT_MAT_EMISSIVE) return fail(PT_ERR_INVALID, "material %u has unknown type %u", i, s->mat_type[i]);
    for (uint32_t i = 0; i < ns; ++i) {
        if (s->sph_mat[i] >= nm) return fail(PT_ERR_INVALID, "sphere %u: material index %u out of range", i, s->sph_mat[i]);
        if (!(s->sph_r[i] > 0)) return fail(PT_ERR_INVALID, "sphere %u: radius must be positive", i);
    }
    for (uint32_t i = 0; i < np; ++i)
        if (s->pl_mat[i] >= nm) return fail(PT_ERR_INVALID, "plane %u: material index %u out of range", i, s->pl_mat[i]);
    for (uint32_t i = 0; i < nt; ++i)
        if (s->tri_mat[i] >= nm) return fail(PT_ERR_INVALID, "triangle %u: material index %u out of range", i, s->tri_mat[i]);

    // ---- light list: spheres whose material is Emissive, in sphere order
    std::vector<int32_t> sphere_light(ns, -1);
    std::vector<uint32_t> light_sphere;
    for (uint32_t i = 0; i < ns; ++i)
        if (s->mat_type[s->sph_mat[i]] == PT_MAT_EMISSIVE) { sphere_light[i] = (int32_t)light_sphere.size(); light_sphere.push_back(i); }
    const uint32_t nl = (uint32_t)light_sphere.size();

    // ---- host staging, one arena: [float4 arrays][u32 arrays][double arrays]
    std::vector<float4> f4;
    auto push4 = [&](float x, float y, float z, float w) { f4.push_back(make_float4(x, y, z, w)); };
    // loop operands, padded to whole batches with primitives no ray can hit
    const size_t o_sph = f4.size();
    for (uint32_t i = 0; i < ns; ++i) { float r = (float)s->sph_r[i]; push4((float)s->sph_cx[i], (float)s->sph_cy[i], (float)s->sph_cz[i], r * r); }
    for (uint32_t i = ns; i % 4 != 0 || i == 0; ++i) push4(0, 0, 0, -1.0f);  // r*r < 0: discriminant < 0 for every ray
    // planes: the unit normal is +-e_k for most (every wall of the Cornell boxes); those go to
    // per-axis lists with the cheap test, the others to the general list.  Ids stay insertion order.
    std::vector<F3> pl_normal(np);
    std::vector<std::pair<float, uint32_t>> axis_list[3];
    std::vector<uint32_t> general;
    for (uint32_t i = 0; i < np; ++i) {  // Plane_t's constructor normalises in float (ImplicitObject.cuh:165)
        F3 n = fnorm(F3{(float)s->pl_nx[i], (float)s->pl_ny[i], (float)s->pl_nz[i]});
        pl_normal[i] = n;
        const float p[3] = {(float)s->pl_px[i], (float)s->pl_py[i], (float)s->pl_pz[i]};
        const float nn[3] = {n.x, n.y, n.z};
        const int zeros = (n.x == 0.0f) + (n.y == 0.0f) + (n.z == 0.0f);
        if (zeros == 2) { const int k = n.x != 0.0f ? 0 : (n.y != 0.0f ? 1 : 2); (void)nn; axis_list[k].push_back({p[k], ns + i}); }
        else if (zeros < 2) general.push_back(i);
        // zeros == 3: a degenerate normal can never be hit (|denom| < 1e-6) and is dropped
    }
    const size_t o_pl = f4.size();
    for (uint32_t i : general) {
        const F3 n = pl_normal[i];
        const float px = (float)s->pl_px[i], py = (float)s->pl_py[i], pz = (float)s->pl_pz[i];
        push4(n.x, n.y, n.z, -(px * n.x + py * n.y + pz * n.z));
        push4(u2f(ns + i), 0, 0, 0);
    }
    for (size_t i = general.size(); i % 4 != 0 || i == 0; ++i) { push4(NAN, NAN, NAN, 0); push4(0, 0, 0, 0); }  // never hit
    const size_t o_axp = f4.size();
    size_t ax_off[3];
    for (int k = 0; k < 3; ++k) {
        ax_off[k] = f4.size() - o_axp;
        const auto& L = axis_list[k];
        for (size_t i = 0; i < L.size() || i == 0; i += 2) {
            const bool a = i < L.size(), b = i + 1 < L.size();
            push4(a ? L[i].first : NAN, u2f(a ? L[i].second : 0u), b ? L[i + 1].first : NAN, u2f(b ? L[i + 1].second : 0u));
        }
    }
    const size_t o_pla = f4.size();
    for (uint32_t i = 0; i < np; ++i) {
        const F3 n = pl_normal[i];
        const float px = (float)s->pl_px[i], py = (float)s->pl_py[i], pz = (float)s->pl_pz[i];
        const bool degenerate = n.x == 0.0f && n.y == 0.0f && n.z == 0.0f;
        if (degenerate) push4(NAN, NAN, NAN, 0); else push4(n.x, n.y, n.z, -(px * n.x + py * n.y + pz * n.z));
    }
    if (np == 0) push4(NAN, NAN, NAN, 0);
    const size_t o_tri = f4.size();
    std::vector<F3> tri_normal(nt);
    for (uint32_t i = 0; i < nt; ++i) {
        // vertices as the reference holds them (float), the affine map in double
        const double v0[3] = {(double)(float)s->tri_v0x[i], (double)(float)s->tri_v0y[i], (double)(float)s->tri_v0z[i]};
        const double v1[3] = {(double)(float)s->tri_v1x[i], (double)(float)s->tri_v1y[i], (double)(float)s->tri_v1z[i]};
        const double v2[3] = {(double)(float)s->tri_v2x[i], (double)(float)s->tri_v2y[i], (double)(float)s->tri_v2z[i]};
        D3h e1{v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]}, e2{v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
        D3h n = dcross(e1, e2);
        const double det = n.x * n.x + n.y * n.y + n.z * n.z;  // e1 . (e2 x n) = |n|^2
        D3h r0{0, 0, 0}, r1{0, 0, 0}, r2{0, 0, 0};
        if (det > 0) { r0 = dmul(dcross(e2, n), 1 / det); r1 = dmul(dcross(n, e1), 1 / det); r2 = dmul(n, 1 / det); }
        auto row = [&](D3h r) { push4((float)r.x, (float)r.y, (float)r.z, (float)-(r.x * v0[0] + r.y * v0[1] + r.z * v0[2])); };
        row(r0); row(r1); row(r2);
        F3 fe1{(float)e1.x, (float)e1.y, (float)e1.z}, fe2{(float)e2.x, (float)e2.y, (float)e2.z};
        tri_normal[i] = fnorm(fcross(fe1, fe2));  // Triangle_t::computeNormal, Polygon.cuh:53-57
    }
    for (uint32_t i = nt; i % 2 != 0 || i == 0; ++i) for (int k = 0; k < 3; ++k) push4(0, 0, 0, 0);  // zero map: t is NaN
    // winner records: one 64-B line per primitive
    const size_t o_rec = f4.size();
    auto push_material = [&](uint32_t m, int32_t light) {
        float rough = (float)s->mat_roughness[m];
        if (s->mat_type[m] == PT_MAT_METAL) rough = rough < 1.0f ? rough : 1.0f;  // Metal_t ctor, Material.cuh:94
        push4((float)s->mat_albedo_r[m], (float)s->mat_albedo_g[m], (float)s->mat_albedo_b[m], u2f(s->mat_type[m]));
        push4(rough, (float)s->mat_metallic[m], (float)s->mat_ior[m], u2f((uint32_t)light));
        push4((float)s->mat_em_r[m], (float)s->mat_em_g[m], (float)s->mat_em_b[m], 0.f);
    };
    for (uint32_t i = 0; i < ns; ++i) {
        push4((float)s->sph_cx[i], (float)s->sph_cy[i], (float)s->sph_cz[i], 1.0f / (float)s->sph_r[i]);
        push_material(s->sph_mat[i], sphere_light[i]);
    }
    for (uint32_t i = 0; i < np; ++i) { push4(pl_normal[i].x, pl_normal[i].y, pl_normal[i].z, 0.f); push_material(s->pl_mat[i], -1); }
    for (uint32_t i = 0; i < nt; ++i) { push4(tri_normal[i].x, tri_normal[i].y, tri_normal[i].z, 0.f); push_material(s->tri_mat[i], -1); }
    const size_t o_lt = f4.size();
    for (uint32_t j = 0; j < nl; ++j) {
        const uint32_t i = light_sphere[j], m = s->sph_mat[i];
        f4.push_back(f4[o_sph + i]);
        push4((float)s->mat_em_r[m], (float)s->mat_em_g[m], (float)s->mat_em_b[m], u2f(i));
        if (s->sph_emissive[i]) push4((float)s->sph_em_r[i], (float)s->sph_em_g[i], (float)s->sph_em_b[i], 0.f);
        else push4((float)s->mat_em_r[m], (float)s->mat_em_g[m], (float)s->mat_em_b[m], 0.f);
    }
    if (nl == 0) for (int k = 0; k < 3; ++k) push4(0, 0, 0, 0);

    std::vector<uint32_t> u32;
    const size_t o_si = u32.size();
    for (uint32_t i = 0; i < ns; ++i) { u32.push_back(s->sph_mat[i]); u32.push_back(s->sph_obj[i]); u32.push_back(s->sph_emissive[i] ? 1u : 0u); }
    const size_t o_pi = u32.size();
    for (uint32_t i = 0; i < np; ++i) { u32.push_back(s->pl_mat[i]); u32.push_back(s->pl_obj[i]); }
    const size_t o_ti = u32.size();
    for (uint32_t i = 0; i < nt; ++i) { u32.push_back(s->tri_mat[i]); u32.push_back(s->tri_obj[i]); }
    while (u32.size() % 4) u32.push_back(0);

    std::vector<double> f8;
    const size_t o_dsph = f8.size();
    for (uint32_t i = 0; i < ns; ++i) { f8.push_back(s->sph_cx[i]); f8.push_back(s->sph_cy[i]); f8.push_back(s->sph_cz[i]); f8.push_back(s->sph_r[i]); }
    const size_t o_dpl = f8.size();
    for (uint32_t i = 0; i < np; ++i) {  // Plane's constructor normalises in double (ImplicitObject.h:143)
        D3h n = dnorm(D3h{s->pl_nx[i], s->pl_ny[i], s->pl_nz[i]});
        f8.push_back(s->pl_px[i]); f8.push_back(s->pl_py[i]); f8.push_back(s->pl_pz[i]);
        f8.push_back(n.x); f8.push_back(n.y); f8.push_back(n.z);
    }
    const size_t o_dtri = f8.size();
    for (uint32_t i = 0; i < nt; ++i) {
        f8.push_back(s->tri_v0x[i]); f8.push_back(s->tri_v0y[i]); f8.push_back(s->tri_v0z[i]);
        f8.push_back(s->tri_v1x[i]); f8.push_back(s->tri_v1y[i]); f8.push_back(s->tri_v1z[i]);
        f8.push_back(s->tri_v2x[i]); f8.push_back(s->tri_v2y[i]); f8.push_back(s->tri_v2z[i]);
    }
    const size_t o_dem = f8.size();
    for (uint32_t i = 0; i < ns; ++i) { f8.push_back(s->sph_em_r[i]); f8.push_back(s->sph_em_g[i]); f8.push_back(s->sph_em_b[i]); }
    const size_t o_dal = f8.size();
    for (uint32_t i = 0; i < nm; ++i) { f8.push_back(s->mat_albedo_r[i]); f8.push_back(s->mat_albedo_g[i]); f8.push_back(s->mat_albedo_b[i]); }
    f8.push_back(0.0);

    const size_t b_f4 = f4.size() * sizeof(float4), b_u32 = u32.size() * 4, b_f8 = f8.size() * 8;
    const size_t total = b_f4 + b_u32 + b_f8;
    std::vector<unsigned char> host(total);
    memcpy(host.data(), f4.data(), b_f4);
    memcpy(host.data() + b_f4, u32.data(), b_u32);
    memcpy(host.data() + b_f4 + b_u32, f8.data(), b_f8);

    HIP_TRY(hipStreamSynchronize(ctx->stream));  // no launch may still read the old scene
    if (ctx->scene_arena) { HIP_TRY(hipFree(ctx->scene_arena)); ctx->scene_arena = nullptr; ctx->have_scene = false; }
    HIP_TRY(hipMalloc(&ctx->scene_arena, total));
    HIP_TRY(hipMemcpy(ctx->scene_arena, host.data(), total, hipMemcpyHostToDevice));

    const float4* F = (const float4*)ctx->scene_arena;
    const uint32_t* U = (const uint32_t*)((const unsigned char*)ctx->scene_arena + b_f4);
    const double* D = (const double*)((const unsigned char*)ctx->scene_arena + b_f4 + b_u32);
    DevSceneF& sf = ctx->sf;
    sf.sph = F + o_sph; sf.pl = F + o_pl; sf.axp = F + o_axp; sf.pla = F + o_pla; sf.tri = F + o_tri; sf.rec = F + o_rec; sf.lights = F + o_lt;
    sf.ns = (int)ns; sf.np = (int)np; sf.nt = (int)nt; sf.nl = (int)nl;
    sf.npg = (int)general.size();
    sf.nax = (int)axis_list[0].size(); sf.nay = (int)axis_list[1].size(); sf.naz = (int)axis_list[2].size();
    sf.ax_off_y = (int)ax_off[1]; sf.ax_off_z = (int)ax_off[2];
    DevSceneD& sd = ctx->sd;
    sd.sph = D + o_dsph; sd.pl = D + o_dpl; sd.tri = D + o_dtri; sd.sph_em = D + o_dem; sd.mat_albedo = D + o_dal;
    sd.sph_info = U + o_si; sd.pl_info = U + o_pi; sd.tri_info = U + o_ti;
    sd.ns = (int)ns; sd.np = (int)np; sd.nt = (int)nt;
    ctx->have_scene = true;
    return PT_OK;
}

int pt_resize(PtContext* ctx, uint32_t width, uint32_t height) {
    if (int rc = use(ctx)) return rc;
    return alloc_frame(ctx, width, height);
}

int pt_set_camera(PtContext* ctx, const PtCamera* c) {
    if (int rc = use(ctx)) return rc;
    if (!c) return fail(PT_ERR_INVALID, "camera is null");
    if (c->model > PT_CAMERA_CPU) return fail(PT_ERR_INVALID, "unknown camera model %u", c->model);
    if (c->width == 0 || c->height == 0) return fail(PT_ERR_INVALID, "camera has zero width or height");
    if (c->model == PT_CAMERA_CPU && (c->width < 2 || c->height < 2))
        return fail(PT_ERR_INVALID, "the CPU camera divides by (width-1) and (height-1)");
    if (!(c->fov_deg > 0.0 && c->fov_deg < 180.0)) return fail(PT_ERR_INVALID, "fov must be in (0,180) degrees");
    if (c->use_dof && !(c->aperture >= 0.0 && c->focus_dist > 0.0)) return fail(PT_ERR_INVALID, "depth of field needs aperture >= 0 and focus_dist > 0");
    if (int rc = launch_pending(ctx)) return rc;  // frames held back for the next overlapped launch belong to the old camera
    if (int rc = drop_ahead(ctx)) return rc;      // ... and so does a frame rendered ahead
    if (int rc = alloc_frame(ctx, c->width, c->height)) return rc;

    // ---- Camera_t::initialize (Camera.cu:3-23), float; initializeGPUData (raytracer_kernel.cu:308-321)
    F3 o{(float)c->origin[0], (float)c->origin[1], (float)c->origin[2]};
    F3 tg{(float)c->target[0], (float)c->target[1], (float)c->target[2]};
    F3 up{(float)c->up[0], (float)c->up[1], (float)c->up[2]};
    if (c->normalize_up) up = fnorm(up);
    F3 w = fnorm(F3{o.x - tg.x, o.y - tg.y, o.z - tg.z});
    F3 u = fnorm(fcross(up, w));
    F3 v = fcross(w, u);
    const float fov = (float)((double)(float)c->fov_deg * M_PI / 180.0f);  // :315
    const float aspect = (float)c->width / (float)(int)c->height;           // :247
    const float vh = 2.0f * tanf(fov * 0.5f);                               // :248
    const float vw = aspect * vh;
    DevCamF& cf = ctx->cf;
    cf.ox = o.x; cf.oy = o.y; cf.oz = o.z;
    cf.fx = -w.x; cf.fy = -w.y; cf.fz = -w.z;
    cf.rx = u.x; cf.ry = u.y; cf.rz = u.z;
    cf.ux = v.x; cf.uy = v.y; cf.uz = v.z;
    cf.half_w = vw * 0.5f; cf.half_h = vh * 0.5f;
    cf.sx = 2.0f * cf.half_w / (float)c->width; cf.sy = 2.0f * cf.half_h / (float)(int)c->height;
    cf.width = c->width; cf.height = c->height;
    cf.lens_radius = c->use_dof ? (float)c->aperture / 2.0f : 0.0f;  // Camera.cu:20
    cf.focus_dist = (float)c->focus_dist;

    DevCamD& cd = ctx->cd;
    memset(&cd, 0, sizeof cd);
    cd.model = c->model; cd.width = c->width; cd.height = c->height;
    if (c->model == PT_CAMERA_CPU) {
        // ---- Camera::initialize, cpu_pathtracer/include/Camera.h:65-85, double
        D3h od{c->origin[0], c->origin[1], c->origin[2]}, td{c->target[0], c->target[1], c->target[2]};
        D3h ud{c->up[0], c->up[1], c->up[2]};
        if (c->normalize_up) ud = dnorm(ud);
        D3h wd = dnorm(dsub(od, td));
        D3h uu = dnorm(dcross(ud, wd));
        D3h vv = dcross(wd, uu);
        const double theta = c->fov_deg * M_PI / 180.0;
        const double vhd = 2.0 * std::tan(theta / 2.0);
        const double vwd = vhd * c->aspect_ratio;
        D3h hor = dmul(dmul(uu, vwd), c->focus_dist), ver = dmul(dmul(vv, vhd), c->focus_dist);
        D3h tl = dsub(dadd(dsub(od, dmul(hor, 1 / 2.0)), dmul(ver, 1 / 2.0)), dmul(wd, c->focus_dist));
        cd.origin[0] = od.x; cd.origin[1] = od.y; cd.origin[2] = od.z;
        cd.top_left[0] = tl.x; cd.top_left[1] = tl.y; cd.top_left[2] = tl.z;
        cd.horizontal[0] = hor.x; cd.horizontal[1] = hor.y; cd.horizontal[2] = hor.z;
        cd.vertical[0] = ver.x; cd.vertical[1] = ver.y; cd.vertical[2] = ver.z;
    } else {
        cd.origin[0] = o.x; cd.origin[1] = o.y; cd.origin[2] = o.z;
        cd.fwd[0] = -w.x; cd.fwd[1] = -w.y; cd.fwd[2] = -w.z;
        cd.right[0] = u.x; cd.right[1] = u.y; cd.right[2] = u.z;
        cd.up[0] = v.x; cd.up[1] = v.y; cd.up[2] = v.z;
        const double aspd = (double)(float)c->width / (double)(int)c->height;
        const double vhd = 2.0 * std::tan((double)fov * 0.5);
        const double vwd = aspd * vhd;
        cd.half_w = vwd * 0.5; cd.half_h = vhd * 0.5;
    }
    ctx->have_camera = true;
    return PT_OK;
}

static int bind_accum(PtContext* ctx, void* dev_float4) {
    if (ctx->width == 0) return fail(PT_ERR_STATE, "set the frame size (pt_set_camera / pt_resize) before binding a buffer");
    if (int rc = join_chains(ctx)) return rc;  // samples in flight belong to the buffer bound so far
    if (dev_float4) {
        if (((uintptr_t)dev_float4 & 15u) != 0) return fail(PT_ERR_INVALID, "accumulation buffer must be 16-byte aligned");
        ctx->accum = (float4*)dev_float4; ctx->accum_bound = true;
    } else {
        ctx->accum = ctx->accum_own; ctx->accum_bound = false;
    }
    return PT_OK;
}

int pt_bind_accum(PtContext* ctx, void* dev_float4) {
    if (int rc = use(ctx)) return rc;
    return bind_accum(ctx, dev_float4);
}

int pt_reset(PtContext* ctx) {
    if (int rc = use(ctx)) return rc;
    if (!ctx->accum) return fail(PT_ERR_STATE, "no frame allocated");
    if (int rc = join_chains(ctx)) return rc;  // leaves the other chains' accumulators zero
    HIP_TRY(hipMemsetAsync(ctx->accum, 0, (size_t)ctx->width * ctx->height * sizeof(float4), ctx->stream));
    ctx->history_frames = 0;
    return PT_OK;
}

static int render(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* prm) {
    if (!prm) return fail(PT_ERR_INVALID, "params is null");
    if (!ctx->have_scene) return fail(PT_ERR_STATE, "no scene uploaded");
    if (!ctx->have_camera) return fail(PT_ERR_STATE, "no camera set");
    if (ctx->cf.width != ctx->width || ctx->cf.height != ctx->height)
        return fail(PT_ERR_STATE, "frame was resized to %ux%u after the camera was set for %ux%u", ctx->width, ctx->height, ctx->cf.width, ctx->cf.height);
    if (sample_count == 0) return PT_OK;
    if ((uint64_t)sample_begin + sample_count > 0xFFFFFFFFull) return fail(PT_ERR_INVALID, "sample range overflows 32 bits");
    if (prm->acc_mode > PT_ACC_RUNNING_MEAN) return fail(PT_ERR_INVALID, "unknown acc_mode %u", prm->acc_mode);
    if (prm->integrator == PT_INTEGRATOR_DIRECT) {
        if (prm->acc_mode != PT_ACC_SUM) return fail(PT_ERR_INVALID, "the direct integrator accumulates sums only");
        if (int rc = join_chains(ctx)) return rc;
        HIP_TRY(pt_launch_direct(ctx->sd, ctx->cd, ctx->accum, sample_begin, sample_count, prm->seed,
                                 (prm->flags & PT_FLAG_COUNTERS) ? ctx->counters : nullptr, ctx->stream));
        ctx->last_acc_mode = PT_ACC_SUM;
        return PT_OK;
    }
    if (prm->integrator != PT_INTEGRATOR_PATH) return fail(PT_ERR_INVALID, "unknown integrator %u", prm->integrator);
    if (ctx->cd.model != PT_CAMERA_CUDA) return fail(PT_ERR_INVALID, "the path integrator uses the CUDA camera model");
    if (prm->max_depth == 0) return fail(PT_ERR_INVALID, "max_depth must be at least 1");
    if (prm->acc_mode == PT_ACC_RUNNING_MEAN && sample_count != 1)
        return fail(PT_ERR_INVALID, "the running-mean mode adds one sample per launch, like renderKernel");
    if (wants_parts(ctx, sample_count, prm)) return render_in_parts(ctx, sample_begin, sample_count, prm);
    return launch_path(ctx, sample_begin, sample_count, prm);
}

int pt_render(PtContext* ctx, uint32_t sample_begin, uint32_t sample_count, const PtRenderParams* prm) {
    if (int rc = use(ctx)) return rc;
    return render(ctx, sample_begin, sample_count, prm);
}

int pt_render_noalloc(PtContext* ctx, void* accum_float4, uint32_t width, uint32_t height, uint32_t sample_begin,
                      uint32_t sample_count, const PtRenderParams* prm) {
    // (not use(): that would first fold frames a launchRenderKernel caller left in flight -- launches on other streams,
    //  possibly the chains' first allocation.  Frames in flight are refused below, whoever left them.)
    pt_clear_error();
    if (!ctx) return fail(PT_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->burst != 0 || ctx->pending.frames != 0) return fail(PT_ERR_STATE, "frames are in flight: call pt_flush first (folding them launches on other streams)");
    if (int rc = end_shim_binding(ctx)) return rc;   // nothing is in flight: this only gives the context its own buffer back
    if (!prm) return fail(PT_ERR_INVALID, "params is null");
    if (!accum_float4 || ((uintptr_t)accum_float4 & 15u) != 0) return fail(PT_ERR_INVALID, "the accumulation buffer must be a 16-byte aligned device pointer");
    if (!ctx->have_scene) return fail(PT_ERR_STATE, "no scene uploaded");
    if (!ctx->have_camera) return fail(PT_ERR_STATE, "no camera set");
    if ((uint64_t)width * height > PT_NOALLOC_MAX_PIXELS)
        return fail(PT_ERR_CAPACITY, "pt_render_noalloc: %ux%u exceeds the maximum of %u pixels per call", width, height, PT_NOALLOC_MAX_PIXELS);
    if (sample_count > PT_NOALLOC_MAX_SAMPLES)
        return fail(PT_ERR_CAPACITY, "pt_render_noalloc: %u samples exceed the maximum of %u per call", sample_count, PT_NOALLOC_MAX_SAMPLES);
    if (width != ctx->cf.width || height != ctx->cf.height)
        return fail(PT_ERR_INVALID, "pt_render_noalloc: %ux%u does not match the camera's %ux%u", width, height, ctx->cf.width, ctx->cf.height);
    if (ctx->cf.width != ctx->width || ctx->cf.height != ctx->height)
        return fail(PT_ERR_STATE, "frame was resized to %ux%u after the camera was set for %ux%u", ctx->width, ctx->height, ctx->cf.width, ctx->cf.height);
    if (sample_count == 0) return PT_OK;
    if ((uint64_t)sample_begin + sample_count > 0xFFFFFFFFull) return fail(PT_ERR_INVALID, "sample range overflows 32 bits");
    if (prm->integrator != PT_INTEGRATOR_PATH) return fail(PT_ERR_INVALID, "pt_render_noalloc runs the path integrator");
    if (ctx->cd.model != PT_CAMERA_CUDA) return fail(PT_ERR_INVALID, "the path integrator uses the CUDA camera model");
    if (prm->acc_mode > PT_ACC_RUNNING_MEAN) return fail(PT_ERR_INVALID, "unknown acc_mode %u", prm->acc_mode);
    if (prm->max_depth == 0) return fail(PT_ERR_INVALID, "max_depth must be at least 1");
    if (prm->acc_mode == PT_ACC_RUNNING_MEAN && sample_count != 1)
        return fail(PT_ERR_INVALID, "the running-mean mode adds one sample per launch, like renderKernel");
    if (!ctx->scratch) return fail(PT_ERR_STATE, "the scratch array was not resolved at pt_create");
    return launch_path(ctx, sample_begin, sample_count, prm, (float4*)accum_float4);
}

int pt_resolve(PtContext* ctx, uint32_t mode) {
    if (int rc = use(ctx)) return rc;
    if (!ctx->accum) return fail(PT_ERR_STATE, "no frame allocated");
    if (mode > PT_RESOLVE_RGB8_PPM) return fail(PT_ERR_INVALID, "unknown resolve mode %u", mode);
    if (int rc = join_chains(ctx)) return rc;
    const uint32_t npix = ctx->width * ctx->height;
    HIP_TRY(pt_launch_resolve(ctx->accum, ctx->display, npix, ctx->last_acc_mode, mode, ctx->stream));
    ctx->display_bytes = (size_t)npix * (mode == PT_RESOLVE_FLOAT_MEAN ? 16 : mode == PT_RESOLVE_RGB8_PPM ? 3 : 4);
    ctx->history_frames = 0;  // the display no longer continues the history planes
    return PT_OK;
}

int pt_resolve_range(PtContext* ctx, uint32_t mode, uint32_t first_pixel, uint32_t pixel_count, void* dev_out) {
    if (int rc = use(ctx)) return rc;
    if (!ctx->accum) return fail(PT_ERR_STATE, "no frame allocated");
    if (mode > PT_RESOLVE_RGB8_PPM) return fail(PT_ERR_INVALID, "unknown resolve mode %u", mode);
    if (!dev_out) return fail(PT_ERR_INVALID, "dev_out is null");
    const uint64_t npix = (uint64_t)ctx->width * ctx->height;
    if ((uint64_t)first_pixel + pixel_count > npix) return fail(PT_ERR_INVALID, "pixels [%u, %u + %u) exceed the frame's %llu", first_pixel, first_pixel, pixel_count, (unsigned long long)npix);
    if (mode == PT_RESOLVE_FLOAT_MEAN && ((uintptr_t)dev_out & 15u) != 0) return fail(PT_ERR_INVALID, "a float4 output must be 16-byte aligned");
    if (mode != PT_RESOLVE_FLOAT_MEAN && mode != PT_RESOLVE_RGB8_PPM && ((uintptr_t)dev_out & 3u) != 0) return fail(PT_ERR_INVALID, "an RGBA8 output must be 4-byte aligned");
    if (int rc = join_chains(ctx)) return rc;
    if (pixel_count == 0) return PT_OK;
    HIP_TRY(pt_launch_resolve(ctx->accum + first_pixel, dev_out, pixel_count, ctx->last_acc_mode, mode, ctx->stream));
    return PT_OK;
}

int pt_readback(PtContext* ctx, void* host_dst, size_t bytes) {
    if (int rc = use(ctx)) return rc;
    if (!host_dst) return fail(PT_ERR_INVALID, "host_dst is null");
    if (ctx->display_bytes == 0) return fail(PT_ERR_STATE, "nothing resolved yet");
    if (bytes > ctx->display_bytes) return fail(PT_ERR_INVALID, "asked for %zu bytes, display buffer holds %zu", bytes, ctx->display_bytes);
    if (int rc = join_chains(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(host_dst, ctx->display, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

uint32_t pt_history_frames(PtContext* ctx) {
    if (use(ctx) != PT_OK || join_chains(ctx) != PT_OK) return 0u;   // (a failure leaves its message in pt_last_error())
    return ctx->history_frames;
}

int pt_readback_history(PtContext* ctx, uint32_t frames_back, void* host_dst, size_t bytes) {
    if (int rc = use(ctx)) return rc;
    if (!host_dst) return fail(PT_ERR_INVALID, "host_dst is null");
    if (int rc = join_chains(ctx)) return rc;
    if (frames_back < 1u || frames_back > ctx->history_frames)
        return fail(PT_ERR_STATE, "frame %u back is not held: the last fold pass displayed %u frames before its newest", frames_back, ctx->history_frames);
    const size_t plane = (size_t)ctx->width * ctx->height * 4;
    if (bytes > plane) return fail(PT_ERR_INVALID, "asked for %zu bytes, a picture holds %zu", bytes, plane);
    // plane f of the history is frame f of the last launch folded; its newest frame (frames - 1) went to the display buffer
    const uint8_t* src = (const uint8_t*)ctx->history + (size_t)(ctx->history_frames - frames_back) * plane;
    HIP_TRY(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_readback_accum(PtContext* ctx, float* host_dst, size_t bytes) {
    if (int rc = use(ctx)) return rc;
    if (!host_dst) return fail(PT_ERR_INVALID, "host_dst is null");
    if (!ctx->accum) return fail(PT_ERR_STATE, "no frame allocated");
    const size_t have = (size_t)ctx->width * ctx->height * sizeof(float4);
    if (bytes > have) return fail(PT_ERR_INVALID, "asked for %zu bytes, accumulation buffer holds %zu", bytes, have);
    if (int rc = join_chains(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(host_dst, ctx->accum, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_upload_accum(PtContext* ctx, const float* host_src, size_t bytes) {
    if (int rc = use(ctx)) return rc;
    if (!host_src) return fail(PT_ERR_INVALID, "host_src is null");
    if (!ctx->accum) return fail(PT_ERR_STATE, "no frame allocated");
    const size_t have = (size_t)ctx->width * ctx->height * sizeof(float4);
    if (bytes != have) return fail(PT_ERR_INVALID, "checkpoint is %zu bytes, accumulation buffer is %zu", bytes, have);
    if (int rc = join_chains(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->accum, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_flush(PtContext* ctx) {
    if (int rc = use(ctx)) return rc;
    return join_chains(ctx);
}

int pt_set_frames_in_flight(PtContext* ctx, uint32_t n) {
    if (int rc = use(ctx)) return rc;
    if (n < 1 || n > PT_MAX_CHAINS) return fail(PT_ERR_INVALID, "frames in flight must be 1..%u, got %u", PT_MAX_CHAINS, n);
    if (int rc = join_chains(ctx)) return rc;
    // Chains that share a hardware queue serialise (4 chains on HIP's default 4 queues: 7.6 Gsamples/s where 3
    // run at 10.3), so the request is granted up to what GPU_MAX_HW_QUEUES leaves room for: 3 with the default,
    // queues - 4 beyond it.  pt_frames_in_flight() tells the caller what it got.
    ctx->flight = n < ctx->burst_chains ? n : ctx->burst_chains;
    if (ctx->flight > 1u) return drop_ahead(ctx, false);   // frames in flight take over from render-ahead
    return PT_OK;
}

int pt_set_render_ahead(PtContext* ctx, uint32_t on) {
    if (int rc = use(ctx)) return rc;
    ctx->ahead_on = on != 0u;
    return on ? PT_OK : drop_ahead(ctx, false);
}

int pt_render_ahead_stats(PtContext* ctx, uint64_t* launched, uint64_t* used) {
    pt_clear_error();
    if (!ctx) return fail(PT_ERR_INVALID, "null context");
    if (launched) *launched = ctx->ahead_launched;
    if (used) *used = ctx->ahead_used;
    return PT_OK;
}

uint64_t pt_path_launches(PtContext* ctx) {
    pt_clear_error();
    return ctx ? ctx->path_launches : 0u;
}

uint32_t pt_frames_in_flight(PtContext* ctx) {
    pt_clear_error();
    return ctx ? ctx->flight : 0u;
}

int pt_sync(PtContext* ctx) {
    if (int rc = use(ctx)) return rc;
    if (int rc = join_chains(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_reduce(PtContext* ctx, void* nccl_comm, int root) {
    if (int rc = use(ctx)) return rc;
    if (!nccl_comm) return fail(PT_ERR_INVALID, "communicator is null");
    if (!ctx->accum) return fail(PT_ERR_STATE, "no frame allocated");
    // ncclResult_t ncclReduce(send, recv, count, datatype, op, root, comm, stream); ncclAllReduce without root
    using reduce_fn = int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
    using allreduce_fn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static void* fn_reduce = nullptr;
    static void* fn_allreduce = nullptr;
    if (!fn_reduce || !fn_allreduce) {
        fn_reduce = dlsym(RTLD_DEFAULT, "ncclReduce");
        fn_allreduce = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!fn_reduce || !fn_allreduce) {
            if (void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL)) {
                fn_reduce = dlsym(h, "ncclReduce");
                fn_allreduce = dlsym(h, "ncclAllReduce");
            }
        }
        if (!fn_reduce || !fn_allreduce) return fail(PT_ERR_STATE, "RCCL is not loaded in this process (ncclReduce / ncclAllReduce not found)");
    }
    if (int rc = join_chains(ctx)) return rc;
    const size_t count = (size_t)ctx->width * ctx->height * 4u;
    const int nccl_float = 7, nccl_sum = 0;  // ncclFloat32, ncclSum (rccl.h)
    const int res = root >= 0 ? ((reduce_fn)fn_reduce)(ctx->accum, ctx->accum, count, nccl_float, nccl_sum, root, nccl_comm, ctx->stream)
                              : ((allreduce_fn)fn_allreduce)(ctx->accum, ctx->accum, count, nccl_float, nccl_sum, nccl_comm, ctx->stream);
    if (res != 0) return fail(PT_ERR_HIP, "RCCL %s failed with ncclResult_t %d", root >= 0 ? "ncclReduce" : "ncclAllReduce", res);
    return PT_OK;
}

int pt_reduce_scatter(PtContext* ctx, void* nccl_comm, uint32_t rank, uint32_t world, uint32_t* first_pixel, uint32_t* pixel_count) {
    if (int rc = use(ctx)) return rc;
    if (!nccl_comm) return fail(PT_ERR_INVALID, "communicator is null");
    if (!ctx->accum) return fail(PT_ERR_STATE, "no frame allocated");
    if (world == 0 || rank >= world) return fail(PT_ERR_INVALID, "rank %u of %u", rank, world);
    const uint64_t npix = (uint64_t)ctx->width * ctx->height;
    if (npix % world != 0) return fail(PT_ERR_INVALID, "pt_reduce_scatter works in place on equal slices: %llu pixels are not divisible by %u ranks (use pt_reduce)", (unsigned long long)npix, world);
    // ncclResult_t ncclReduceScatter(send, recv, recvcount, datatype, op, comm, stream); in place: recv == send + rank * recvcount
    using rs_fn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static void* fn = nullptr;
    if (!fn) {
        fn = dlsym(RTLD_DEFAULT, "ncclReduceScatter");
        if (!fn) if (void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL)) fn = dlsym(h, "ncclReduceScatter");
        if (!fn) return fail(PT_ERR_STATE, "RCCL is not loaded in this process (ncclReduceScatter not found)");
    }
    if (int rc = join_chains(ctx)) return rc;
    const uint64_t slice = npix / world;
    const int nccl_float = 7, nccl_sum = 0;  // ncclFloat32, ncclSum (rccl.h)
    const int res = ((rs_fn)fn)(ctx->accum, ctx->accum + (size_t)rank * slice, (size_t)slice * 4u, nccl_float, nccl_sum, nccl_comm, ctx->stream);
    if (res != 0) return fail(PT_ERR_HIP, "RCCL ncclReduceScatter failed with ncclResult_t %d", res);
    if (first_pixel) *first_pixel = (uint32_t)(rank * slice);
    if (pixel_count) *pixel_count = (uint32_t)slice;
    return PT_OK;
}

void* pt_accum_device_ptr(PtContext* ctx) { return ctx ? (void*)ctx->accum : nullptr; }
void* pt_display_device_ptr(PtContext* ctx) { return ctx ? ctx->display : nullptr; }

int pt_primary_ids(PtContext* ctx, uint32_t* host_obj, uint32_t* host_mat) {
    if (int rc = use(ctx)) return rc;
    if (!ctx->have_scene) return fail(PT_ERR_STATE, "no scene uploaded");
    if (!ctx->have_camera) return fail(PT_ERR_STATE, "no camera set");
    if (ctx->cd.width != ctx->width || ctx->cd.height != ctx->height)  // the kernel iterates the camera's pixels
        return fail(PT_ERR_STATE, "frame was resized to %ux%u after the camera was set for %ux%u", ctx->width, ctx->height, ctx->cd.width, ctx->cd.height);
    const size_t npix = (size_t)ctx->width * ctx->height;
    if (!ctx->ids) HIP_TRY(hipMalloc((void**)&ctx->ids, 2 * npix * sizeof(uint32_t)));
    HIP_TRY(pt_launch_primary_ids(ctx->sd, ctx->cd, ctx->ids, ctx->ids + npix, ctx->stream));
    if (host_obj) HIP_TRY(hipMemcpyAsync(host_obj, ctx->ids, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (host_mat) HIP_TRY(hipMemcpyAsync(host_mat, ctx->ids + npix, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_counters(PtContext* ctx, PtCounters* out) {
    if (int rc = use(ctx)) return rc;
    if (!out) return fail(PT_ERR_INVALID, "out is null");
    static_assert(sizeof(PtCounters) == 12 * sizeof(uint64_t), "PtCounters is 12 u64 in kernel order");
    if (int rc = join_chains(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->counters, sizeof(PtCounters), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_reset_counters(PtContext* ctx) {
    if (int rc = use(ctx)) return rc;
    if (int rc = join_chains(ctx)) return rc;
    HIP_TRY(hipMemsetAsync(ctx->counters, 0, (12 + 64 * PT_HISTOGRAM_BINS) * sizeof(uint64_t), ctx->stream));
    return PT_OK;
}

int pt_path_length_histogram(PtContext* ctx, uint64_t* out_bins) {
    if (int rc = use(ctx)) return rc;
    if (!out_bins) return fail(PT_ERR_INVALID, "out_bins is null");
    if (int rc = join_chains(ctx)) return rc;
    std::vector<uint64_t> copies(64 * PT_HISTOGRAM_BINS);   // the kernel keeps 64 copies (one per blockIdx % 64)
    HIP_TRY(hipMemcpyAsync(copies.data(), ctx->counters + 12, copies.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t b = 0; b < PT_HISTOGRAM_BINS; ++b) {
        out_bins[b] = 0;
        for (uint32_t c = 0; c < 64u; ++c) out_bins[b] += copies[c * PT_HISTOGRAM_BINS + b];
    }
    return PT_OK;
}

int pt_valu_peak(PtContext* ctx, uint32_t packed, uint32_t iters, double* out_tflops) {
    if (int rc = use(ctx)) return rc;
    if (!out_tflops || iters == 0) return fail(PT_ERR_INVALID, "bad arguments");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 8u;  // 8 blocks x 4 waves = 8 waves / SIMD
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = pt_launch_valu_peak(ctx->sink, packed, iters, blocks, ctx->stream);  // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) e = pt_launch_valu_peak(ctx->sink, packed, iters, blocks, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(PT_ERR_HIP, "ceiling micro-benchmark failed: %s", hipGetErrorString(e));
    const double flops = (double)blocks * 256.0 * (double)iters * 16.0 * 2.0;
    *out_tflops = flops / ((double)ms * 1e-3) / 1e12;
    return PT_OK;
}

int pt_make_current(PtContext* ctx) {
    pt_clear_error();
    g_current = ctx;
    return PT_OK;
}

void launchRenderKernel(void* output, uint32_t width, uint32_t height, uint32_t frame_count, PtDim3 /*grid*/, PtDim3 /*block*/) {
    pt_clear_error();
    PtContext* ctx = g_current;
    if (!ctx) { fail(PT_ERR_STATE, "launchRenderKernel: no current context (pt_make_current)"); return; }
    if (!output) { fail(PT_ERR_INVALID, "launchRenderKernel: output is null"); return; }
    if (width != ctx->width || height != ctx->height) { fail(PT_ERR_INVALID, "launchRenderKernel: %ux%u does not match the camera's %ux%u", width, height, ctx->width, ctx->height); return; }
    // The reference's loop works on the legacy default stream throughout: renderKernel<<<grid, block>>> and then
    // convertToRGBAKernel on the same `output` (raytracer_kernel.cu:326, Window.cuh:155-160).  A context whose stream
    // nobody chose therefore moves to the null stream on its first call through this shim, so that the caller's own
    // kernels and copies on `output` stay ordered with the launch as they were (the context's own stream is
    // non-blocking: nothing would order them).  pt_set_stream / pt_use_own_stream keep their say.
    const hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { fail(PT_ERR_HIP, "launchRenderKernel: hipSetDevice(%d) failed: %s", ctx->device, hipGetErrorString(e)); return; }
    if (!ctx->stream_chosen && ctx->stream == ctx->own_stream) {
        if (end_shim_binding(ctx) != PT_OK || switch_stream(ctx, nullptr) != PT_OK) return;
    }
    // The caller's `output` becomes the accumulation buffer for this call.  Without frames in flight the context's own
    // buffer is bound again before the call returns.  With them the binding stays while consecutive calls name the same
    // buffer, so that the frames can be collected into one launch (a caller that does not read `output` between its
    // launches gets the multi-frame launch; pt_flush() before it reads); the next entry point other than this one
    // folds what is held back into `output` and undoes the binding (use()).
    if (!(ctx->shim_active && ctx->accum == (float4*)output)) {
        if (end_shim_binding(ctx) != PT_OK) return;
        float4* const saved = ctx->accum;
        const bool saved_bound = ctx->accum_bound;
        if (bind_accum(ctx, output) != PT_OK) return;
        ctx->shim_active = true; ctx->shim_saved = saved; ctx->shim_saved_bound = saved_bound;
    }
    PtRenderParams p{};
    p.integrator = PT_INTEGRATOR_PATH; p.max_depth = 50; p.rr_depth = 3; p.seed = 0x5EED0001u;  // :294, :233
    p.acc_mode = PT_ACC_RUNNING_MEAN;
    (void)render(ctx, frame_count, 1, &p);  // a failure leaves its message in pt_last_error()
    if (ctx->flight <= 1u) {   // nothing is held back: the binding ends here (join_chains inside finds nothing to do)
        std::string msg = g_err;
        (void)end_shim_binding(ctx);
        g_err = msg;
    }
}

}  // extern "C"
