"""bench.py — Msamples/s of the per-pixel render path on N MI355X (one process per GPU).

Headline workload (BASELINE.json configs[1], the configuration the metric is quoted on that fits one
GPU): the Cornell box of cuda_pathtracer/src/main.cu at 1280x720, progressive rendering at
1 sample per pixel per frame with the display (RGBA8) made for every frame.
A "step" is one frame: one pt_render call over all 921 600 pixels, the call the reference's frame loop makes
(launchRenderKernel, main.cu:287).  With frames in flight (the default here) the library collects consecutive frames
into one launch of the path kernel (up to 64; include/pt.h) and folds them into the accumulation buffer one by one,
making every frame's RGBA8 display on the way: same sums and same pictures as one launch per frame, bit for bit.
  N = 1 : K frames on one GPU.
  N > 1 : sample-split, weak scaling — every rank renders K frames of 1 spp for its own
          disjoint sample indices, then the float4 accumulation buffers are combined once with an
          RCCL reduce-scatter (every rank receives and adds the sums of 1/N of the pixels), each rank
          resolves its own slice and the 4-byte pixels are all-gathered (inside the timed region;
          `render_ms`, `reduce_ms` and `display_ms` report the parts on rank 0's stream from HIP events
          -- a rank that ends its frames early waits inside the reduce-scatter for the slowest -- with no
          host synchronisation between the render and the closing barrier).
BASELINE's metric is quoted "at 1280x720 & 4K", and north_star's scaling target is defined on
config 3 (4K at 1024 spp, sample-split with one RCCL sum-reduce), so every run also carries that
workload as a second leg under `legs`: 3840x2160, 16 spp per step, 64 steps = 1024 spp, the steps
DIVIDED among the ranks (strong scaling), wall time including the reduce.  At N = 1 it is the
single-GPU 4K figure the N > 1 runs compare with.  (`--no-4k-leg` leaves it out.)

Scene, camera and accumulation buffer are resident in HBM before the timed region; the
display buffer is never copied to the host inside it (the reference's per-frame D2H + XPutImage
is the X11 path, out of scope).

One JSON line on rank 0.  `roofline` prices the path kernel against the FP32 vector peak
(north_star: "FP32 VALU-bound, not a dense contraction, so no MFMA"): achieved = algorithmic
FLOPs per launch (event counters x the fixed cost table of SURVEY §8(d)) / time per launch from HIP
events on the launch stream around the K launches (the overlapped launches fork from and join that
stream, so this is first start to last end / K, the figure tools/trace_span.py derives from a
rocprofv3 kernel trace of the same command).  Beside it the same frames one at a time, without frames in flight:
`kernel_ms_serial` / `frac_serial` strictly serial (render-ahead off: what rocprofv3 shows per dispatch) and
`frame_ms_render_ahead` / `frac_render_ahead` as the library runs such a loop by default (the next frame rendered
ahead, pt_set_render_ahead; same bits).  `cpu_baseline` times the CPU oracle
(oracle/libpt_oracle.so, OpenMP) on a bounded number of frames of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (vector), spec
FP64_VALU_PEAK_TFLOPS = 78.6   # the FP64 vector rate is half the FP32 vector rate on CDNA (v_fma_f64 issues over twice the cycles); AMD's MI355X figure

WORKLOADS = {
    # name: (scene, width, height, spp per step, max_depth)
    "cornell_720p_progressive_1spp": ("cuda_default", 1280, 720, 1, 50),
    "cornell_4k_16spp_per_step": ("cuda_default", 3840, 2160, 16, 50),
    "glass_stress_720p_16spp_per_step": ("glass_stress", 1280, 720, 16, 32),
    "random4096_4k_1spp_per_step": ("random4096", 3840, 2160, 1, 16),
    # the scene sizes between the Cornell box and config 5 that the launch heuristics are swept on (tools/bench_sweep.sh)
    "spheres150_720p_1spp_per_step": ("random149", 1280, 720, 1, 16),
    "spheres1000_360p_1spp_per_step": ("random999", 640, 360, 1, 16),
}
LEG_4K = ("cornell_4k_16spp_per_step", 64, 2)  # BASELINE config 3: 64 steps of 16 spp = 1024 spp; 2 warm-up steps


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="cornell_720p_progressive_1spp", choices=sorted(WORKLOADS))
    ap.add_argument("--mis", action="store_true", help="next-event estimation + MIS (north_star variant)")
    ap.add_argument("--pixels-per-wave", type=int, default=0)
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=tuple(range(1, 9)),
                    help="launches overlapped on internal streams, each holding up to 64 consecutive frames (include/pt.h); "
                         "1 = one serial launch per frame.  2 is enough (1 024 frames: 12.1 Gsamples/s with 2 or 4, 11.95 with 3); "
                         "up to 3 run on HIP's default of 4 hardware queues per process")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg; 0 disables it")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="headline leg only. weak: every rank renders K steps (default, the driver's contract); strong: the K "
                         "steps are divided among the ranks")
    ap.add_argument("--no-4k-leg", action="store_true", help="leave out the 4K / 1024 spp strong-scaling leg (BASELINE config 3)")
    ap.add_argument("--no-direct-leg", action="store_true", help="leave out the FP64 `direct` leg (BASELINE config 1: cpu_default, 1280x720, 64 spp; N = 1 only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks over gloo, all on GPU 0: exercises the multi-rank code path on a one-GPU box (the "
                         "numbers mean nothing)")
    args = ap.parse_args()

    # more than 3 chains need more hardware queues than HIP's default of 4 per process (the library grants only what
    # the variable leaves room for, pt_set_frames_in_flight); the default of 2 runs on the environment as it is
    if args.frames_in_flight > 3 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < args.frames_in_flight + 4:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.frames_in_flight + 4)
    import torch
    import torch.distributed as dist

    from path_tracer_amd import abi, scenes
    from path_tracer_amd.distributed import gather_display, pixel_slice, progressive_schedule, reduce_scatter_accum, strong_schedule
    from path_tracer_amd.renderer import Renderer, algorithmic_flops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    def stage(msg):
        if os.environ.get("BENCH_TRACE"):  # a file name prefix: one line per stage and rank
            with open(f"{os.environ['BENCH_TRACE']}_rank{rank}.log", "a") as f:
                f.write(f"{time.perf_counter():.3f} {msg}\n")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}`", file=sys.stderr)
            return 2
        args.gpus = world
    rehearsal = args.rehearse_on_one_gpu
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    stage("process group up")

    def barrier():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    flags = abi.FLAG_NEE_MIS if args.mis else 0
    # everything below -- the library's launches, the timing events, the reduce -- runs on one stream of torch's that
    # is not the null stream (which synchronises with every other blocking stream of the process)
    torch.cuda.set_stream(torch.cuda.Stream())
    r = Renderer(local_rank)
    r.use_torch_stream()

    def run_leg(workload, K, Wm, scaling, serial_probe):
        """One workload: set-up, Wm untimed warm-up steps, then K timed steps bracketed by barrier + synchronize on
        both sides (the reduce and rank 0's resolve inside), MAX over ranks.  Returns the numbers of the leg."""
        scene_name, W, H, spp, max_depth = WORKLOADS[workload]
        scene, cam = scenes.builtin(scene_name, W, H)
        params = abi.render_params(max_depth=max_depth, flags=flags, fused_rgba8=1, pixels_per_wave=args.pixels_per_wave)
        if r._bound is not None:
            r.unbind()
        r.set_frames_in_flight(1)
        r.set_scene(scene)
        r.set_camera(cam)
        accum = torch.zeros((H, W, 4), device="cuda", dtype=torch.float32)
        r.bind_tensor(accum)
        r.set_frames_in_flight(args.frames_in_flight)
        flight = r.frames_in_flight()

        if scaling == "strong":   # warm-up on every rank; then this rank's contiguous share of the K timed steps
            sched = progressive_schedule(0, Wm, world, rank, spp) + strong_schedule(Wm * world * spp, K, world, rank, spp)
        else:
            sched = progressive_schedule(0, Wm + K, world, rank, spp)

        # set-up, not a step: one launch loads the kernels and allocates the chains' buffers (hipMalloc on first
        # use), then the buffer is cleared again
        for k in range(max(flight, 2)):
            r.render(k * spp, spp, params)
        r.flush()
        torch.cuda.synchronize()
        accum.zero_()
        torch.cuda.synchronize()
        stage(f"{workload}: set-up done")
        for s0, n in sched[:Wm]:
            r.render(s0, n, params)
        r.flush()
        npix = W * H
        chunk = -(-npix // world)
        my_first, my_count = pixel_slice(npix, world, rank)
        disp_slice = torch.zeros(chunk, device="cuda", dtype=torch.int32) if world > 1 else None
        if world > 1:  # warm the collectives too (on a copy: the buffer keeps its samples)
            reduce_scatter_accum(accum.clone())
            gather_display(disp_slice, npix)
        barrier()
        stage(f"{workload}: timed region starts")
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launches0 = r.path_launches()
        t0 = time.perf_counter()
        ev0.record()
        for s0, n in sched[Wm:]:
            r.render(s0, n, params)
        r.flush()  # frames in flight: launches what is held back; the launch stream waits for the last fold pass (asynchronous)
        ev1.record()
        launches = max(r.path_launches() - launches0, 1)
        ev2 = ev3 = None
        picture = None
        if world > 1:
            # Combine: reduce-scatter of the float4 buffers (every rank receives and adds the sums of 1/N of the
            # pixels), each rank resolves its own slice, the 4-byte pixels are gathered.  Everything is enqueued on this
            # stream -- torch's collectives make the current stream wait, the library's resolve runs on it -- and timed with
            # events: no host synchronisation between the render and the barrier that ends the timed region.
            ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            first, count = reduce_scatter_accum(accum)
            ev2.record()
            if count:
                r.resolve_range(abi.RESOLVE_RGBA8_CUDA, first, count, disp_slice)
            picture = gather_display(disp_slice, npix)
            ev3.record()
        else:
            r.resolve(abi.RESOLVE_RGBA8_CUDA)
        barrier()
        t_end = time.perf_counter()
        elapsed = t_end - t0
        stage(f"{workload}: timed region done")
        total_steps = K if scaling == "strong" else world * K
        # every pixel of this rank's slice (the whole frame at N = 1) now holds the samples of every rank (warm-up steps
        # included), each once: the count channel is known analytically and checked, not trusted
        want_w = float((world * Wm + total_steps) * spp)
        if my_count:
            mine = accum.view(-1, 4)[my_first:my_first + my_count, 3]
            w_min, w_max = float(mine.min()), float(mine.max())
            if w_min != want_w or w_max != want_w:
                raise RuntimeError(f"{workload}: rank {rank}'s pixels hold {w_min}..{w_max} samples, expected {want_w}")
        if picture is not None and rank == 0 and int((picture != 0).sum()) == 0:
            raise RuntimeError(f"{workload}: the gathered picture is empty")
        my_steps = len(sched) - Wm
        # HIP events on the launch stream around this rank's frames (the chains fork from and join this stream): first
        # launch start to last fold pass end, per frame -- the path kernel with its launches overlapping, plus the fold
        # pass of the last launch and the queue latencies between them
        frame_ms = ev0.elapsed_time(ev1) / max(my_steps, 1)
        serial_ms = ahead_ms = None
        if serial_probe and flight > 1 and my_steps > 0:
            # the same launches strictly one after another, for comparison with rocprofv3's per-dispatch durations
            r.set_frames_in_flight(1)
            n_serial = min(my_steps, 128)
            probe = {}
            for ahead in (False, True):
                # strictly one launch after the other (what rocprofv3's per-dispatch durations show), and the library's
                # default for a caller that renders one frame at a time: the next frame rendered ahead (pt_set_render_ahead)
                r.set_render_ahead(ahead)
                for s0, n in sched[Wm:Wm + 4]:    # untimed: starts render-ahead (its planes are allocated on first use)
                    r.render(s0, n, params)
                torch.cuda.synchronize()
                sv0, sv1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # (ev2 / ev3 are the combine's)
                sv0.record()
                for s0, n in sched[Wm + 4:Wm + 4 + n_serial]:
                    r.render(s0, n, params)
                sv1.record()
                torch.cuda.synchronize()
                probe[ahead] = sv0.elapsed_time(sv1) / max(len(sched[Wm + 4:Wm + 4 + n_serial]), 1)
            serial_ms, ahead_ms = probe[False], probe[True]
        # the parts of the timed region, from the events (device time on this rank's stream): render = first launch to last
        # fold pass; reduce = the reduce-scatter behind it; display = this rank's resolve + the all-gather of the pictures
        render_s = ev0.elapsed_time(ev1) * 1e-3 if world > 1 else elapsed
        reduce_s = ev1.elapsed_time(ev2) * 1e-3 if world > 1 else 0.0
        display_s = ev2.elapsed_time(ev3) * 1e-3 if world > 1 else 0.0
        if world > 1:
            # MAX over ranks of the timed region and of the frame time; the three parts stay rank 0's own (they follow one
            # another on its stream, so they add up to no more than the region; a rank that finishes its frames early
            # spends the difference inside the reduce-scatter, waiting for the slowest)
            t = torch.tensor([elapsed, frame_ms], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, frame_ms = (float(v) for v in t)
        out = {"workload": workload, "scene": scene_name, "width": W, "height": H, "spp_per_step": spp, "max_depth": max_depth,
               "steps": K, "warmup": Wm, "scaling": scaling if world > 1 else "weak", "frames_in_flight": flight,
               "elapsed_s": elapsed, "total_steps": total_steps, "samples_per_step": W * H * spp,
               "value": total_steps * W * H * spp / elapsed / 1e6, "frame_ms": frame_ms, "launches": launches, "serial_ms": serial_ms, "ahead_ms": ahead_ms,
               "render_ms": render_s * 1e3, "reduce_ms": reduce_s * 1e3, "display_ms": display_s * 1e3, "my_steps": my_steps,
               "reduce_bytes": (world - 1) * chunk * 16 if world > 1 else 0, "gather_bytes": (world - 1) * chunk * 4 if world > 1 else 0}
        if rank == 0:
            # ---- algorithmic FLOPs per launch from the instrumented kernel (same samples as timed steps) --
            r.unbind()
            r.set_frames_in_flight(1)
            r.reset()
            r.reset_counters()
            cp = abi.render_params(max_depth=max_depth, flags=flags | abi.FLAG_COUNTERS, pixels_per_wave=args.pixels_per_wave)
            n_cnt = max(1, min(my_steps, 8 if W * H * spp < (1 << 24) else 1))
            for s0, n in sched[Wm:Wm + n_cnt]:
                r.render(s0, n, cp)
            cnt = r.counters().as_dict()
            out["flops_per_frame"] = algorithmic_flops(cnt, scene.num_spheres, scene.num_planes, scene.num_triangles, W * H * n_cnt) / n_cnt
            out["rays_per_sample"] = cnt["rays"] / max(cnt["samples"], 1)
        out["_scene"], out["_cam"], out["_params"] = scene, cam, params
        return out

    def run_direct_leg(steps=4, cpu_seconds=4.0):
        """BASELINE config 1: the CPU program's own scene and integrator (cpu_pathtracer/src/main.cpp:14-53,63-131) on the GPU, FP64:
        cpu_default, 1280x720, one step = one pt_render call of 64 spp; the oracle's restatement of renderRegion beside it."""
        from path_tracer_amd.renderer import direct_flops
        W, H, spp = 1280, 720, 64
        scene, cam = scenes.cpu_default(W, H)
        p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
        if r._bound is not None:
            r.unbind()
        r.set_frames_in_flight(1)
        r.set_scene(scene)
        r.set_camera(cam)
        r.reset()
        r.render(0, spp, p)   # warm-up: loads the kernel
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for k in range(steps):
            r.render((1 + k) * spp, spp, p)
        ev1.record()
        r.resolve(abi.RESOLVE_RGBA8_CPU)    # the CPU app's accumulate + gamma + pack (main.cpp:223-237)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / steps
        w = r.accum()[..., 3]
        if float(w.min()) != (steps + 1) * spp or float(w.max()) != (steps + 1) * spp:
            raise RuntimeError("direct leg: wrong sample count in the accumulation buffer")
        r.reset_counters()
        r.render(spp, spp, abi.render_params(integrator=abi.INTEGRATOR_DIRECT, flags=abi.FLAG_COUNTERS))
        cnt = r.counters().as_dict()
        flops = direct_flops(cnt, scene.num_spheres, scene.num_planes, scene.num_triangles, W * H)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        leg = {"baseline_config": 1, "metric": "Msamples/s", "value": round(steps * W * H * spp / elapsed / 1e6, 3), "n_gpus": 1,
               "scene": "cpu_default", "integrator": "direct", "dtype": "f64", "width": W, "height": H, "spp_per_step": spp, "steps": steps, "warmup": 1,
               "ms_per_step": round(elapsed / steps * 1e3, 4), "mrays_per_s": round(steps * W * H * spp / elapsed / 1e6 * cnt["rays"] / cnt["samples"], 1),
               "rays_per_sample": round(cnt["rays"] / cnt["samples"], 4),
               "roofline": {"bound": "valu_fp64", "kernel": "pt_direct_kernel", "achieved": round(achieved, 4), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / FP64_VALU_PEAK_TFLOPS, 5), "traffic": None, "flops_per_launch": flops,
                            "flops_per_sample": round(flops / (W * H * spp), 1), "kernel_ms": round(kernel_ms, 4)}}
        if cpu_seconds > 0:
            from tests.oracle_binding import load_oracle
            o = load_oracle()
            t = time.perf_counter()
            acc = o.render_direct(scene, cam, 0, 1, p)
            one = time.perf_counter() - t
            n = max(1, min(64, int(cpu_seconds / max(one, 1e-3))))
            t = time.perf_counter()
            o.render_direct(scene, cam, 1, n, p, accum=acc)
            dt = time.perf_counter() - t
            try:
                import ctypes as C
                cores = int(C.CDLL("libgomp.so.1").omp_get_max_threads())
            except OSError:
                cores = os.cpu_count() or 1
            leg["cpu_baseline"] = {"value": round(W * H * n / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                   "sample": f"{n} spp of {W}x{H} ({W * H * n} samples, {dt:.1f} s), oracle/libpt_oracle.so pto_render_direct "
                                             "(the restatement of renderRegion, FP64, OpenMP over rows)"}
        return leg

    try:
        head = run_leg(args.workload, args.steps, args.warmup, args.scaling, serial_probe=True)
        leg4k = None
        if not args.no_4k_leg and not args.mis:
            leg4k = run_leg(LEG_4K[0], LEG_4K[1], LEG_4K[2], "strong", serial_probe=False)
        leg_direct = None
        if world == 1 and not args.no_direct_leg and not args.mis:
            leg_direct = run_direct_leg(cpu_seconds=min(4.0, args.cpu_seconds))
    except RuntimeError as e:
        print(f"bench.py: {e}", file=sys.stderr)
        return 3

    def roofline(leg, with_traffic):
        # per launch of the path kernel: a launch holds my_steps / launches frames (1 when the launches are serial)
        fpl = leg["my_steps"] / leg["launches"]
        flops_per_launch = leg["flops_per_frame"] * fpl
        launch_ms = leg["frame_ms"] * fpl   # first start to last end of this rank's launches / launches
        achieved = flops_per_launch / (launch_ms * 1e-3) / 1e12
        d = {"bound": "valu", "kernel": "pt_path_kernel", "achieved": round(achieved, 4), "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 5), "traffic": None, "flops_per_launch": flops_per_launch,
             "flops_per_sample": round(leg["flops_per_frame"] / leg["samples_per_step"], 1),
             "launches": leg["launches"], "steps_per_launch": round(fpl, 3),
             "kernel_ms": round(launch_ms, 5), "frame_ms": round(leg["frame_ms"], 5), "launch_overlap": leg["frames_in_flight"]}
        if with_traffic:
            pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
            if os.path.exists(pmc):
                try:
                    per_frame = json.load(open(pmc)).get(leg["workload"], {}).get("hbm_bytes_per_frame")   # PMC, path launch of many frames
                    d["traffic"] = round(per_frame * fpl) if per_frame else None
                except Exception:
                    pass
        if leg["serial_ms"]:
            d["kernel_ms_serial"] = round(leg["serial_ms"], 5)
            d["frac_serial"] = round(leg["flops_per_frame"] / (leg["serial_ms"] * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 5)
        if leg.get("ahead_ms"):   # one frame at a time with the next one rendered ahead (the library's default without frames in flight)
            d["frame_ms_render_ahead"] = round(leg["ahead_ms"], 5)
            d["frac_render_ahead"] = round(leg["flops_per_frame"] / (leg["ahead_ms"] * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 5)
        return d

    result = None
    if rank == 0:
        K = args.steps
        result = {
            "metric": "Msamples/s", "value": round(head["value"], 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": K, "warmup": args.warmup, "ms_per_step": round(head["elapsed_s"] / K * 1e3, 5), "higher_is_better": True,
            "scaling": head["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": head["workload"], "scene": head["scene"], "width": head["width"], "height": head["height"],
                       "spp_per_step": head["spp_per_step"], "max_depth": head["max_depth"], "integrator": "path+nee_mis" if args.mis else "path",
                       "accumulate": "sum+display_rgba8_per_frame", "frames_in_flight": head["frames_in_flight"],
                       "parallelism": f"sample_split{world}" if world > 1 else "single",
                       "reduce": ("gloo_rehearsal_on_one_gpu" if rehearsal else "rccl_reduce_scatter_sum_once+resolve_own_slice+all_gather_rgba8") if world > 1 else None},
            "mrays_per_s": round(head["value"] * head["rays_per_sample"], 1), "rays_per_sample": round(head["rays_per_sample"], 4),
            "roofline": roofline(head, True),
        }
        if world > 1:  # the collective and the rest of the timed region, apart (rank 0's stream; the region itself is the MAX over ranks)
            result["render_ms"] = round(head["render_ms"], 4)
            result["reduce_ms"] = round(head["reduce_ms"], 4)
            result["display_ms"] = round(head["display_ms"], 4)
            result["reduce_bytes_per_rank"] = head["reduce_bytes"]
            result["gather_bytes_per_rank"] = head["gather_bytes"]
            result["host_and_barrier_ms"] = round(head["elapsed_s"] * 1e3 - head["render_ms"] - head["reduce_ms"] - head["display_ms"], 4)
        if leg4k:
            l = leg4k
            result["legs"] = {"cornell_4k_1024spp_strong_split": {
                "baseline_config": 3, "metric": "Msamples/s", "value": round(l["value"], 3), "mrays_per_s": round(l["value"] * l["rays_per_sample"], 1),
                "n_gpus": world, "scaling": "strong", "width": l["width"], "height": l["height"], "spp_total": l["steps"] * l["spp_per_step"],
                "spp_per_step": l["spp_per_step"], "steps": l["steps"], "steps_per_rank_max": -(-l["steps"] // world), "warmup": l["warmup"],
                "wall_ms": round(l["elapsed_s"] * 1e3, 3), "render_ms": round(l["render_ms"], 3), "reduce_ms": round(l["reduce_ms"], 3),
                "display_ms": round(l["display_ms"], 3), "reduce_bytes_per_rank": l["reduce_bytes"], "gather_bytes_per_rank": l["gather_bytes"],
                "frames_in_flight": l["frames_in_flight"],
                "roofline": roofline(l, False)}}
        if leg_direct:
            result.setdefault("legs", {})["cpu_default_direct_720p_64spp"] = leg_direct
        if world == 1 and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline(head["_scene"], head["_cam"], head["_params"], head["spp_per_step"], args.cpu_seconds)
    if world > 1:
        dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
    r.close()
    return 0


def cpu_baseline(scene, cam, params, spp, budget_s):
    """The CPU oracle (a port: plain C + OpenMP, same integrator, same RNG) timed on this box's
    host cores on a bounded number of frames of the same workload.  A reported baseline only."""
    import ctypes as C

    from path_tracer_amd import abi
    from tests.oracle_binding import load_oracle
    o = load_oracle()
    try:
        omp = C.CDLL("libgomp.so.1")
        cores = int(omp.omp_get_max_threads())
    except OSError:
        cores = os.cpu_count() or 1
    p = abi.render_params(max_depth=params.max_depth, flags=params.flags & ~abi.FLAG_COUNTERS)
    t = time.perf_counter()
    acc = o.render_path(scene, cam, 0, spp, p, precision=32)
    one = time.perf_counter() - t
    frames = max(1, min(512, int(budget_s / max(one, 1e-3)) - 1))
    t = time.perf_counter()
    for f in range(frames):
        o.render_path(scene, cam, (1 + f) * spp, spp, p, precision=32, accum=acc)
    dt = time.perf_counter() - t
    n = cam.width * cam.height * spp * frames
    return {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{frames} frames of {cam.width}x{cam.height} at {spp} spp ({n} samples, {dt:.1f} s), "
                      "oracle/libpt_oracle.so float build (gcc -O2 -ffp-contract=off -fopenmp: the checker's own build, "
                      "made in the build container; -march=native cannot travel to the GPU box)"}


if __name__ == "__main__":
    sys.exit(main())
