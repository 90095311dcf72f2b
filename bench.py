"""bench.py — Msamples/s of the per-pixel render path on N MI355X (one process per GPU).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on that fits one
GPU): the Cornell box of cuda_pathtracer/src/main.cu at 1280x720, progressive rendering at
1 sample per pixel per frame with the display (RGBA8) resolve fused into every frame.
A "step" is one frame: one launch of the path kernel over all 921 600 pixels.
  N = 1 : K frames on one GPU.
  N > 1 : sample-split, weak scaling — every rank renders K frames of 1 spp for its own
          disjoint sample indices, then the float4 accumulation buffers are summed once with an
          RCCL reduce to rank 0 (inside the timed region), which resolves the image.
          `--scaling strong` divides the K steps among the ranks instead (BASELINE config 3, e.g.
          `--workload cornell_4k_16spp_per_step --steps 64` = 4K at 1024 spp split N ways).
Scene, camera and accumulation buffer are resident in HBM before the timed region; the
display buffer is never copied to the host inside it (the reference's per-frame D2H + XPutImage
is the X11 path, out of scope).

One JSON line on rank 0.  `roofline` prices the path kernel against the FP32 vector peak
(north_star: "FP32 VALU-bound, not a dense contraction, so no MFMA"): achieved = algorithmic
FLOPs per launch (event counters x the fixed cost table of SURVEY §8(d)) / average launch
duration from HIP events on the launch stream.  `cpu_baseline` times the CPU oracle
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

WORKLOADS = {
    # name: (scene, width, height, spp per step, max_depth)
    "cornell_720p_progressive_1spp": ("cuda_default", 1280, 720, 1, 50),
    "cornell_4k_16spp_per_step": ("cuda_default", 3840, 2160, 16, 50),
    "glass_stress_720p_16spp_per_step": ("glass_stress", 1280, 720, 16, 32),
    "random4096_4k_1spp_per_step": ("random4096", 3840, 2160, 1, 16),
}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="cornell_720p_progressive_1spp", choices=sorted(WORKLOADS))
    ap.add_argument("--mis", action="store_true", help="next-event estimation + MIS (north_star variant)")
    ap.add_argument("--pixels-per-wave", type=int, default=0)
    ap.add_argument("--frames-in-flight", type=int, default=8, choices=tuple(range(1, 9)),
                    help="consecutive launches overlapped on internal streams (include/pt.h); 1 = strictly serial launches")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg; 0 disables it")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: every rank renders K steps (default, the driver's contract); strong: the K steps are "
                         "divided among the ranks (BASELINE config 3: a fixed sample count split across the GPUs)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks over gloo, all on GPU 0: exercises the multi-rank code path on a one-GPU box (the "
                         "number means nothing)")
    args = ap.parse_args()

    # frames in flight need one hardware queue per chain plus a few for the launch stream and torch's own;
    # HIP's default of 4 per process makes chains share a queue (and serialise) from 4 chains on
    # (8 chains: 6.5 Gsamples/s with 8 queues, 9.3 with 12, 9.5 with 16)
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 16:
        os.environ["GPU_MAX_HW_QUEUES"] = "16"
    import torch
    import torch.distributed as dist

    from path_tracer_amd import abi, scenes
    from path_tracer_amd.distributed import progressive_schedule, reduce_accum
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

    scene_name, W, H, spp, max_depth = WORKLOADS[args.workload]
    scene, cam = scenes.builtin(scene_name, W, H)
    flags = abi.FLAG_NEE_MIS if args.mis else 0
    params = abi.render_params(max_depth=max_depth, flags=flags, fused_rgba8=1, pixels_per_wave=args.pixels_per_wave)

    stage("process group up")
    r = Renderer(local_rank)
    r.use_torch_stream()
    r.set_scene(scene)
    r.set_camera(cam)
    accum = torch.zeros((H, W, 4), device="cuda", dtype=torch.float32)
    r.bind_tensor(accum)
    r.set_frames_in_flight(args.frames_in_flight)

    K, Wm = args.steps, args.warmup
    sched = progressive_schedule(0, Wm + K, world, rank, spp)
    if args.scaling == "strong" and world > 1:
        # warm-up as above; then this rank's contiguous share of the K timed steps
        from path_tracer_amd.distributed import partition
        first = Wm * world * spp
        begin, count = partition(K, world, rank)
        sched = sched[:Wm] + [(first + (begin + g) * spp, spp) for g in range(count)]

    def barrier():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    # set-up, not a step: one launch per chain loads the kernels and allocates the chains' buffers
    # (hipMalloc on first use), then the buffer is cleared again
    for k in range(args.frames_in_flight):
        r.render(k, spp, params)
    r.flush()
    torch.cuda.synchronize()
    accum.zero_()
    torch.cuda.synchronize()
    stage("set-up done")
    for s0, n in sched[:Wm]:
        r.render(s0, n, params)
    r.flush()
    stage("warm-up enqueued")
    if world > 1:  # warm the collective too
        reduce_accum(accum.clone())
        stage("collective warmed")
    barrier()
    stage("timed region starts")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for s0, n in sched[Wm:]:
        r.render(s0, n, params)
    r.flush()  # frames in flight: the launch stream waits for every chain and folds their sums (asynchronous)
    ev1.record()
    if world > 1:
        reduce_accum(accum, dst=0)
    if rank == 0:
        r.resolve(abi.RESOLVE_RGBA8_CUDA)
    barrier()
    elapsed = time.perf_counter() - t0
    stage("timed region done")
    if rank == 0:
        # every pixel now holds the samples of every rank (warm-up frames included), each once
        want_w = float((world * Wm + (K if args.scaling == "strong" else world * K)) * spp)
        w_min, w_max = float(accum[..., 3].min()), float(accum[..., 3].max())
        if w_min != want_w or w_max != want_w:
            print(f"bench.py: accumulation buffer holds {w_min}..{w_max} samples per pixel, expected {want_w}", file=sys.stderr)
            return 3
    # HIP events on the launch stream around the K launches (the chains fork from and join this stream):
    # the time per launch the chip needed, with `frames_in_flight` launches overlapping
    my_steps = len(sched) - Wm
    kernel_ms = ev0.elapsed_time(ev1) / max(my_steps, 1)
    # the same launches strictly one after another, for comparison with rocprofv3's per-dispatch durations
    serial_ms = kernel_ms
    if args.frames_in_flight > 1:
        r.set_frames_in_flight(1)
        n_serial = min(my_steps, 128)
        torch.cuda.synchronize()
        ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev2.record()
        for s0, n in sched[Wm:Wm + n_serial]:
            r.render(s0, n, params)
        ev3.record()
        torch.cuda.synchronize()
        serial_ms = ev2.elapsed_time(ev3) / max(n_serial, 1)
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    samples_per_step = W * H * spp
    total_steps = K if args.scaling == "strong" else world * K
    value = total_steps * samples_per_step / elapsed / 1e6

    result = None
    if rank == 0:
        # ---- algorithmic FLOPs per launch from the instrumented kernel (same samples as timed steps) --
        r.unbind()
        r.reset()
        r.reset_counters()
        cp = abi.render_params(max_depth=max_depth, flags=flags | abi.FLAG_COUNTERS, pixels_per_wave=args.pixels_per_wave)
        n_cnt = max(1, min(my_steps, 8))
        for s0, n in sched[Wm:Wm + n_cnt]:
            r.render(s0, n, cp)
        cnt = r.counters().as_dict()
        flops_per_launch = algorithmic_flops(cnt, scene.num_spheres, scene.num_planes, scene.num_triangles, W * H * n_cnt) / n_cnt
        rays_per_sample = cnt["rays"] / max(cnt["samples"], 1)
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 5), "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H, "spp_per_step": spp,
                       "max_depth": max_depth, "integrator": "path+nee_mis" if args.mis else "path",
                       "accumulate": "sum+fused_rgba8", "frames_in_flight": args.frames_in_flight, "parallelism": f"sample_split{world}" if world > 1 else "single",
                       "reduce": ("gloo_rehearsal_on_one_gpu" if rehearsal else "rccl_reduce_sum_once") if world > 1 else None},
            "mrays_per_s": round(value * rays_per_sample, 1), "rays_per_sample": round(rays_per_sample, 4),
            "roofline": {"bound": "valu", "kernel": "pt_path_kernel", "achieved": round(achieved, 4),
                         "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 5),
                         "traffic": traffic, "flops_per_launch": flops_per_launch,
                         "flops_per_sample": round(flops_per_launch / samples_per_step, 1),
                         "kernel_ms": round(kernel_ms, 5), "launch_overlap": args.frames_in_flight,
                         "kernel_ms_serial": round(serial_ms, 5),
                         "frac_serial": round(flops_per_launch / (serial_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 5)},
        }
        if world == 1 and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline(scene, cam, params, spp, args.cpu_seconds)
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
                      "oracle/libpt_oracle.so float build, gcc -O2 -fopenmp"}


if __name__ == "__main__":
    sys.exit(main())
