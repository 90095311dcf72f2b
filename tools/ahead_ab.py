"""The one-frame-at-a-time loop with and without render-ahead (pt_set_render_ahead): 720p, 1 spp, fused display.
Three hosts: one that never waits (enqueue only), one that waits for every frame (pt_sync), one that waits and reads
the 3.7-MB display back (the reference's updateScreen).  python tools/ahead_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
CASES = (("cuda_default", (1280, 720), False), ("cuda_default", (1280, 720), True), ("glass_stress", (1280, 720), False), ("cuda_default", (1920, 1080), False))
if len(sys.argv) > 1 and sys.argv[1] == "quick": CASES = CASES[:1]      # (with PT_AMD_GRID_PCT / PT_AMD_FEW_MAX in the environment: tuning runs)
for name, size, mean in CASES:
    sc, cam = scenes.builtin(name, *size)
    p = abi.render_params(fused_rgba8=1, acc_mode=abi.ACC_RUNNING_MEAN if mean else abi.ACC_SUM, max_depth=32 if name == "glass_stress" else 50)
    for host in ("enqueue", "sync", "sync+readback"):
        row = []
        for rep in range(2):
            for on in (False, True):
                r.set_render_ahead(on); r.set_scene(sc); r.set_camera(cam); r.reset()
                for f in range(8): r.render(f, 1, p)
                r.sync(); n = 200; t = time.perf_counter()
                for f in range(8, 8 + n):
                    r.render(f, 1, p)
                    if host != "enqueue": r.sync()
                    if host == "sync+readback": r.readback(abi.RESOLVE_RGBA8_CUDA)
                r.sync(); dt = time.perf_counter() - t
                row.append(f"{'ahead' if on else 'serial'} {dt / n * 1e3:.4f} ms")
        print(name, size, "mean" if mean else "sum", host, " | ".join(row), flush=True)
print("stats (launched ahead, used):", r.render_ahead_stats())
