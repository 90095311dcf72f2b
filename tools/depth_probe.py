"""Serial 720p / 1 spp frame time against max_depth: how much of the one-launch-per-frame time is the tail of the
deepest paths (the mirror corridor runs to max_depth with no Russian roulette).
python tools/depth_probe.py [variant library name]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] != "main":
    os.environ["PT_AMD_LIB"] = os.path.join(ROOT, "path_tracer_amd", "lib", f"libpt_amd_{sys.argv[1]}.so")
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
sc, cam = scenes.cuda_default(1280, 720)
r.set_scene(sc); r.set_camera(cam); r.reset()
for depth in (4, 8, 12, 20, 30, 50):
    p = abi.render_params(max_depth=depth, fused_rgba8=1)
    best = 1e9
    for rep in range(4):
        for i in range(8): r.render(i, 1, p)
        r.sync(); t = time.perf_counter()
        for i in range(100): r.render(8 + i, 1, p)
        r.sync(); best = min(best, (time.perf_counter() - t) / 100)
    pc = abi.render_params(max_depth=depth, flags=abi.FLAG_COUNTERS)
    r.reset_counters(); r.render(0, 1, pc); c = r.counters().as_dict()
    print(f"max_depth {depth:2d}: {best * 1e6:7.1f} us per serial frame, {c['rays'] / c['samples']:.2f} rays per sample", flush=True)
r.close()
