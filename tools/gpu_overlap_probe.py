"""Upper bound for more frame overlap than one context offers: several contexts, each with F frames in
flight, render alternate frames concurrently (python tools/gpu_overlap_probe.py; GPU_MAX_HW_QUEUES=16)."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("PROBE_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
def run(nctx, flight, w, h, spp, launches, blocks):
    if blocks: os.environ["PT_AMD_MAX_BLOCKS"] = str(blocks)
    else: os.environ.pop("PT_AMD_MAX_BLOCKS", None)
    rs = [Renderer(0) for _ in range(nctx)]
    sc, cam = scenes.cuda_default(w, h)
    for r in rs: r.use_own_stream(); r.set_scene(sc); r.set_camera(cam); r.reset(); r.set_frames_in_flight(flight)
    p = abi.render_params(fused_rgba8=1)
    best = 0
    for rep in range(3):
        for r in rs: r.render(0, spp, p)
        for r in rs: r.sync()
        t = time.time()
        for i in range(launches):
            rs[i % nctx].render((1 + i // nctx) * spp, spp, p)
        for r in rs: r.sync()
        dt = time.time() - t
        best = max(best, w * h * spp * launches / dt / 1e6)
    for r in rs: r.close()
    return round(best, 1)
for (w, h, spp, n) in ((1280, 720, 1, 480), (1280, 720, 16, 48)):
    for nctx, flight, blocks in ((1, 4, 0), (2, 3, 350), (2, 3, 300), (2, 4, 260), (2, 4, 220), (3, 2, 350), (3, 3, 230)):
        print(w, h, spp, "contexts", nctx, "flight", flight, "blocks", blocks, run(nctx, flight, w, h, spp, n, blocks), flush=True)
