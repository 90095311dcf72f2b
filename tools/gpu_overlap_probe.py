"""Upper bound for frame overlap: two contexts on their own streams render the same frames concurrently."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
def run(nctx, w, h, spp, launches):
    rs = [Renderer(0) for _ in range(nctx)]
    sc, cam = scenes.cuda_default(w, h)
    for r in rs: r.use_own_stream(); r.set_scene(sc); r.set_camera(cam); r.reset()
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
    return round(best, 1)
for (w, h, spp, n) in ((1280, 720, 1, 400), (1280, 720, 16, 24), (3840, 2160, 1, 60)):
    print(w, h, spp, {k: run(k, w, h, spp, n) for k in (1, 2, 3)}, flush=True)
