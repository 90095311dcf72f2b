"""720p, 1 spp per frame, fused display: Msamples/s against pixels per ticket and frames in flight."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
sc, cam = scenes.cuda_default(1280, 720)
r.set_scene(sc); r.set_camera(cam)
for flight in (8, 4):
    r.set_frames_in_flight(flight)
    for ppw in (0, 64, 96, 128, 192, 256, 320, 384, 512):
        p = abi.render_params(fused_rgba8=1, pixels_per_wave=ppw)
        r.reset()
        best = 0
        for rep in range(3):
            for i in range(8): r.render(i, 1, p)
            r.sync(); t = time.time()
            for i in range(400): r.render(8 + i, 1, p)
            r.sync(); best = max(best, 1280 * 720 * 400 / (time.time() - t) / 1e6)
        print("flight", flight, "ppw", ppw, round(best, 1), flush=True)
