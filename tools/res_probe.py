"""Steady-state rate of multi-frame launches (two chains) by frame size and samples per call.  With PT_AMD_CHUNK=0 calls of
several samples are not cut into frames of 1-4 samples (pt_api.cpp, wants_parts): a lane then stays on a pixel for all samples
of the call -- where 4K's +8 % over 720p in bench.py's legs come from.
    python tools/res_probe.py;  PT_AMD_CHUNK=0 python tools/res_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
for (w, h, spp, steps) in ((1280, 720, 1, 1024), (1280, 720, 2, 512), (1280, 720, 4, 512), (1280, 720, 16, 256), (1280, 720, 64, 64), (1920, 1080, 16, 128), (3840, 2160, 16, 48), (3840, 2160, 64, 16)):
    sc, cam = scenes.cuda_default(w, h)
    r.set_frames_in_flight(1); r.set_scene(sc); r.set_camera(cam); r.reset(); r.set_frames_in_flight(2)
    p = abi.render_params(fused_rgba8=1)
    for k in range(8): r.render(k * spp, spp, p)
    r.flush(); r.sync()
    best = 0
    for rep in range(2):
        l0 = r.path_launches(); t = time.time()
        for k in range(steps): r.render((8 + k) * spp, spp, p)
        r.flush(); r.sync(); dt = time.time() - t
        best = max(best, w * h * spp * steps / dt / 1e9)
    print(f"chunk={os.environ.get('PT_AMD_CHUNK')} {w}x{h} {spp:2d} spp/frame {steps:5d} frames {r.path_launches()-l0} launches: {best:6.2f} Gsamples/s", flush=True)
