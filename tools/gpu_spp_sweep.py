"""Msamples/s at 720p as a function of samples per launch, serial and with frames in flight."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1: os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
out = {}
for name, (w, h) in {"720p": (1280, 720), "1080p": (1920, 1080), "360p": (640, 360)}.items():
    sc, cam = scenes.cuda_default(w, h)
    r.set_scene(sc); r.set_camera(cam)
    for flight in (1,):
        r.set_frames_in_flight(flight)
        for spl in (16, 64, 256, 2048):
            total = 1024 if name != "4k" else 256
            launches = max(1, total // spl)
            r.reset(); r.render(0, spl, abi.render_params()); r.sync()
            best = 0.0
            for rep in range(2):
                t = time.time()
                for i in range(launches): r.render((1 + i) * spl, spl, abi.render_params())
                r.sync(); dt = time.time() - t
                best = max(best, w * h * spl * launches / dt / 1e6)
            out[f"{name}_flight{flight}_spl{spl}"] = round(best, 1)
            print(name, flight, spl, round(best, 1), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "spp_sweep.json"), "w"), indent=1)
