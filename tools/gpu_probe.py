"""First-light / probe script for the GPU box: smoke, VALU ceiling, a few timings."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
import __graft_entry__ as g

out = {}
g.smoke()
r = Renderer(0)
for packed in (False, True):
    for it in (1 << 14, 1 << 16):
        out[f"valu_peak_packed{int(packed)}_it{it}"] = r.valu_peak(packed, it)
print(json.dumps(out, indent=1), flush=True)

def timeit(name, scene, cam, params, spp_per_launch, launches, warm=3):
    r.set_scene(scene); r.set_camera(cam); r.reset()
    for i in range(warm): r.render(i * spp_per_launch, spp_per_launch, params)
    r.sync(); t = time.time()
    for i in range(launches): r.render((warm + i) * spp_per_launch, spp_per_launch, params)
    r.sync(); dt = time.time() - t
    ms = cam.width * cam.height * spp_per_launch * launches / dt / 1e6
    out[name] = dict(msamples_per_s=ms, ms_per_launch=dt / launches * 1e3)
    print(name, out[name], flush=True)

sc, cam = scenes.cuda_default(1280, 720)
for ppw in (0, 64, 128, 256, 512, 1024):
    timeit(f"720p_1spp_ppw{ppw}", sc, cam, abi.render_params(pixels_per_wave=ppw), 1, 200)
timeit("720p_16spp", sc, cam, abi.render_params(), 16, 20)
timeit("720p_16spp_mis", sc, cam, abi.render_params(flags=abi.FLAG_NEE_MIS), 16, 20)
sc4, cam4 = scenes.cuda_default(3840, 2160)
for ppw in (0, 256, 1024, 4096):
    timeit(f"4k_16spp_ppw{ppw}", sc4, cam4, abi.render_params(pixels_per_wave=ppw), 16, 4, warm=1)
r.set_scene(sc); r.set_camera(cam); r.reset(); r.reset_counters()
r.render(0, 4, abi.render_params(flags=abi.FLAG_COUNTERS)); out["counters_720p_4spp"] = r.counters().as_dict()
print(out["counters_720p_4spp"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)
