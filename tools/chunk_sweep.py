"""Samples per part when one pt_render call for many samples is cut into frames of a multi-frame launch
(PT_AMD_CHUNK, pt_api.cpp part_size): python tools/chunk_sweep.py [chunk ...]   (0 = never cut)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
out = {}
r = Renderer(0)
for name, size in (("720p", (1280, 720)), ("1080p", (1920, 1080)), ("4k", (3840, 2160))):
    sc, cam = scenes.cuda_default(*size)
    for flight in (1, 3):
        for spp in ((4, 8, 16, 64, 256) if name != "4k" else (4, 16, 64)):
            r.set_frames_in_flight(1); r.set_scene(sc); r.set_camera(cam); r.reset(); r.set_frames_in_flight(flight)
            p = abi.render_params(fused_rgba8=1)
            best = 0.0
            for rep in range(4):
                r.sync(); t = time.perf_counter()
                r.render(rep * spp, spp, p)
                r.sync(); dt = time.perf_counter() - t
                if rep: best = max(best, cam.width * cam.height * spp / dt / 1e6)
            out["%%s_f%%d_%%dspp" %% (name, flight, spp)] = round(best)
r.close()
print(json.dumps(out))
''' % ROOT
for c in (sys.argv[1:] or ["default", "0", "4", "8", "16", "32"]):
    env = dict(os.environ)
    if c != "default": env["PT_AMD_CHUNK"] = c
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print("chunk", c, json.loads(line[-1]) if line else p.stderr[-300:], flush=True)
