"""Sweeps of the launch heuristics' knobs (PT_AMD_FEW_MAX, PT_AMD_REGEN_MIN, PT_AMD_GRID_PCT, PT_AMD_IDLE_MIN) on scenes of 19 /
150 / 1000 / 4097 primitives, serial one-frame launches and frames in flight: the tables the rules in pt_api.cpp
(launch_now) were fitted on, and what tests/test_parity_gpu.py::test_launch_heuristics_are_near_the_best_of_a_sweep guards.
    python tools/heuristics_sweep.py [knob=v1,v2,...] ...     (default: few_max over 0..48)
Each setting runs in a child process (the knobs are read at pt_create)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
out = {}
r = Renderer(0)
for prims, count, size, depth in ((19, 0, (1280, 720), 50), (150, 149, (1280, 720), 16), (1000, 999, (640, 360), 16), (4097, 4096, (640, 360), 16)):
    sc, cam = scenes.cuda_default(*size) if prims == 19 else scenes.random4096(*size, count=count)
    for mode, flight, frames in (("serial", 1, 8 if prims <= 150 else 3), ("in_flight", 3, 32 if prims <= 150 else 8)):
        r.set_frames_in_flight(1); r.set_scene(sc); r.set_camera(cam); r.reset(); r.set_frames_in_flight(flight)
        p = abi.render_params(max_depth=depth, fused_rgba8=1)
        best = 0.0
        for rep in range(7):
            r.sync(); t = time.perf_counter()
            for f in range(frames): r.render(rep * frames + f, 1, p)
            r.sync(); dt = time.perf_counter() - t
            if rep: best = max(best, cam.width * cam.height * frames / dt / 1e6)
        out["%%d_%%s" %% (prims, mode)] = round(best, 1)
r.close()
print(json.dumps(out))
''' % ROOT
ENV = {"few_max": "PT_AMD_FEW_MAX", "regen_min": "PT_AMD_REGEN_MIN", "grid": "PT_AMD_GRID_PCT", "idle_min": "PT_AMD_IDLE_MIN", "prefetch": "PT_AMD_PREFETCH"}
sweeps = sys.argv[1:] or ["few_max=0,3,6,8,12,16,24,32,48"]
settings = [("default", {})]
for sw in sweeps:
    knob, vals = sw.split("=")
    settings += [(f"{knob}={v}", {ENV[knob]: v}) for v in vals.split(",")]
res = {}
for name, env in settings:
    p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    res[name] = json.loads(line[-1]) if line else {"error": p.stderr[-300:]}
    print(name, res[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "heuristics_sweep.json"), "w"), indent=1)
