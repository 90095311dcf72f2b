"""A/B timing of kernel-variant builds: python tools/gpu_ab.py a0 a1 ...  (each in a child process;
interleaved rounds would be better for <3% deltas, these are coarse variant screens)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
import os
r = Renderer(0); out = {}
if os.environ.get("PT_FLIGHT"): r.set_frames_in_flight(int(os.environ["PT_FLIGHT"]))
def timeit(name, sc, cam, params, spl, launches, warm=2):
    r.set_scene(sc); r.set_camera(cam); r.reset()
    best = 0
    for rep in range(3):
        for i in range(warm): r.render(i*spl, spl, params)
        r.sync(); t = time.time()
        for i in range(launches): r.render((warm+i)*spl, spl, params)
        r.sync(); dt = time.time()-t
        best = max(best, cam.width*cam.height*spl*launches/dt/1e6)
    out[name] = round(best, 1)
sc, cam = scenes.cuda_default(1280, 720)
timeit("720p_1spp", sc, cam, abi.render_params(fused_rgba8=1), 1, 300)
timeit("720p_1spp_ppw256", sc, cam, abi.render_params(fused_rgba8=1, pixels_per_wave=256), 1, 300)
timeit("720p_16spp", sc, cam, abi.render_params(), 16, 20)
timeit("720p_16spp_mis", sc, cam, abi.render_params(flags=abi.FLAG_NEE_MIS), 16, 20)
sc4, cam4 = scenes.cuda_default(3840, 2160)
timeit("4k_16spp_ppw256", sc4, cam4, abi.render_params(pixels_per_wave=256), 16, 3, warm=1)
g, gc = scenes.glass_stress(1280, 720)
timeit("glass_16spp", g, gc, abi.render_params(max_depth=32), 16, 10)
rs, rc = scenes.random4096(1280, 720)
timeit("rand4096_720p_1spp", rs, rc, abi.render_params(max_depth=16), 1, 3, warm=1)
# bit-identity check between builds: md5 of small accumulation buffers (same samples, same launch shape)
import hashlib
for nm, (s_, c_), dep in (("md5_cornell", scenes.cuda_default(160, 90), 50), ("md5_glass", scenes.glass_stress(96, 54), 32), ("md5_rand4096", scenes.random4096(64, 36), 16)):
    for fl in (0, abi.FLAG_NEE_MIS):
        r.set_scene(s_); r.set_camera(c_); r.reset()
        r.render(0, 4, abi.render_params(max_depth=dep, flags=fl))
        out[nm + ("_mis" if fl else "")] = hashlib.md5(r.accum().tobytes()).hexdigest()[:8]
print(json.dumps(out))
''' % ROOT
res = {}
for v in sys.argv[1:]:
    env = dict(os.environ)
    for kv in v.split(","):   # e.g. flight=3,blocks=768,few=2 or a variant library name
        if kv.startswith("blocks="): env["PT_AMD_MAX_BLOCKS"] = kv.split("=")[1]
        elif kv.startswith("flight="): env["PT_FLIGHT"] = kv.split("=")[1]
        elif kv.startswith("few="): env["PT_AMD_FEW_MAX"] = kv.split("=")[1]
        elif kv.startswith("regen="): env["PT_AMD_REGEN_MIN"] = kv.split("=")[1]
        elif kv.startswith("idle="): env["PT_AMD_IDLE_MIN"] = kv.split("=")[1]
        elif kv.startswith("prefetch="): env["PT_AMD_PREFETCH"] = kv.split("=")[1]
        elif kv.startswith("queues="): env["GPU_MAX_HW_QUEUES"] = kv.split("=")[1]
        elif kv != "main": env["PT_AMD_LIB"] = os.path.join(ROOT, "path_tracer_amd", "lib", f"libpt_amd_{kv}.so")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=280)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    res[v] = json.loads(line[-1]) if line else {"error": p.stderr[-400:]}
    print(v, res[v], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ab.json"), "w"), indent=1)
