"""A/B of library builds on the FP64 side (`direct` integrator, id pass): python tools/direct_ab.py main v1 ...
Each build in a child process, two interleaved rounds; prints Gsamples/s of BASELINE config 1 (cpu_default, 1280x720,
64 spp per step) and of cuda_default under the same integrator, and md5s of the buffers (a build that only skips work
must leave every bit where it was)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, hashlib
sys.path.insert(0, %r)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0); out = {}
p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
def timeit(name, sc, cam, spl, steps):
    r.set_scene(sc); r.set_camera(cam); r.reset()
    best = 0
    for rep in range(3):
        r.render(0, spl, p); r.sync(); t = time.time()
        for i in range(steps): r.render((1 + i) * spl, spl, p)
        r.sync(); dt = time.time() - t
        best = max(best, cam.width * cam.height * spl * steps / dt / 1e9)
    out[name] = round(best, 3)
timeit("cpu_default_720p_64spp", *scenes.cpu_default(1280, 720), 64, 10)
timeit("cuda_default_720p_16spp", *scenes.cuda_default(1280, 720), 16, 10)
for nm, (s_, c_) in (("cpu_default", scenes.cpu_default(640, 360)), ("cuda_default", scenes.cuda_default(320, 180)),
                     ("random149", scenes.builtin("random149", 160, 90))):
    r.set_scene(s_); r.set_camera(c_); r.reset()
    r.render(0, 8, p)
    out["md5_" + nm] = hashlib.md5(r.accum().tobytes()).hexdigest()[:8]
    ids, mats = r.primary_ids()
    out["ids_" + nm] = hashlib.md5(ids.tobytes() + mats.tobytes()).hexdigest()[:8]
print(json.dumps(out))
''' % ROOT
for rep in range(2):
    for v in sys.argv[1:]:
        env = dict(os.environ)
        if v != "main":
            env["PT_AMD_LIB"] = os.path.join(ROOT, "path_tracer_amd", "lib", f"libpt_amd_{v}.so")
        q = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=280)
        line = [l for l in q.stdout.splitlines() if l.startswith("{")]
        print(v, json.loads(line[-1]) if line else {"error": q.stderr[-600:]}, flush=True)
