"""Random frame sizes, call sizes and frames in flight against single serial launches (PT_AMD_CHUNK=0): every pixel's
sample count exact, sums equal up to float summation order, display equal to the resolve pass -- and against the same
calls made with one frame in flight: the same image bit for bit, whatever the number of chains."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
os.environ["PT_AMD_CHUNK"] = "0"
ref = Renderer(0)
ref.set_render_ahead(False)   # the reference side: strictly one launch per call
del os.environ["PT_AMD_CHUNK"]
r = Renderer(0)
one = Renderer(0)   # the library's defaults with one frame in flight
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    w, h = int(rng.integers(1, 400)), int(rng.integers(1, 260))
    name = ["cuda_default", "glass_stress", "cpu_default"][int(rng.integers(0, 3))]
    sc, cam = scenes.builtin(name, w, h); cam.model = abi.CAMERA_CUDA
    flags = abi.FLAG_NEE_MIS if rng.integers(0, 3) == 0 else 0
    flight = int(rng.integers(1, 9))
    calls = [int(c) for c in rng.choice([1, 2, 5, 11, 12, 13, 31, 48, 49, 64, 100, 150], size=int(rng.integers(1, 5)))]
    fused = int(rng.integers(0, 2))
    p = abi.render_params(flags=flags, max_depth=int(rng.choice([1, 4, 50])), fused_rgba8=fused)
    q = abi.render_params(flags=flags, max_depth=p.max_depth)
    ref.set_frames_in_flight(1); ref.set_scene(sc); ref.set_camera(cam); ref.reset()
    r.set_frames_in_flight(flight); r.set_scene(sc); r.set_camera(cam); r.reset()
    one.set_scene(sc); one.set_camera(cam); one.reset()
    s0 = int(rng.integers(0, 1000)); s = s0
    for c in calls:
        ref.render(s, c, q); r.render(s, c, p); one.render(s, c, p); s += c
    want, got = ref.accum(), r.accum()
    ok = np.array_equal(got[..., 3], want[..., 3]) and np.all(got[..., 3] == s - s0) and np.allclose(got[..., :3], want[..., :3], rtol=2e-5, atol=2e-5)
    ok = ok and np.array_equal(got, one.accum())
    if fused:
        ok = ok and np.array_equal(r.readback(abi.RESOLVE_RGBA8_CUDA), one.readback(abi.RESOLVE_RGBA8_CUDA))
        ok = ok and np.array_equal(r.readback(abi.RESOLVE_RGBA8_CUDA), r.resolve_to_host(abi.RESOLVE_RGBA8_CUDA))
    if not ok:
        bad += 1
        print("MISMATCH", it, name, w, h, flags, flight, calls, fused, p.max_depth, float(np.abs(got - want).max()), flush=True)
print("fuzz done, mismatches:", bad, flush=True)
sys.exit(1 if bad else 0)
