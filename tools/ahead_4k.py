"""Render-ahead on frames of more than 2^22 pixels (calls of any sample count are one launch there): python tools/ahead_4k.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
for size, spp in (((3840, 2160), 16), ((3840, 2160), 1), ((3840, 2160), 4), ((2560, 1440), 1)):
    sc, cam = scenes.cuda_default(*size)
    p = abi.render_params(fused_rgba8=1)
    row = []
    for rep in range(2):
        for on in (False, True):
            r.set_render_ahead(on); r.set_scene(sc); r.set_camera(cam); r.reset()
            for f in range(4): r.render(f * spp, spp, p)
            r.sync(); n = 24; t = time.perf_counter()
            for f in range(4, 4 + n): r.render(f * spp, spp, p)
            r.sync(); dt = time.perf_counter() - t
            row.append(f"{'ahead' if on else 'serial'} {dt / n * 1e3:.3f} ms ({size[0] * size[1] * spp * n / dt / 1e9:.2f} G)")
    print(size, spp, "spp per call:", " | ".join(row), flush=True)
print(r.render_ahead_stats())
