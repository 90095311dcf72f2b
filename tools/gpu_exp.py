import sys, time, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
def timeit(name, sc, cam, params, spl, launches, warm=3):
    r.set_scene(sc); r.set_camera(cam); r.reset()
    best = 0
    for rep in range(3):
        for i in range(warm): r.render(i*spl, spl, params)
        r.sync(); t = time.time()
        for i in range(launches): r.render((warm+i)*spl, spl, params)
        r.sync(); dt = time.time()-t
        best = max(best, cam.width*cam.height*spl*launches/dt/1e6)
    print(f"{name:40s} {best:9.1f} Msamples/s  {cam.width*cam.height*spl/best/1e3:8.3f} ms/launch", flush=True)
sc, cam = scenes.cuda_default(1280, 720)
for md in (50, 16, 8, 4, 2, 1):
    timeit(f"720p_1spp_depth{md}", sc, cam, abi.render_params(fused_rgba8=1, max_depth=md), 1, 300)
for spl in (1, 2, 4, 8, 16, 64):
    timeit(f"720p_{spl}spp", sc, cam, abi.render_params(), spl, max(3, 300//spl))
sc4, cam4 = scenes.cuda_default(3840, 2160)
for spl in (1, 4, 16):
    timeit(f"4k_{spl}spp", sc4, cam4, abi.render_params(), spl, max(3, 40//spl))
for ppw in (64, 128, 256, 512):
    timeit(f"4k_16spp_ppw{ppw}", sc4, cam4, abi.render_params(pixels_per_wave=ppw), 16, 3)
