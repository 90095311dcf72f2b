"""Host side of the frame loop: time to enqueue K frames against the time the GPU needs for them."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
r = Renderer(0)
sc, cam = scenes.cuda_default(1280, 720)
r.set_scene(sc); r.set_camera(cam); r.reset()
p = abi.render_params(fused_rgba8=1)
for flight in (1, 8):
    r.set_frames_in_flight(flight)
    for i in range(64): r.render(i, 1, p)
    r.sync()
    K = 2048
    t0 = time.perf_counter()
    for i in range(K): r.render(64 + i, 1, p)
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print(f"flight {flight}: enqueue {1e6 * (t1 - t0) / K:.1f} us/frame, total {1e6 * (t2 - t0) / K:.1f} us/frame", flush=True)
r.close()
