"""GPU render of cuda_default at the screenshot's resolution vs the reference's own images/cuda.png fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_cuda_screenshot_480x270.npz"))["rgb"].astype(np.float64) / 255.0
r = Renderer(0)
sc, cam = scenes.cuda_default(1920, 1080)
r.set_scene(sc); r.set_camera(cam)
for spp in (256, 2048, 8192):
    r.reset()
    for s0 in range(0, spp, 64): r.render(s0, 64, abi.render_params())
    acc = r.accum()
    disp = np.clip(acc[..., :3] / acc[..., 3:4], 0, 1).astype(np.float64)
    small = disp.reshape(270, 4, 480, 4, 3).mean(axis=(1, 3))
    d = np.abs(small - ref)
    print(spp, "MAE", d.mean().round(4), "RMSE", np.sqrt((d ** 2).mean()).round(4), "p99", np.percentile(d, 99).round(3),
          "means gpu", small.mean(axis=(0, 1)).round(4), "ref", ref.mean(axis=(0, 1)).round(4), flush=True)
# clamp before averaging is what a screenshot of clamped pixels holds; MIS render for comparison (same expectation)
r.reset()
for s0 in range(0, 2048, 64): r.render(s0, 64, abi.render_params(flags=abi.FLAG_NEE_MIS))
acc = r.accum(); small = np.clip(acc[..., :3] / acc[..., 3:4], 0, 1).astype(np.float64).reshape(270, 4, 480, 4, 3).mean(axis=(1, 3))
d = np.abs(small - ref); print("mis 2048 MAE", d.mean().round(4), "means", small.mean(axis=(0, 1)).round(4))
