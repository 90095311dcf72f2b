import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
from tests.oracle_binding import load_oracle
o = load_oracle(); r = Renderer(0)
for name, (w, h) in [("cpu_default", (1280, 720)), ("cuda_default", (96, 64)), ("cuda_default", (1280, 720)), ("glass_stress", (320, 180))]:
    sc, cam = scenes.builtin(name, w, h)
    r.set_scene(sc); r.set_camera(cam)
    gi, gm = r.primary_ids(); oi, om = o.primary_ids(sc, cam)
    bad = np.argwhere(gi != oi)
    print(name, w, h, "id mismatches", len(bad), "mat mismatches", int((gm != om).sum()))
    for y, x in bad[:8]: print("   ", x, y, "gpu", gi[y, x], "oracle", oi[y, x])
    print("   hist gpu", np.bincount(gi.ravel()[gi.ravel() != abi.MISS], minlength=8)[:20].tolist())
    print("   hist ora", np.bincount(oi.ravel()[oi.ravel() != abi.MISS], minlength=8)[:20].tolist())
sc, cam = scenes.builtin("cuda_default", 96, 64)
r.set_scene(sc); r.set_camera(cam)
gi, gm = r.primary_ids(); oi, om = o.primary_ids(sc, cam)
bad = np.argwhere(gm != om)
for y, x in bad[:5]: print("mat mismatch at", x, y, "obj", gi[y, x], "gpu mat", gm[y, x], "oracle mat", om[y, x])
print("tri_mat", sc.arrays["tri_mat"], "tri_obj", sc.arrays["tri_obj"], "pl_mat", sc.arrays["pl_mat"])
