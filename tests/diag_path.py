"""Diagnostic: path kernel vs float oracle, counters and per-depth images on a small frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import Renderer
from tests.oracle_binding import load_oracle
o = load_oracle(); r = Renderer(0)
FLAGS = int(os.environ.get("DIAG_FLAGS", "0"))
for name in sys.argv[1:] or ["cuda_default"]:
    sc, cam = scenes.builtin(name, 64, 36)
    r.set_scene(sc); r.set_camera(cam)
    gi, _ = r.primary_ids()
    for depth in (1, 2, 3, 50):
        p = abi.render_params(max_depth=depth, flags=abi.FLAG_COUNTERS | FLAGS)
        r.resize(64, 36) if hasattr(r, "resize") else None
        r.reset(); r.reset_counters()
        r.render(0, 4, p)
        g = r.accum(); gc = r.counters().as_dict()
        w, oc = o.render_path(sc, cam, 0, 4, p, precision=32, counters=True)
        oc = oc.as_dict()
        diff = np.abs(g[..., :3] - w[..., :3]).max(axis=-1)
        bad = diff > 1e-3 * (1 + np.abs(w[..., :3]).max(axis=-1))
        print(name, "depth", depth, "bad pixels", int(bad.sum()), "of", bad.size)
        print("   gpu", {k: gc[k] for k in ("samples", "rays", "hits", "sphere_roots", "bounces", "nee_samples", "shadow_rays")})
        print("   ora", {k: oc[k] for k in ("samples", "rays", "hits", "sphere_roots", "bounces", "nee_samples", "shadow_rays")})
        print("   counts min/max", g[..., 3].min(), g[..., 3].max(), " mean gpu", g[..., :3].mean(), "ora", w[..., :3].mean())
        if bad.any():
            ids, cnt = np.unique(gi[bad], return_counts=True)
            print("   primary ids of bad pixels", dict(zip(ids.tolist(), cnt.tolist())))
            ids, cnt = np.unique(gi, return_counts=True)
            print("   primary ids of all pixels", dict(zip(ids.tolist(), cnt.tolist())))
            for y, x in np.argwhere(bad)[:4]:
                print("     ", x, y, "id", gi[y, x], "gpu", g[y, x], "oracle", w[y, x])
