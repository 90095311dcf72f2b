"""How tests/cpu_screenshot.py's POSE was found: the camera of the reference's grab images/cpu.png is not recorded, so
its five numbers -- position, yaw, pitch of cpu_pathtracer/src/main.cpp:157-193 -- are fitted to the COLOUR CLASSES of
the grab's lit pixels: the number of lit pixels that lie on an object which cannot show their colour (primary ids of the
oracle's FP64 id pass at the candidate pose) is minimised, first by differential evolution over the whole box, then by
Nelder-Mead.  No pixel VALUE enters the fit; tests/cpu_screenshot.py compares the values afterwards.
The objective is flat at its bottom (edge pixels): this script, from scratch, ends at (0.0429, -1.0660, 2.4293, -1.56946,
-1.04409) with 77 of 93 755 lit pixels misplaced; POSE in tests/cpu_screenshot.py, from an earlier run started elsewhere,
is (0.0385, -1.0692, 2.4273, -1.56856, -1.04287) with 73 -- the same camera to 0.005 units and 0.001 rad.
Needs the fixture and the oracle only (not the reference tree); about eight minutes on 8 cores:
    python tests/golden/fit_cpu_screenshot_pose.py
"""
import os
import sys
import time

import numpy as np
from scipy.optimize import differential_evolution, minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes          # noqa: E402
from tests import cpu_screenshot as cs           # noqa: E402
from tests.oracle_binding import load_oracle     # noqa: E402

oracle = load_oracle()
scene, _ = scenes.cpu_default(cs.W, cs.H)
shot = cs.load()
hh, ww = shot.shape[:2]
cls = cs.classes(shot)
# colour class -> objects that can show it (tests/cpu_screenshot.py, check (2))
ALLOWED = {1: (0, 3, 4, 5, 6, 7), 2: (1, 6), 3: (2,), 4: (5,), 5: (6,), 6: (2,)}
MASKS = {k: cls == k for k in ALLOWED}


def camera(p):
    px, py, pz, yaw, pitch = p
    f = (np.cos(pitch) * np.cos(yaw), np.sin(pitch), np.cos(pitch) * np.sin(yaw))
    return abi.make_camera((px, py, pz), (px + f[0], py + f[1], pz + f[2]), (0, 1, 0), 90.0, cs.W, cs.H, abi.CAMERA_CPU, normalize_up=0)


def misplaced(p):
    """Lit pixels of the grab that lie on an object which cannot show their colour."""
    if abs(p[0]) > 1.95 or abs(p[1]) > 1.95 or p[2] < -1.9 or p[2] > 8.0 or abs(p[4]) > np.pi / 2 - 0.1:   # inside the box; main.cpp:158-159
        return 1e9
    ids = oracle.primary_ids(scene, camera(p))[0][:hh, :ww]
    return float(sum((~np.isin(ids[m], ALLOWED[k])).sum() for k, m in MASKS.items()))


if __name__ == "__main__":
    t = time.time()
    # the room has no front wall (main.cpp:63-131): z may exceed 2
    box = [(-1.9, 1.9), (-1.9, 1.9), (-1.9, 6.0), (-np.pi, np.pi), (-np.pi / 2 + 0.1, np.pi / 2 - 0.1)]
    r = differential_evolution(misplaced, box, seed=1, popsize=30, maxiter=60, tol=1e-8, polish=False)
    print("differential evolution:", r.x.tolist(), r.fun, r.nfev, f"{time.time() - t:.0f} s", flush=True)
    p = r.x
    for rnd in range(4):
        r = minimize(misplaced, p, method="Nelder-Mead", options=dict(xatol=1e-7, fatol=0.5, maxfev=800,
                     initial_simplex=[p + dv for dv in np.vstack([np.zeros(5), np.eye(5) * (0.1 / (rnd + 1))])]))
        p = r.x
        print("Nelder-Mead:", repr(p.tolist()), r.fun, f"{time.time() - t:.0f} s", flush=True)
    print("POSE =", tuple(float(v) for v in p), "misplaced", misplaced(p), "of", int(sum(m.sum() for m in MASKS.values())))
