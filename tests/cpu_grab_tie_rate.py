"""SURVEY D6, measured on the reference's own grab (tests/cpu_screenshot.py): a `direct` light sample counts only if the
shadow ray's hit on the sampled light sphere itself comes out NOT nearer than the sampled point -- two roundings of one
number.  Per object of the grab: the share of light samples that exact geometry lets through (receiver faces the point,
the point lies on the near side of the light, below the ceiling plane, no other sphere in between; Monte Carlo at the
fitted camera), the share of pixels the grab shows lit, and their ratio = how often the reference's build resolved the
tie to `visible` -- and the same two columns for the IEEE restatement (the oracle; the HIP kernel's frames are the same bits).
    python tests/cpu_grab_tie_rate.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import cpu_screenshot as cs

shot = cs.load(); hh, ww = shot.shape[:2]
lit = cs.classes(shot) != 0
PLANES = [((0, 0, -2), (0, 0, 1), 0), ((-2, 0, 0), (1, 0, 0), 1), ((2, 0, 0), (-1, 0, 0), 2), ((0, -2, 0), (0, 1, 0), 3), ((0, 2, 0), (0, -1, 0), 4)]   # main.cpp:63-131
SPHERES = [((-1.0, -1.2, -1.0), 0.8, 5), ((1.0, -1.6, -1.0), 0.4, 6), ((0, 1.9, 0), 0.2, 7)]
NAMES = {0: "back wall", 1: "red wall", 2: "green wall", 3: "floor", 5: "white sphere", 6: "small sphere"}
unit = lambda v: v / np.linalg.norm(v, axis=-1, keepdims=True)

px, py, pz, yaw, pitch = cs.POSE                                   # Camera::initialize / getRayForPixel, Camera.h:65-85,124-142
o = np.array([px, py, pz]); f = np.array([np.cos(pitch) * np.cos(yaw), np.sin(pitch), np.cos(pitch) * np.sin(yaw)])
w = unit(-f); u = unit(np.cross([0, 1.0, 0], w)); v = np.cross(w, u)
hor, ver = u * (2 * 16 / 9) * 10.0, v * 2.0 * 10.0
ys, xs = np.mgrid[0:hh, 0:ww].astype(np.float64)
loc = (o - hor / 2 + ver / 2 - w * 10.0) + hor * ((xs + 0.5) / (cs.W - 1))[..., None] + ver * (1.0 - (ys + 0.5) / (cs.H - 1))[..., None]
d = unit(loc - o)

def sphere_roots(org, dr, c, r):
    oc = org - np.array(c); hb = (dr * oc).sum(-1); disc = hb * hb - ((oc * oc).sum(-1) - r * r)
    sq = np.sqrt(np.maximum(disc, 0)); return disc > 0, -hb - sq, -hb + sq

tb = np.full((hh, ww), np.inf); ib = np.full((hh, ww), -1)
for p0, n, i in PLANES:
    den = d @ np.array(n, float); t = ((np.array(p0, float) - o) @ np.array(n, float)) / np.where(den == 0, 1e-30, den)
    t = np.where((t > 1e-8) & (np.abs(den) > 1e-12), t, np.inf); m = t < tb; tb = np.where(m, t, tb); ib = np.where(m, i, ib)
for c, r, i in SPHERES:
    ok, t1, t2 = sphere_roots(o, d, c, r); t = np.where(t1 > 1e-8, t1, t2); t = np.where(ok & (t > 1e-8), t, np.inf)
    m = t < tb; tb = np.where(m, t, tb); ib = np.where(m, i, ib)
hit = o + d * tb[..., None]; nrm = np.zeros_like(hit)
for p0, n, i in PLANES: nrm[ib == i] = n
for c, r, i in SPHERES: nrm[ib == i] = (hit[ib == i] - np.array(c)) / r
rng = np.random.default_rng(0); K = 24; geo = np.zeros((hh, ww)); C, R = np.array(SPHERES[2][0], float), SPHERES[2][1]
for _ in range(K):
    th = 2 * np.pi * rng.random((hh, ww)); ph = np.arccos(2 * rng.random((hh, ww)) - 1)     # Sphere::getRandomPoint
    dl = np.stack([np.cos(th) * np.sin(ph), np.sin(th) * np.sin(ph), np.cos(ph)], -1); lp = C + R * dl
    tol = lp - hit; dist = np.linalg.norm(tol, axis=-1); ld = tol / dist[..., None]
    ok = ((nrm * ld).sum(-1) > 0) & ((ld * dl).sum(-1) < 0) & (lp[..., 1] < 2.0)
    for c, r, i in SPHERES[:2]:
        has, t1, t2 = sphere_roots(hit, ld, c, r)
        ok &= ~(has & (((t1 > 1e-6) & (t1 < dist)) | ((t2 > 1e-6) & (t2 < dist) & (ib != i))))
    geo += ok
geo /= K
from path_tracer_amd import abi                      # the restatement's own frames at the same camera
from tests.oracle_binding import load_oracle
oracle = load_oracle(); sc, cam = cs.scene_and_camera(); prm = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
ours = np.zeros((hh, ww)); N = 16
for s_ in range(N):
    wds = oracle.resolve(oracle.render_direct(sc, cam, s_, 1, prm), abi.ACC_SUM, abi.RESOLVE_RGBA8_CPU).reshape(cs.H, cs.W)
    ours += cs.classes(cs.unpack(wds)[:hh, :ww]) != 0
ours /= N
print(f"{'object':14s} {'pixels':>8s} {'let through by geometry':>24s} {'lit in the grab':>16s} {'tie -> visible':>15s} {'lit in ours':>12s} {'tie -> visible':>15s}")
for i, name in NAMES.items():
    m = (ib == i) & (geo > 0.2)          # receivers in the open (not in shadow, not facing away)
    if m.sum() >= 500:
        print(f"{name:14s} {int(m.sum()):8d} {geo[m].mean():24.4f} {lit[m].mean():16.4f} {lit[m].mean() / geo[m].mean():15.3f} {ours[m].mean():12.4f} {ours[m].mean() / geo[m].mean():15.3f}")
