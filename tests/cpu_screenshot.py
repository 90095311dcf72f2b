"""The reference's own screenshot of the CPU app (images/cpu.png) as a pin for the `direct` path.

The grab was taken while the camera was being moved: every pixel shows ONE sample of renderRegion's estimator
(cpu_pathtracer/src/main.cpp:23-46), averaged over one frame, gamma 2.2, truncated to 8 bits (main.cpp:229-236,
WindowX11.h:134-136).  A pixel is therefore either unlit -- ambient 0.1 alone, byte 89 -- or holds one light sample,
`ambient + colour * 50 * cos / d^2`.  Which samples come out visible is decided by floating-point rounding in the
reference's -ffast-math build and cannot be restated; WHAT a pixel shows when it is lit, and WHERE each
object is on the screen, can.  The fixture is the client area of the grab (tests/golden/reference_cpu_screenshot_711x1270.npz,
made by make_reference_cpu_screenshot_fixture.py: data only).

The camera of the grab is not recorded.  `POSE` was fitted (position, yaw, pitch of main.cpp:157-193; five numbers) to the
COLOUR CLASSES of the lit pixels alone -- which wall or sphere a pixel belongs to -- and never saw a pixel value; the values
the light term gives at that pose are then compared with the screenshot's as an independent check.

Used by tests/test_oracle_pins.py (the oracle) and tests/test_parity_gpu.py (the HIP kernels): `frames` and `ids` come
from whichever renderer is under test.
"""
from __future__ import annotations

import os

import numpy as np

from path_tracer_amd import abi, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "reference_cpu_screenshot_711x1270.npz")
W, H = 1280, 720
# position, yaw, pitch (main.cpp:157-193) -- fitted to the colour classes of the lit pixels, see the module text
POSE = (0.0384945931522387, -1.0692219592452927, 2.4273277949838663, -1.5685574125539894, -1.042869467278662)
UNLIT = (89, 89, 89)
# object ids of scenes.cpu_default(): walls 0 back, 1 left (red), 2 right (green), 3 floor, 4 ceiling; 5 white sphere,
# 6 the (0.7, 0.3, 0.3) sphere, 7 the light


def scene_and_camera(dx=0.0):
    sc, _ = scenes.cpu_default(W, H)
    px, py, pz, yaw, pitch = POSE
    px += dx
    f = (np.cos(pitch) * np.cos(yaw), np.sin(pitch), np.cos(pitch) * np.sin(yaw))      # main.cpp:162-166
    cam = abi.make_camera((px, py, pz), (px + f[0], py + f[1], pz + f[2]), (0, 1, 0), 90.0, W, H, abi.CAMERA_CPU, normalize_up=0)
    return sc, cam


def load():
    """The client area of the grab as [711, 1270, 3] ints."""
    fx = np.load(FIXTURE)
    return fx["palette"][fx["index"]].astype(np.int64)


def unpack(words):
    """0x00RRGGBB words (PT_RESOLVE_RGBA8_CPU) -> [..., 3] ints."""
    w = words.astype(np.int64)
    return np.stack([(w >> 16) & 255, (w >> 8) & 255, w & 255], -1)


def classes(rgb):
    """Per pixel: 0 unlit, 1 white, 2 (255, g, g), 3 (r, 255, r), 4 grey below 255, 5 (R, g, g) with g < R < 255,
    6 (r, G, r) with r < G < 255, 7 anything else."""
    R, G, B = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    out = np.full(R.shape, 7, np.uint8)
    out[(R == G) & (G == B) & (R < 255)] = 4
    out[(R == 255) & (G == 255) & (B == 255)] = 1
    out[(R == 255) & (G == B) & (G < 255)] = 2
    out[(G == 255) & (R == B) & (R < 255)] = 3
    out[(R < 255) & (G == B) & (G < R)] = 5
    out[(G < 255) & (R == B) & (R < G)] = 6
    out[(R == UNLIT[0]) & (G == UNLIT[1]) & (B == UNLIT[2])] = 0
    return out


def check(frames, ids, record=None):
    """`frames`: list of [720, 1280] word images (PT_RESOLVE_RGBA8_CPU), one sample per pixel each, rendered at POSE;
    `ids`: [720, 1280] primary object ids at POSE.  Returns the measured figures (also asserted here)."""
    shot = load()
    hh, ww = shot.shape[:2]
    cls = classes(shot)
    ids = ids[:hh, :ww].astype(np.int64)
    out = {}

    # (1) what the grab holds: unlit pixels at byte 89, and nothing outside the families a light sample can produce
    assert tuple(shot.reshape(-1, 3)[np.bincount(np.load(FIXTURE)["index"].ravel()).argmax()]) == UNLIT
    assert (cls == 7).sum() == 0
    out["lit_fraction_of_the_screenshot"] = float((cls != 0).mean())

    # (2) every lit pixel of the grab lies on an object that can show its colour class
    allowed = {1: (0, 3, 4, 5, 6, 7), 2: (1, 6), 3: (2,), 4: (5,), 5: (6,), 6: (2,)}   # (the small sphere saturates to (255, g, g) and to white)
    lit = 0; wrong = 0
    for k, objs in allowed.items():
        m = cls == k
        lit += int(m.sum()); wrong += int((~np.isin(ids[m], objs)).sum())
    out["lit_pixels"] = lit
    out["lit_pixels_on_another_object"] = wrong
    assert wrong <= 0.002 * lit, f"{wrong} of {lit} lit pixels of the grab lie on another object"        # measured 73 of 93 755: pixel edges (the grab's rays are jittered, the id pass is not)

    # (3) the renderer's own frames: unlit pixels are byte 89 too, and its colours fall into the same families
    rgbs = [unpack(f)[:hh, :ww] for f in frames]
    mine = [classes(r) for r in rgbs]
    for c, r in zip(mine, rgbs):
        assert (c == 7).sum() == 0
        assert (c == 0).any() and tuple(r[c == 0][0]) == UNLIT
    # (4) the light term, pixel by pixel: where the grab shows a lit pixel of a coloured wall, the mean of the renderer's
    #     lit samples of that pixel (the channel that is not saturated: 0.1 + 0.2 * 50 cos / d^2) against the grab's one
    #     sample.  The sampled point moves over a light of radius 0.2, so one sample scatters by a few levels; the MEAN
    #     difference over ~17 000 and ~21 000 pixels is the test.
    for name, k, ch in (("red_wall", 2, 1), ("green_wall", 3, 0)):
        s = np.zeros((hh, ww)); n = np.zeros((hh, ww))
        for c, r in zip(mine, rgbs):
            m = c == k
            s[m] += r[..., ch][m]; n[m] += 1
        m = (cls == k) & (n > 0)
        d = s[m] / n[m] - shot[..., ch][m]
        out[f"{name}/pixels_compared"] = int(m.sum())
        out[f"{name}/mean_difference_levels"] = float(d.mean())
        out[f"{name}/rms_difference_levels"] = float(np.sqrt((d ** 2).mean()))
        assert m.sum() > 0.7 * (cls == k).sum(), name
        assert abs(d.mean()) < 0.4, (name, d.mean())          # measured 0.02 / 0.06 of ~200 levels
        assert np.sqrt((d ** 2).mean()) < 7.0, name            # measured 4.6 / 4.7 (one sample against the mean)
    # (5) the pixel formula, free of the pose: on the (0.7, 0.3, 0.3) sphere and at the far end of the green wall the
    #     bright channel is not saturated, so a pixel shows two different functions of ONE light term L --
    #     byte(255 (0.1 + 0.7 L)^(1/2.2)) and byte(255 (0.1 + 0.3 L)^(1/2.2)).  For every value of the bright byte that both
    #     the grab and the renderer show, the dim bytes must agree (a bright byte spans an interval of L that crosses at
    #     most one step of the dim byte: the union of both sides' dim bytes is one value or two neighbours).
    for name, k, bright, dim in (("sphere_0.7_0.3", 5, 0, 1), ("green_wall_unsaturated", 6, 1, 0)):
        theirs = {}
        for b, dm in zip(shot[..., bright][cls == k], shot[..., dim][cls == k]):
            theirs.setdefault(int(b), set()).add(int(dm))
        ours = {}
        for c, r in zip(mine, rgbs):
            m = c == k
            for b, dm in zip(r[..., bright][m], r[..., dim][m]):
                ours.setdefault(int(b), set()).add(int(dm))
        common = sorted(set(theirs) & set(ours))
        bad = [b for b in common if max(theirs[b] | ours[b]) - min(theirs[b] | ours[b]) > 1]
        meet = sum(1 for b in common if theirs[b] & ours[b])
        out[f"{name}/bright_bytes_compared"] = len(common)
        out[f"{name}/bright_bytes_with_a_common_dim_byte"] = meet
        assert len(common) >= (60 if k == 5 else 5), (name, len(common))
        assert not bad, (name, bad[:5])
        assert meet >= 0.9 * len(common), (name, meet, len(common))
    # (6) shadows and unlit sides: a 16x16 block of the grab without one lit pixel, surrounded by such blocks (chance at the
    #     grab's 11 % lit rate: 1e-13 a block) is never lit by the renderer either
    bs = 16
    by, bx = hh // bs, ww // bs
    dark = (cls[:by * bs, :bx * bs] == 0).reshape(by, bs, bx, bs).all(axis=(1, 3))
    inner = dark.copy()
    inner[1:, :] &= dark[:-1, :]; inner[:-1, :] &= dark[1:, :]; inner[:, 1:] &= dark[:, :-1]; inner[:, :-1] &= dark[:, 1:]
    inner[1:, 1:] &= dark[:-1, :-1]; inner[:-1, :-1] &= dark[1:, 1:]; inner[1:, :-1] &= dark[:-1, 1:]; inner[:-1, 1:] &= dark[1:, :-1]
    anylit = np.zeros((hh, ww), bool)
    for c in mine:
        anylit |= c != 0
    lit_blocks = anylit[:by * bs, :bx * bs].reshape(by, bs, bx, bs).sum(axis=(1, 3))
    out["dark_blocks"] = int(inner.sum())
    out["lit_pixels_inside_dark_blocks"] = int(lit_blocks[inner].sum())
    assert inner.sum() >= 40, int(inner.sum())
    assert lit_blocks[inner].sum() == 0
    if record is not None:
        for key, v in out.items():
            record(key, v)
    return out
