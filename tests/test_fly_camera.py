"""The scripted stand-in for the CPU app's mouse/keyboard camera (cpu_pathtracer/src/main.cpp:157-205).
Host arithmetic only: runs without a GPU."""
import math

import numpy as np
import pytest

from path_tracer_amd import abi, scenes
from path_tracer_amd.renderer import FlyCamera


def _restated_step(pos, yaw, pitch, dx, dy, keys):
    """main.cpp:157-195 restated with Python floats (IEEE doubles, same operation order)."""
    yaw += dx * 0.002
    pitch = min(max(pitch - dy * 0.002, -math.pi / 2.0 + 0.1), math.pi / 2.0 - 0.1)
    f = np.array([math.cos(pitch) * math.cos(yaw), math.sin(pitch), math.cos(pitch) * math.sin(yaw)])
    r = np.array([f[1] * 0.0 - f[2] * 1.0, f[2] * 0.0 - f[0] * 0.0, f[0] * 1.0 - f[1] * 0.0])
    n = math.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if n > 0:
        r = r * (1.0 / n)
    pos = np.array(pos, dtype=np.float64)
    up = np.array([0.0, 1.0, 0.0])
    for bit, v, s in ((abi.KEY_W, f, 0.1), (abi.KEY_S, f, -0.1), (abi.KEY_A, r, -0.1), (abi.KEY_D, r, 0.1),
                      (abi.KEY_SPACE, up, 0.1), (abi.KEY_CTRL, up, -0.1)):
        if keys & bit:
            pos = pos + v * s
    moved = dx != 0 or dy != 0 or (keys & 63) != 0
    return pos, yaw, pitch, f, moved


def test_first_iteration_gives_the_running_app_camera():
    # the loop's first lookAt, before any input, turns the constructor camera to +X (SURVEY D5 / BASELINE.md §2)
    _, cam = scenes.cpu_default(1280, 720)
    _, want = scenes.cpu_default(1280, 720, running_app_camera=True)
    fly = FlyCamera(cam)
    assert fly.step(0, 0, 0) is False
    for k in range(3):
        assert fly.camera.origin[k] == want.origin[k]
        assert fly.camera.target[k] == want.target[k]
        assert fly.camera.up[k] == want.up[k]
    assert fly.camera.normalize_up == 0 == want.normalize_up


def test_scripted_sequence_matches_restated_formulas():
    _, cam = scenes.cpu_default(640, 360)
    fly = FlyCamera(cam)
    pos, yaw, pitch = [cam.origin[k] for k in range(3)], 0.0, 0.0
    rng = np.random.default_rng(7)
    script = [(0, 0, 0), (12.0, -3.0, 0), (0, 0, abi.KEY_W), (0, 0, abi.KEY_A | abi.KEY_SPACE), (-40.0, 900.0, abi.KEY_S | abi.KEY_D),
              (0, -5000.0, abi.KEY_CTRL)]
    script += [(float(rng.integers(-30, 30)), float(rng.integers(-30, 30)), int(rng.integers(0, 64))) for _ in range(50)]
    for dx, dy, keys in script:
        moved = fly.step(dx, dy, keys)
        pos, yaw, pitch, f, want_moved = _restated_step(pos, yaw, pitch, dx, dy, keys)
        assert moved == want_moved
        assert fly.yaw.value == yaw and fly.pitch.value == pitch
        np.testing.assert_allclose([fly.camera.origin[k] for k in range(3)], pos, rtol=0, atol=1e-15)
        np.testing.assert_allclose([fly.camera.target[k] for k in range(3)], pos + f, rtol=0, atol=1e-15)
    # pitch is clamped 0.1 rad short of the poles (main.cpp:158-159)
    assert -math.pi / 2 + 0.1 <= fly.pitch.value <= math.pi / 2 - 0.1


def test_pitch_clamp_and_error_code():
    from path_tracer_amd._lib import lib
    _, cam = scenes.cpu_default(64, 36)
    fly = FlyCamera(cam)
    fly.step(0, -1e6, 0)
    assert fly.pitch.value == math.pi / 2.0 - 0.1
    fly.step(0, 1e7, 0)
    assert fly.pitch.value == -math.pi / 2.0 + 0.1
    assert lib().pt_fly_camera_step(None, None, None, 0.0, 0.0, 0) == abi.PT_ERR_INVALID
