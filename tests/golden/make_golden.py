"""Generates the committed golden fixtures of the ORACLE (not of the reference: the reference's
CPU program cannot be built in this image and holds no fixtures — see tests/test_oracle_pins.py).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from path_tracer_amd import abi, scenes  # noqa: E402
from tests.oracle_binding import load_oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
o = load_oracle()
L = o.lib

hash_in = [0, 1, 2, 0x5EED0001, 0xFFFFFFFF, 0xC0FFEE, 123456789]
key_in = [[0x5EED0001, 0, 0], [0x5EED0001, 1, 0], [0x5EED0001, 0, 1], [7, 921599, 1023], [0, 8294399, 4095]]
draw_in = [[0, 0, 0], [0xDEADBEEF, 0, 1], [0xDEADBEEF, 1, 0], [0xDEADBEEF, 50, 7], [12345, 3, 4]]
json.dump(dict(hash_in=hash_in, hash_out=[L.pto_hash(x) for x in hash_in],
               key_in=key_in, key_out=[L.pto_path_key(*k) for k in key_in],
               draw_in=draw_in, draw_out=[L.pto_draw(*k) for k in draw_in]),
          open(os.path.join(HERE, "rng.json"), "w"), indent=1)

frames = {}
sc, cam = scenes.cuda_default(48, 27)
frames["cuda_default_path_f32_8spp"] = o.render_path(sc, cam, 0, 8, abi.render_params(), precision=32)
frames["cuda_default_path_f32_8spp_mis_colour"] = o.render_path(
    sc, cam, 0, 8, abi.render_params(flags=abi.FLAG_NEE_MIS | abi.FLAG_LIGHT_COLOUR), precision=32)
sc, cam = scenes.cpu_default(64, 36)
frames["cpu_default_direct_4spp"] = o.render_direct(sc, cam, 0, 4, abi.render_params(integrator=abi.INTEGRATOR_DIRECT))
frames["cuda_default_ids"] = o.primary_ids(*scenes.cuda_default(64, 36))[0]
np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **frames)
print("wrote", os.listdir(HERE))
