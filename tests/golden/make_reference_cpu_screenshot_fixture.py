"""Turns the reference's own screenshot of the CPU app, images/cpu.png (a grab of the window while the camera was being
moved: one sample per pixel of renderRegion's `direct` estimator, gamma 2.2, 8 bits; the grab holds the title bar, a
9-pixel strip of another window on the left and all but the last 10 columns and 9 rows of the 1280x720 client area),
into a fixture: the client area as an indexed image (palette of the distinct colours + one index per pixel).
Data only -- no reference source.  Run in the build container (the reference tree is not on the GPU box):
    python tests/golden/make_reference_cpu_screenshot_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/images/cpu.png"

im = np.asarray(Image.open(SRC).convert("RGB"))
assert im.shape == (747, 1279, 3), im.shape
client = im[36:, 9:]                                # rows 0..35 title bar, columns 0..8 another window's edge
assert client.shape == (711, 1270, 3)
flat = client.reshape(-1, 3)
palette, index = np.unique(flat, axis=0, return_inverse=True)
assert len(palette) < 65536
np.savez_compressed(os.path.join(HERE, "reference_cpu_screenshot_711x1270.npz"), palette=palette.astype(np.uint8),
                    index=index.reshape(711, 1270).astype(np.uint16),
                    note=np.array("Qervas/path_tracer images/cpu.png, client area rows 0..710, columns 0..1269 of 1280x720"))
print("wrote", client.shape, "colours", len(palette), "most common", palette[np.bincount(index.ravel()).argmax()])
