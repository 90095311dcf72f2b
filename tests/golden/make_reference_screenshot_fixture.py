"""Turns the reference's own published render, images/cuda.png (a screenshot of the CUDA app's window showing
its default Cornell box, 1920x1080 client area, window title bar and a 1-pixel border around it), into a small
fixture: the client area, box-filtered 4x to 480x270, 8 bits per channel.  Data only — no reference source.
Run in the build container (the reference tree is not on the GPU box):
    python tests/golden/make_reference_screenshot_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/images/cuda.png"

im = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float64)
assert im.shape == (1119, 1921, 3), im.shape
client = im[40:1119, 1:1921]                       # rows 0..39 title bar, column 0 border; 1079 rows remain
client = np.concatenate([client, client[-1:]], 0)  # the bottom row is cut off in the screenshot: repeat the last one
assert client.shape == (1080, 1920, 3)
small = client.reshape(270, 4, 480, 4, 3).mean(axis=(1, 3))
out = np.clip(np.rint(small), 0, 255).astype(np.uint8)
np.savez_compressed(os.path.join(HERE, "reference_cuda_screenshot_480x270.npz"), rgb=out,
                    note=np.array("Qervas/path_tracer images/cuda.png, client area 1920x1080 (row 1079 repeated from 1078), 4x4 box filter"))
print("wrote", out.shape, out.dtype, "mean", out.mean(axis=(0, 1)))
