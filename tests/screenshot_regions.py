"""Region-by-region comparison with the reference's published render (images/cuda.png, held as the fixture
tests/golden/reference_cuda_screenshot_480x270.npz).

A whole-image mean absolute error lets one wrong BSDF hide in the average.  The regions here are the objects of
the reference's scene as the camera sees them: a cell of the box-filtered image belongs to object k when every
full-resolution pixel under it has primary-hit id k (from the FP64 id pass, which is pinned bit-exactly).  That
gives one region per wall (floor, ceiling, red, green, the two mirrors where visible), per sphere (glass, glossy,
metal, the three small ones, the three lights) and per tetrahedron face, without hand-placed boxes.
"""
import numpy as np

# object ids of cuda_default = insertion order of cuda_pathtracer/src/main.cu:102-200
OBJECT_NAMES = {0: "floor", 1: "ceiling", 2: "left wall (red)", 3: "right wall (green)", 4: "back wall (mirror)",
                5: "front wall (mirror)", 6: "glass sphere", 7: "glossy sphere", 8: "metal sphere", 9: "small glass",
                10: "small glossy 1", 11: "small glossy 2", 12: "main light", 13: "blue light", 14: "orange light",
                15: "tetrahedron face 0", 16: "tetrahedron face 1", 17: "tetrahedron face 2", 18: "tetrahedron face 3"}
TETRAHEDRON = (15, 16, 17, 18)


def pure_cells(ids_full: np.ndarray, factor: int) -> np.ndarray:
    """ids [H, W] -> [H/factor, W/factor] int array: the object id of cells whose pixels all agree, else -1."""
    H, W = ids_full.shape
    b = ids_full.reshape(H // factor, factor, W // factor, factor).astype(np.int64)
    lo, hi = b.min(axis=(1, 3)), b.max(axis=(1, 3))
    return np.where(lo == hi, lo, -1)


def region_stats(img: np.ndarray, ref: np.ndarray, cells: np.ndarray, min_cells: int = 12) -> dict:
    """Per region: number of cells, |mean colour difference| (max over channels) and mean absolute error.
    The tetrahedron's four faces are one region (its back faces are seen only through the front ones)."""
    out = {}
    groups = {k: (k,) for k in OBJECT_NAMES if k not in TETRAHEDRON}
    groups["tetrahedron"] = TETRAHEDRON
    for key, members in groups.items():
        m = np.isin(cells, members)
        n = int(m.sum())
        if n < min_cells:
            continue
        a, b = img[m], ref[m]
        name = key if isinstance(key, str) else OBJECT_NAMES[key]
        out[name] = {"cells": n, "mean_diff": float(np.abs(a.mean(axis=0) - b.mean(axis=0)).max()),
                     "mae": float(np.abs(a - b).mean()), "ref_mean": [float(v) for v in b.mean(axis=0)]}
    return out
