"""A scene of the composite objects of SURVEY §8(f).4 -- two rectangles (an area light facing down, a red panel), a
tetrahedron, a sphere, two walls -- shared by the CPU and the GPU tests, with the rectangles' (origin, u, v) kept so
that the oracle can test them by the reference's own Rectangle::intersect (Polygon.h:103-153, Polygon.cuh:130-168)."""
import numpy as np

from path_tracer_amd import abi


def composite_scene(w, h):
    from path_tracer_amd.scenes import Emissive, Lambertian, Metal, SceneBuilder
    b = SceneBuilder()
    white, red, mirror = b.material(Lambertian(0.73)), b.material(Lambertian((0.65, 0.05, 0.05))), b.material(Metal((0.9, 0.9, 0.9), 0.0))
    lamp = b.material(Emissive((1.0, 0.9, 0.8), 15.0))
    b.addObject(b.Plane((0, -2, 0), (0, 1, 0), white))
    b.addObject(b.Plane((0, 0, -3), (0, 0, 1), white))
    rect_specs = [((-1.0, 1.9, -2.0), (2.0, 0, 0), (0, 0, 1.5), lamp),      # area light facing down
                  ((-2.0, -2.0, -2.5), (0, 0, 2.0), (0, 2.5, 0), red)]      # a red panel
    rects = []
    for o, u, v, m in rect_specs:
        b.addObject(b.Rectangle(o, u, v, m))
        rects.append((o, u, v, len(b.objects) - 1, m))                     # object id = insertion index
    b.addObject(b.Tetrahedron([(0.0, 0.2, -1.2), (-0.7, -1.0, -0.6), (0.7, -1.0, -0.6), (0.0, -1.0, -1.9)], mirror))
    b.addObject(b.Sphere((1.2, -1.4, -1.0), 0.6, white))
    cam = abi.make_camera((0, 0, 1.6), (0, -0.3, -1), (0, 1, 0), 70.0, w, h, abi.CAMERA_CUDA)
    return b.flatten(), cam, rects


def rectangle_ids_agree(ids, mats, rect_ids, rect_mats, rect_objs):
    """Compares an id map made on the flattened triangles with one made by Rectangle::intersect.  They must be equal
    everywhere but on a rectangle's border or its triangles' shared diagonal, where a ray's membership is decided by
    rounding: returns the number of differing pixels, after asserting that every one of them involves a rectangle and
    has a pixel of the other verdict next to it (i.e. lies on such a line)."""
    diff = (ids != rect_ids) | (mats != rect_mats)
    ys, xs = np.nonzero(diff)
    H, W = ids.shape
    for y, x in zip(ys, xs):
        assert ids[y, x] in rect_objs or rect_ids[y, x] in rect_objs, (x, y, ids[y, x], rect_ids[y, x])
        y0, y1, x0, x1 = max(y - 1, 0), min(y + 2, H), max(x - 1, 0), min(x + 2, W)
        assert (ids[y0:y1, x0:x1] == rect_ids[y, x]).any(), (x, y)        # the other verdict is one pixel away at most
    # both maps show both rectangles
    for o in rect_objs:
        assert (ids == o).sum() > 50 and (rect_ids == o).sum() > 50, o
    return int(diff.sum())
