"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bars (north_star / task ③):
  * primary-hit object and material ids: bit-exact (FP64 kernels, contraction off);
  * `direct` integrator: deterministic terms exact, the FP-rounding-governed light term
     statistically;
  * `path` integrator (FP32): same counter RNG on both sides, so almost every sample follows
    the same path; compared per pixel against the float oracle, and against the double oracle
    within 3 sigma of the estimator with a global RMSE bound;
  * RGBA8 / PPM packing: bit-exact on the same accumulation buffer.
"""
import ctypes as C
import os

import numpy as np
import pytest

from path_tracer_amd import abi, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# Every statistical bound below goes through _bound(): it asserts and records (measured, limit), and the module
# writes the record to gpurun_out/parity_measured.json (copied to profiles/ per round).  Limits are set to at most
# three times the value measured on the MI355X (recorded next to each), so a regression that triples the share of
# paths that branch differently fails, instead of hiding under a generous constant.
_MEASURED = {}


def _bound(name, value, limit, at_least=False):
    value, limit = float(value), float(limit)
    _MEASURED[name] = {"measured": value, "limit": limit, "kind": "at_least" if at_least else "below"}
    if at_least:
        assert value >= limit, f"{name}: measured {value:.6g}, must be at least {limit:.6g}"
    else:
        assert value < limit, f"{name}: measured {value:.6g}, bound {limit:.6g}"


@pytest.fixture(scope="module")
def gpu():
    from path_tracer_amd.renderer import Renderer
    r = Renderer(0)
    yield r
    r.close()
    try:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.json"), "w") as f:
            json.dump(_MEASURED, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _setup(r, scene, cam):
    r.set_scene(scene)
    r.set_camera(cam)
    r.reset()


def _rgb(packed):
    """RGBA8 words (R << 16 | G << 8 | B) -> int array [..., 3]."""
    p = packed.astype(np.int64)
    return np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], axis=-1)


def _mean(acc):
    return acc[..., :3] / np.maximum(acc[..., 3:4], 1.0)


# ------------------------------------------------------------------------- T1: ids -----
HIST_CONSTRUCTOR_CAMERA = [0, 39340, 39621, 0, 832589, 0, 0, 10050]
HIST_RUNNING_APP_CAMERA = [0, 0, 1280, 0, 920320, 0, 0, 0]


@pytest.mark.parametrize("running_app,hist", [(False, HIST_CONSTRUCTOR_CAMERA), (True, HIST_RUNNING_APP_CAMERA)])
def test_ids_cpu_default_full_size(gpu, oracle, running_app, hist):
    sc, cam = scenes.cpu_default(1280, 720, running_app_camera=running_app)
    _setup(gpu, sc, cam)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, cam)
    assert np.array_equal(ids, oid) and np.array_equal(mats, omat)
    assert np.bincount(ids.ravel(), minlength=8).tolist() == hist   # the reference's own known answer


@pytest.mark.parametrize("name,size", [("cuda_default", (1280, 720)), ("cuda_default", (1920, 1080)), ("cuda_default", (3840, 2160)),
                                        ("glass_stress", (1280, 720)), ("random4096", (480, 270)),
                                        ("cuda_default", (37, 21))])
def test_ids_bit_exact(gpu, oracle, name, size):
    sc, cam = scenes.builtin(name, *size)
    _setup(gpu, sc, cam)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, cam)
    assert np.array_equal(ids, oid)
    assert np.array_equal(mats, omat)
    # material id = the material bound to that object
    obj2mat = {}
    for pre in ("sph", "pl", "tri"):
        for o, m in zip(sc.arrays[f"{pre}_obj"], sc.arrays[f"{pre}_mat"]):
            obj2mat[int(o)] = int(m)
    hit = ids != abi.MISS
    assert np.array_equal(mats[hit], np.vectorize(obj2mat.get)(ids[hit]))


def test_ids_cpu_camera_on_cuda_scene(gpu, oracle):
    """cuda_default's spheres and planes through the CPU camera formula."""
    sc, _ = scenes.cuda_default(640, 360)
    cam = abi.make_camera((0, 0, 1.6), (0, 0, -1), (0, 1, 0), 90.0, 640, 360, abi.CAMERA_CPU)
    _setup(gpu, sc, cam)
    ids, _ = gpu.primary_ids()
    assert np.array_equal(ids, oracle.primary_ids(sc, cam)[0])


def test_ids_exact_distance_ties_go_to_the_first_object_inserted(gpu, oracle):
    """Scene::intersect keeps the first of two objects at the same distance (strict '<' in insertion order,
    cpu_pathtracer/include/Scene.h:57-69).  The SoA tests spheres, planes, triangles class by class, so a tie across
    classes is decided by the object ids: a sphere, a plane and a triangle that the central ray meets at exactly t = 4,
    inserted in all six orders."""
    import itertools
    from path_tracer_amd.scenes import Lambertian, SceneBuilder
    W = H = 65                                           # generateCameraRay's u = (x + 0.5) / W is 0.5 at x = 32: the axis itself
    cam = abi.make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 60.0, W, H, abi.CAMERA_CUDA)
    for order in itertools.permutations(("sphere", "plane", "triangle")):
        b = SceneBuilder()
        m = {k: b.material(Lambertian(0.5)) for k in order}
        for k in order:
            if k == "sphere": b.addObject(b.Sphere((0, 0, -5), 1.0, m[k]))
            elif k == "plane": b.addObject(b.Plane((0, 0, -4), (0, 0, 1), m[k]))
            else: b.addObject(b.Triangle((-1, -1, -4), (1, -1, -4), (0, 1, -4), m[k]))
        sc = b.flatten()
        _setup(gpu, sc, cam)
        ids, mats = gpu.primary_ids()
        want_ids, want_mats = oracle.primary_ids(sc, cam)
        assert np.array_equal(ids, want_ids) and np.array_equal(mats, want_mats), order
        assert ids[32, 32] == 0 and mats[32, 32] == m[order[0]], (order, ids[32, 32])   # the three-way tie: first inserted
        # the `direct` integrator shades the same winner
        gpu.render(0, 1, abi.render_params(integrator=abi.INTEGRATOR_DIRECT))
        assert np.all(gpu.accum()[..., 3] == 1)


# --------------------------------------------------------------- T2: direct integrator ---
def test_direct_matches_oracle(gpu, oracle):
    sc, cam = scenes.cpu_default(320, 180)
    p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
    _setup(gpu, sc, cam)
    gpu.render(0, 8, p)
    got = gpu.accum()
    want = oracle.render_direct(sc, cam, 0, 8, p)
    assert np.all(got[..., 3] == 8)
    gm, wm = _mean(got), _mean(want)
    # deterministic floor: every sample carries the ambient 0.1; light pixels carry emission 100
    ids, _ = gpu.primary_ids()
    assert np.all(gm >= 0.1 - 1e-6)
    inner_light = (ids == 7)
    inner_light[1:-1, 1:-1] &= (ids[:-2, 1:-1] == 7) & (ids[2:, 1:-1] == 7) & (ids[1:-1, :-2] == 7) & (ids[1:-1, 2:] == 7)
    inner_light[0] = inner_light[-1] = False
    assert np.allclose(gm[inner_light], 100.1, rtol=1e-6) and np.allclose(wm[inner_light], 100.1, rtol=1e-6)
    # the stochastic term: identical sample points, and sin / cos / acos of the light sample are specified (oracle/pt_oracle.c,
    # pt_f64_kernels.hip: IEEE +, -, *, /, sqrt without contraction), so even the rounding-governed tie of D6 falls the same
    # way on both sides: the frames are equal bit for bit
    assert np.array_equal(got, want)
    _MEASURED["direct_320x180_8spp/pixels_exactly_equal"] = {"measured": float((got == want).all(axis=2).mean()), "limit": 1.0, "kind": "at_least"}


def test_direct_config0_bit_exact_at_full_size(gpu, oracle):
    """BASELINE config 1 (cpu_default, `direct`, 1280x720): 64 spp on a band of rows at full width and 2 spp on the whole
    frame, in calls of 16 and of 1 as a frame loop makes them -- the accumulation buffers equal the oracle's bit for bit
    (FP64 arithmetic without contraction on both sides, the float adds of the buffer in the same order)."""
    sc, cam = scenes.cpu_default(1280, 720)
    p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
    _setup(gpu, sc, cam)
    want = None
    for s0 in range(2):
        gpu.render(s0, 1, p)
        want = oracle.render_direct(sc, cam, s0, 1, p, accum=want)
    got = gpu.accum()
    assert np.array_equal(got, want)
    cpu8 = gpu.resolve_to_host(abi.RESOLVE_RGBA8_CPU)
    assert np.array_equal(cpu8, oracle.resolve(want, abi.ACC_SUM, abi.RESOLVE_RGBA8_CPU).reshape(cpu8.shape))
    for s0 in range(2, 66, 16):
        gpu.render(s0, 16, p)
    got = gpu.accum()
    rows = (300, 340)
    band = want.copy()
    for s0 in range(2, 66, 16):
        band = oracle.render_direct(sc, cam, s0, 16, p, accum=band, rows=rows)
    assert np.array_equal(got[rows[0]:rows[1]], band[rows[0]:rows[1]])
    assert np.all(got[..., 3] == 66)
    _MEASURED["direct_1280x720/pixels_exactly_equal"] = {"measured": 1.0, "limit": 1.0, "kind": "at_least"}


def test_direct_against_the_references_own_cpu_screenshot(gpu):
    """The HIP `direct` kernel, the FP64 id pass and the CPU-style resolve against images/cpu.png, the reference's own grab
    of its CPU app at one sample per pixel (tests/cpu_screenshot.py; the oracle passes the same checks in
    tests/test_oracle_pins.py): the unlit byte, every object's place on the screen, the light term pixel by pixel on both
    coloured walls, the pixel formula where the bright channel is not saturated, the shadows."""
    from tests import cpu_screenshot as cs
    sc, cam = cs.scene_and_camera()
    p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
    _setup(gpu, sc, cam)
    ids = gpu.primary_ids()[0]
    frames = []
    for s in range(16):
        gpu.reset()
        gpu.render(s, 1, p)
        frames.append(gpu.resolve_to_host(abi.RESOLVE_RGBA8_CPU).copy())

    def record(key, value):
        _MEASURED[f"cpu_screenshot/{key}"] = {"measured": float(value), "limit": None, "kind": "info"}
    got = cs.check(frames, ids, record)
    assert got["lit_pixels_on_another_object"] < 100 and got["sphere_0.7_0.3/bright_bytes_compared"] > 100


# ----------------------------------------------------------------- T3: path integrator ---
def _frac_differing(gm, wm, tol=1e-3):
    return float((np.abs(gm - wm).max(axis=2) > tol * (1.0 + np.abs(wm).max(axis=2))).mean())


# share of pixels that hold a sample whose path branched differently under FP32 rounding (FMA contraction on the GPU,
# none in the oracle) / relative difference of the image means: limits = 3 x the values measured on the MI355X in
# round 2 (profiles/r2_parity_measured.json)
_N, _M, _C = abi.FLAG_NEE_MIS, abi.FLAG_NEE_MIS | abi.FLAG_LIGHT_COLOUR, abi.FLAG_LIGHT_COLOUR
#                                        measured -> limit                            measured -> limit
PATH_FRAC_LIMIT = {("cuda_default", 0): 0.0034,   # 0.00111       (NEE: a light pick, a cone sample and a shadow test
                   ("cuda_default", _N): 0.030,   # 0.01014        more per bounce that can branch)
                   ("cuda_default", _C): 0.0034,  # 0.00111
                   ("cuda_default", _M): 0.028,   # 0.00931
                   ("glass_stress", 0): 0.0013,   # 0.00042
                   ("cpu_default", 0): 0.0005}    # 0.00007 (one pixel of 14 400)
PATH_MEAN_LIMIT = {("cuda_default", 0): 0.0016,   # 0.00051
                   ("cuda_default", _N): 0.0007,  # 0.00022
                   ("cuda_default", _C): 0.0020,  # 0.00065
                   ("cuda_default", _M): 0.0008,  # 0.00025
                   ("glass_stress", 0): 0.0002,   # 0.00004
                   ("cpu_default", 0): 0.0007}    # 0.00023
# random4096: coordinates reach 50 units, where one float ulp of |o - c|^2 (2.4e-4) is the size of the reference's ray
# epsilon (1e-4, Ray.cuh:38) and of its origin offsets (1e-3; none for Glossy, Material.cuh:204).  Whether a scattered
# ray re-hits the sphere it leaves is then decided by rounding, which FMA contraction (hipcc, as nvcc's default for the
# reference) and the oracle's uncontracted arithmetic round differently: ~25x more paths branch than in the Cornell box.
# That claim is MEASURED in the test: the float and the double build of the oracle (same formulas, same RNG, only the
# rounding differs) disagree on a share of the pixels and of the events; the GPU may differ from the float oracle by at
# most 1.5 x that share plus a small absolute allowance, and its event counts must lie within the band the two oracle
# builds span, widened by the same factor.  (Round 2 used constants here: 0.030 / 0.15 / 0.15.)
RANDOM4096_MEAN_LIMIT = {0: 0.011, _N: 0.02, _M: 0.02}         # measured 0.0036 (the mean is a sum over branched paths: no rounding baseline)


@pytest.mark.parametrize("name,flags,depth", [
    ("cuda_default", 0, 50),
    ("cuda_default", abi.FLAG_NEE_MIS, 50),
    ("cuda_default", abi.FLAG_LIGHT_COLOUR, 50),
    ("cuda_default", abi.FLAG_NEE_MIS | abi.FLAG_LIGHT_COLOUR, 50),
    ("glass_stress", 0, 32),
    ("cpu_default", 0, 50),
])
def test_path_matches_float_oracle(gpu, oracle, name, flags, depth):
    """Same RNG, same formulas: all but a few percent of the pixels agree to 1e-3 after 16 spp
    (a pixel disagrees when one of its paths took a different branch under FP32 rounding)."""
    sc, cam = scenes.builtin(name, 160, 90)
    cam.model = abi.CAMERA_CUDA
    p = abi.render_params(flags=flags, max_depth=depth)
    _setup(gpu, sc, cam)
    for s in range(16):
        gpu.render(s, 1, p)
    got = gpu.accum()
    want = oracle.render_path(sc, cam, 0, 16, p, precision=32)
    assert np.all(got[..., 3] == 16) and np.isfinite(got).all()
    gm, wm = _mean(got), _mean(want)
    tag = f"path_{name}_flags{flags}_160x90_16spp"
    _bound(tag + "/frac_pixels_differing", _frac_differing(gm, wm), PATH_FRAC_LIMIT[(name, flags)])
    _bound(tag + "/rel_mean_diff", abs(gm.mean() - wm.mean()) / wm.mean(), PATH_MEAN_LIMIT[(name, flags)])


@pytest.mark.parametrize("flags", [0, abi.FLAG_NEE_MIS, abi.FLAG_NEE_MIS | abi.FLAG_LIGHT_COLOUR])
def test_path_random_spheres_matches_float_oracle(gpu, oracle, flags):
    """BASELINE config 5's scene (4096 spheres, 64 of them lights) at a size the oracle referees in seconds: primary
    ids, the image, and the event counters; with PT_FLAG_NEE_MIS the light is picked among all 64."""
    sc, cam = scenes.random4096(96, 54)
    assert sc.num_spheres == 4096 and int((sc.arrays["mat_type"][sc.arrays["sph_mat"]] == abi.MAT_EMISSIVE).sum()) == 64
    _setup(gpu, sc, cam)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, cam)
    assert np.array_equal(ids, oid) and np.array_equal(mats, omat)
    p = abi.render_params(max_depth=16, flags=flags | abi.FLAG_COUNTERS)
    gpu.reset_counters()
    gpu.render(0, 4, p)
    got = gpu.accum()
    want, oc = oracle.render_path(sc, cam, 0, 4, p, precision=32, counters=True)
    w64, oc64 = oracle.render_path(sc, cam, 0, 4, p, precision=64, counters=True)
    assert np.all(got[..., 3] == 4) and np.isfinite(got).all()
    tag = f"path_random4096_flags{flags}_96x54_4spp"
    gm, wm, dm = _mean(got), _mean(want), _mean(w64)
    # the rounding baseline: float oracle against double oracle, at both tolerances
    base_1e3, base_2pc = _frac_differing(wm, dm), _frac_differing(wm, dm, tol=2e-2)
    _MEASURED[tag + "/rounding_baseline_frac_1e-3"] = {"measured": base_1e3, "limit": None, "kind": "baseline"}
    _MEASURED[tag + "/rounding_baseline_frac_2pct"] = {"measured": base_2pc, "limit": None, "kind": "baseline"}
    _bound(tag + "/frac_pixels_differing", _frac_differing(gm, wm), 1.5 * base_1e3 + 0.01)
    # ... most of which is numerical jitter, not a path that branched: a sphere of radius 0.3 seen from 50 units away
    # has normals good to 1e-5 (p - c cancels six digits), so a grazing light sample's cosine differs by a fraction
    # of a per cent between two correctly rounded evaluations.  At a 2 % tolerance only the branched paths remain.
    _bound(tag + "/frac_pixels_differing_beyond_2pct", _frac_differing(gm, wm, tol=2e-2), 1.5 * base_2pc + 0.005)
    # the GPU is as close to the double referee as the float oracle is
    _bound(tag + "/frac_pixels_differing_from_double", _frac_differing(gm, dm), 1.5 * base_1e3 + 0.01)
    _bound(tag + "/rel_mean_diff", abs(gm.mean() - wm.mean()) / wm.mean(), RANDOM4096_MEAN_LIMIT[flags])
    g, o, o64 = gpu.counters().as_dict(), oc.as_dict(), oc64.as_dict()
    assert g["samples"] == o["samples"] == o64["samples"] == 96 * 54 * 4
    for k in ("rays", "hits", "sphere_roots", "diffuse_updates", "nee_samples", "shadow_rays"):
        # between the two oracle builds, or outside by no more than half their distance plus counting noise
        lo, hi = min(o[k], o64[k]), max(o[k], o64[k])
        slack = 0.5 * (hi - lo) + 4.0 * np.sqrt(max(hi, 1)) + 0.002 * hi
        _MEASURED[f"{tag}/counter_{k}"] = {"measured": g[k], "limit": [lo - slack, hi + slack], "kind": "band", "float_oracle": o[k], "double_oracle": o64[k]}
        assert lo - slack <= g[k] <= hi + slack, (k, g[k], o[k], o64[k])
    if flags & abi.FLAG_NEE_MIS:
        assert g["nee_samples"] > 0 and g["shadow_rays"] > 0


@pytest.mark.parametrize("size", [(64, 36), (160, 90)])
def test_path_three_sigma_and_rmse(gpu, oracle, size):
    """north_star's tolerance, at two resolutions the double-precision oracle finishes in seconds (2 304 and 14 400
    pixels; 9.4 M and 59 M samples): per-pixel mean within 3 sigma of the estimator (sigma from the oracle's own
    per-sample variance), and global RMSE of the displayed ([0,1]-clamped) image below 1e-3 at 4096 spp.  GPU and
    oracle trace the same sample set, so the RMSE is not Monte-Carlo noise: it is the few paths per thousand that
    branch differently under FP32 rounding, each moving one pixel by (its radiance) / 4096.  The share of such pixels
    and the RMSE without them are recorded too: the margin to 1e-3 is a property of the scene's brightest paths
    (light hits are 15.0 before the clamp), not of the resolution."""
    W, H = size
    N = 4096
    sc, cam = scenes.cuda_default(W, H)
    p = abi.render_params()
    _setup(gpu, sc, cam)
    for s0 in range(0, N, 64):
        gpu.render(s0, 64, p)
    gm = _mean(gpu.accum()).astype(np.float64)
    mean, var = oracle.path_stats(sc, cam, 0, N, p)
    sigma = np.sqrt(var / N)
    tag = f"three_sigma_{W}x{H}_4096spp"
    # The rounding baseline: the FLOAT build of the oracle (the reference's arithmetic, uncontracted) against the same
    # double referee.  Both float implementations differ from the referee by the paths that branch under FP32 rounding;
    # what the tolerance can ask of the GPU is to be no further away than correctly rounded float arithmetic is.
    fm = _mean(oracle.render_path(sc, cam, 0, N, p, precision=32)).astype(np.float64)
    base_inside = float((np.abs(fm - mean) <= 3.0 * sigma + 1e-4).mean())
    base_rmse = float(np.sqrt(((np.clip(fm, 0, 1) - np.clip(mean, 0, 1)) ** 2).mean()))
    _MEASURED[tag + "/float_oracle_frac_inside"] = {"measured": base_inside, "limit": None, "kind": "baseline"}
    _MEASURED[tag + "/float_oracle_rmse_clamped"] = {"measured": base_rmse, "limit": None, "kind": "baseline"}
    # same sample set on both sides: the difference is far inside the estimator's own 3 sigma (the pixels outside are
    # those whose estimator has no variance to speak of -- a light seen directly is 15.0 in every sample -- and one
    # branched path among 4096)
    inside = np.abs(gm - mean) <= 3.0 * sigma + 1e-4
    _bound(tag + "/frac_inside", inside.mean(), min(0.999, base_inside - 0.001), at_least=True)
    diff = np.clip(gm, 0, 1) - np.clip(mean, 0, 1)
    rmse = float(np.sqrt((diff ** 2).mean()))
    _bound(tag + "/rmse_clamped", rmse, 1e-3 * np.sqrt(4096.0 / N))   # north_star: RMSE < 1e-3 at 4096 spp
    _bound(tag + "/rmse_clamped_over_float_oracle_baseline", rmse / base_rmse, 1.3)
    # where the RMSE comes from: pixels that differ by more than float summation noise hold a path that branched
    branched = np.abs(diff).max(axis=2) > 2e-5
    _MEASURED[tag + "/frac_pixels_with_a_branched_path"] = {"measured": float(branched.mean()), "limit": None, "kind": "info"}
    _MEASURED[tag + "/rmse_clamped_without_them"] = {"measured": float(np.sqrt((diff[~branched] ** 2).mean())), "limit": None, "kind": "info"}
    _bound(tag + "/rmse_clamped_without_branched_pixels", float(np.sqrt((diff[~branched] ** 2).mean())), 2e-5)
    # and in relative terms (unclamped means): the image means agree to a few parts in 10^5
    _bound(tag + "/rel_mean_diff", abs(gm.mean() - mean.mean()) / mean.mean(), 2.5e-5)   # measured 3.4e-6 / 8.1e-6


def test_mis_on_equals_mis_off_in_expectation(gpu):
    sc, cam = scenes.cuda_default(64, 36)
    means = []
    for flags in (0, abi.FLAG_NEE_MIS):
        _setup(gpu, sc, cam)
        for s0 in range(0, 2048, 128):
            gpu.render(s0, 128, abi.render_params(flags=flags))
        means.append(_mean(gpu.accum()).mean(axis=(0, 1)))
    _bound("mis_on_vs_off_64x36_2048spp/max_rel_channel_mean_diff", np.abs(means[1] / means[0] - 1.0).max(), 0.001)   # measured 0.00033


def test_counters_match_oracle(gpu, oracle):
    sc, cam = scenes.cuda_default(160, 90)
    for flags in (0, abi.FLAG_NEE_MIS):
        p = abi.render_params(flags=flags | abi.FLAG_COUNTERS)
        _setup(gpu, sc, cam)
        gpu.reset_counters()
        gpu.render(0, 8, p)
        g = gpu.counters().as_dict()
        _, oc = oracle.render_path(sc, cam, 0, 8, p, precision=32, counters=True)
        o = oc.as_dict()
        assert g["samples"] == o["samples"] == 160 * 90 * 8
        # a fraction of a percent of the paths branch differently under FP32 rounding
        for k in ("rays", "hits", "sphere_roots", "diffuse_updates", "nee_samples", "shadow_rays"):
            assert abs(g[k] - o[k]) <= 0.005 * max(o[k], 1), (k, g[k], o[k])
        for a, b in zip(g["bounces"], o["bounces"]):
            assert abs(a - b) <= 0.005 * max(b, 1) + 4 * np.sqrt(max(b, 1))  # rare events: Poisson spread
    # the instrumented kernel renders the same image as the plain one
    _setup(gpu, sc, cam)
    gpu.render(0, 4, abi.render_params())
    plain = gpu.accum()
    _setup(gpu, sc, cam)
    gpu.render(0, 4, abi.render_params(flags=abi.FLAG_COUNTERS))
    assert np.array_equal(plain, gpu.accum())


# ------------------------------------------------- launch-shape independence / determinism ---
def test_result_independent_of_launch_shape(gpu):
    """A pixel's samples are summed in sample order by one lane: the image is bit-identical for
    any pool size, for repeated runs, and for odd frame sizes."""
    for size in ((160, 90), (37, 21), (8, 8), (1, 1), (130, 7)):
        sc, cam = scenes.cuda_default(*size)
        ref = None
        for ppw in (0, 64, 128, 448, 4096):
            _setup(gpu, sc, cam)
            gpu.render(0, 6, abi.render_params(pixels_per_wave=ppw))
            acc = gpu.accum()
            assert np.all(acc[..., 3] == 6)
            if ref is None:
                ref = acc
            assert np.array_equal(acc, ref), (size, ppw)


def test_sample_split_equals_single_range(gpu):
    """Samples [0,32) in one launch vs 4 launches of 8 vs 32 launches of 1: equal up to FP32
    summation order (identical sample values, different grouping of the adds)."""
    sc, cam = scenes.cuda_default(96, 54)
    imgs = []
    for chunk in (32, 8, 1):
        _setup(gpu, sc, cam)
        for s0 in range(0, 32, chunk):
            gpu.render(s0, chunk, abi.render_params())
        imgs.append(gpu.accum())
    for other in imgs[1:]:
        np.testing.assert_allclose(other, imgs[0], rtol=2e-5, atol=1e-5)
    # disjoint ranges on "two GPUs" then a sum == the whole range (the RCCL reduce, on the host)
    _setup(gpu, sc, cam)
    gpu.render(0, 16, abi.render_params())
    a = gpu.accum()
    _setup(gpu, sc, cam)
    gpu.render(16, 16, abi.render_params())
    b = gpu.accum()
    np.testing.assert_allclose(a + b, imgs[0], rtol=2e-5, atol=1e-5)


# --------------------------------------------------------- accumulation / resolve / shim ---
def test_running_mean_mode_matches_oracle(gpu, oracle):
    sc, cam = scenes.cuda_default(96, 54)
    p = abi.render_params(acc_mode=abi.ACC_RUNNING_MEAN)
    _setup(gpu, sc, cam)
    for s in range(12):
        gpu.render(s, 1, p)
    got = gpu.accum()
    want = oracle.render_path(sc, cam, 0, 12, p, precision=32)
    assert np.all(got[..., 3] == 1.0)
    _bound("running_mean_96x54_12frames/frac_pixels_differing", _frac_differing(got[..., :3], want[..., :3]), 0.0006)   # measured 0.00019


def test_launch_render_kernel_shim(gpu):
    """The reference's call shape: launchRenderKernel(output, w, h, frame_count, grid, block)."""
    import torch
    sc, cam = scenes.cuda_default(96, 54)
    _setup(gpu, sc, cam)
    out = torch.full((54, 96, 4), 7.0, device="cuda", dtype=torch.float32)  # frame_count == 0 overwrites
    torch.cuda.synchronize()
    for frame in range(5):
        gpu.launch_render_kernel(out.data_ptr(), frame)
    gpu.sync()
    shim = out.cpu().numpy()
    _setup(gpu, sc, cam)
    for frame in range(5):
        gpu.render(frame, 1, abi.render_params(acc_mode=abi.ACC_RUNNING_MEAN))
    assert np.array_equal(shim, gpu.accum())
    assert np.all(shim[..., 3] == 1.0)


def test_launch_render_kernel_shim_is_ordered_with_the_null_stream():
    """The reference's frame loop runs everything on the legacy default stream: the launch, then its own
    convertToRGBAKernel on the same buffer (main.cu:287-291, Window.cuh:155-160).  A context whose stream nobody chose
    works on the null stream once it is driven through the shim, so a kernel the caller puts on the null stream
    right behind the launch -- here torch's copy, no synchronisation in between -- sees the finished frame, and the
    next launch waits for it."""
    import torch
    from path_tracer_amd.renderer import Renderer
    sc, cam = scenes.cuda_default(1280, 720)      # a launch long enough (0.2 ms) for a missing order to show
    r = Renderer(0)
    try:
        _setup(r, sc, cam)
        out = torch.zeros((720, 1280, 4), device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        snaps = []
        for frame in range(4):
            r.launch_render_kernel(out.data_ptr(), frame)
            snaps.append(out.clone())               # on the null stream, like the reference's convertToRGBAKernel
        torch.cuda.synchronize()
        _setup(r, sc, cam)
        for frame in range(4):
            r.render(frame, 1, abi.render_params(acc_mode=abi.ACC_RUNNING_MEAN))
            assert np.array_equal(snaps[frame].cpu().numpy(), r.accum()), frame
    finally:
        r.close()


def test_resolve_bit_exact_on_same_buffer(gpu, oracle):
    sc, cam = scenes.cuda_default(160, 90)
    _setup(gpu, sc, cam)
    gpu.render(0, 8, abi.render_params())
    acc = gpu.accum()
    for mode in (abi.RESOLVE_RGBA8_CUDA, abi.RESOLVE_FLOAT_MEAN):
        got = gpu.resolve_to_host(mode)
        want = oracle.resolve(acc, abi.ACC_SUM, mode).reshape(got.shape)
        assert np.array_equal(got, want), mode
    # the gamma modes evaluate std::pow in double in the reference's order (main.cpp:229-234; Color.h:45-51,86-91):
    # byte work, bit-exact like the others
    for mode in (abi.RESOLVE_RGBA8_CPU, abi.RESOLVE_RGB8_PPM):
        got = gpu.resolve_to_host(mode)
        want = oracle.resolve(acc, abi.ACC_SUM, mode).reshape(got.shape)
        assert np.array_equal(got, want), mode
    # ... also on a buffer that exercises the clamps and the whole 8-bit range: a ramp of means from -0.1 to 1.2
    ramp = np.zeros((90, 160, 4), np.float32)
    ramp[..., 0] = np.linspace(-0.1, 1.2, 160 * 90, dtype=np.float32).reshape(90, 160) * 3.0
    ramp[..., 1] = np.linspace(0.0, 1.0, 160 * 90, dtype=np.float32).reshape(90, 160)[::-1] * 3.0
    ramp[..., 2] = (np.arange(160 * 90, dtype=np.float32).reshape(90, 160) % 256) / 255.0 * 3.0
    ramp[..., 3] = 3.0
    gpu.upload_accum(ramp)
    for mode in (abi.RESOLVE_RGBA8_CUDA, abi.RESOLVE_RGBA8_CPU, abi.RESOLVE_FLOAT_MEAN, abi.RESOLVE_RGB8_PPM):
        got = gpu.resolve_to_host(mode)
        want = oracle.resolve(ramp, abi.ACC_SUM, mode).reshape(got.shape)
        assert np.array_equal(got, want), mode


def test_fused_rgba8_equals_resolve_pass(gpu):
    sc, cam = scenes.cuda_default(160, 90)
    for mode in (abi.ACC_SUM, abi.ACC_RUNNING_MEAN):
        _setup(gpu, sc, cam)
        for s in range(4):
            gpu.render(s, 1, abi.render_params(acc_mode=mode, fused_rgba8=1))
        fused = gpu.readback(abi.RESOLVE_RGBA8_CUDA)
        assert np.array_equal(fused, gpu.resolve_to_host(abi.RESOLVE_RGBA8_CUDA))


def test_checkpoint_resume(gpu):
    sc, cam = scenes.cuda_default(64, 36)
    _setup(gpu, sc, cam)
    gpu.render(0, 4, abi.render_params())
    half = gpu.accum()
    gpu.render(4, 4, abi.render_params())
    full = gpu.accum()
    _setup(gpu, sc, cam)
    gpu.upload_accum(half)
    gpu.render(4, 4, abi.render_params())
    assert np.array_equal(full, gpu.accum())


def test_reset_clears_accumulation(gpu):
    sc, cam = scenes.cuda_default(64, 36)
    _setup(gpu, sc, cam)
    gpu.render(0, 2, abi.render_params())
    gpu.reset()
    assert not gpu.accum().any()


def test_bound_torch_tensor_is_the_accumulation_buffer(gpu):
    import torch
    sc, cam = scenes.cuda_default(64, 36)
    _setup(gpu, sc, cam)
    gpu.render(0, 3, abi.render_params())
    own = gpu.accum()
    t = torch.zeros((36, 64, 4), device="cuda", dtype=torch.float32)
    gpu.use_torch_stream()
    gpu.bind_tensor(t)
    gpu.render(0, 3, abi.render_params())
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), own)
    gpu.unbind()
    gpu.use_own_stream()
    gpu.render(3, 1, abi.render_params())
    assert np.all(gpu.accum()[..., 3] == 4)  # back on the context's own buffer


def test_stream_switch_keeps_queued_work_ordered(gpu):
    """pt_set_stream / pt_use_own_stream with work still queued: the frame clear and the launches
    queued on the old stream stay ordered before the launches queued on the new one."""
    import torch
    sc, cam = scenes.cuda_default(640, 360)
    _setup(gpu, sc, cam)
    for s in range(6):
        gpu.render(s, 1, abi.render_params())
    want = gpu.accum()
    side = torch.cuda.Stream()
    for flight in (1, 3):
        gpu.set_frames_in_flight(flight)
        gpu.use_own_stream()
        gpu.set_camera(scenes.cuda_default(64, 36)[1])  # reallocates: the clear below is queued, not waited for
        gpu.set_camera(cam)
        gpu.render(0, 1, abi.render_params())
        gpu.render(1, 1, abi.render_params())
        with torch.cuda.stream(side):
            gpu.use_torch_stream()
            gpu.render(2, 1, abi.render_params())
            gpu.render(3, 1, abi.render_params())
        gpu.use_own_stream()
        gpu.render(4, 1, abi.render_params())
        gpu.render(5, 1, abi.render_params())
        got = gpu.accum()
        assert np.array_equal(got[..., 3], want[..., 3])
        assert np.array_equal(got, want)      # overlapped launches fold in launch order: the serial image bit for bit
    gpu.set_frames_in_flight(1)


# ------------------------------------------------------------------ frames in flight ---
@pytest.mark.parametrize("n", [2, 3, 4, 8])
def test_frames_in_flight_match_serial(gpu, n):
    """Launches overlapped on n chains give the serial launches' image BIT FOR BIT: each writes its own sums into
    its chain's buffer and the fold passes add them to the accumulation buffer in launch order, so the buffer grows
    as acc += launch k, k = 0, 1, 2, ... exactly as under serial launches, however the launches happened to overlap;
    and the display made behind launch k is the serial display after frame k."""
    sc, cam = scenes.cuda_default(320, 180)
    frames = 11  # not a multiple of n: the chains end a burst with unequal counts
    _setup(gpu, sc, cam)
    for s in range(frames):
        gpu.render(s, 1, abi.render_params(fused_rgba8=1))
    serial = gpu.accum()
    serial_rgba = gpu.resolve_to_host(abi.RESOLVE_RGBA8_CUDA)
    gpu.set_frames_in_flight(n)
    try:
        runs = []
        for rep in range(2):
            _setup(gpu, sc, cam)
            for s in range(frames):
                gpu.render(s, 1, abi.render_params(fused_rgba8=1))
            runs.append(gpu.accum())     # reads join the chains
        assert np.array_equal(runs[0], runs[1])
        assert np.all(runs[0][..., 3] == frames)
        assert np.array_equal(runs[0], serial)
        assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), serial_rgba)        # the display of the last fold pass
        assert np.array_equal(gpu.resolve_to_host(abi.RESOLVE_RGBA8_CUDA), serial_rgba)
        # a burst, a flush and another burst
        _setup(gpu, sc, cam)
        for s in range(5):
            gpu.render(s, 2, abi.render_params())        # 2 spp per launch
        gpu.flush()
        for s in range(5, 8):
            gpu.render(s, 2, abi.render_params())
        got = gpu.accum()
        assert np.all(got[..., 3] == 16)
        _setup(gpu, sc, cam)
        gpu.set_frames_in_flight(1)
        for s in range(8):
            gpu.render(s, 2, abi.render_params())
        assert np.array_equal(got, gpu.accum())
    finally:
        gpu.set_frames_in_flight(1)


def test_frames_in_flight_make_every_frames_display(gpu):
    """Frames in flight collect consecutive frames into one launch (up to 64) and fold them one by one: the display
    of EVERY frame is made, from the sums as they stand after that frame -- the newest in the display buffer, the
    ones before it in the history planes -- and each equals, byte for byte, the display the serial loop shows after
    that frame (renderKernel + convertToRGBAKernel per frame, main.cu:284-303)."""
    from path_tracer_amd._lib import PtError
    sc, cam = scenes.cuda_default(192, 108)
    p = abi.render_params(fused_rgba8=1)
    frames = 70   # one launch of 64 frames and one of 6
    _setup(gpu, sc, cam)
    serial = []
    for s in range(frames):
        gpu.render(s, 1, p)
        serial.append(gpu.readback(abi.RESOLVE_RGBA8_CUDA))
    serial_acc = gpu.accum()
    assert gpu.history_frames() == 0
    gpu.set_frames_in_flight(3)
    try:
        _setup(gpu, sc, cam)
        for s in range(11):
            gpu.render(s, 1, p)
        assert gpu.history_frames() == 10               # (joins: launches the 11 frames held back and folds them)
        assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), serial[10])
        for back in range(1, 11):
            assert np.array_equal(gpu.readback_history(back), serial[10 - back]), back
        with pytest.raises(PtError, match="not held"):
            gpu.readback_history(11)
        with pytest.raises(PtError, match="not held"):
            gpu.readback_history(0)
        for s in range(11, frames):                      # 59 more: the burst continues on the same buffer
            gpu.render(s, 1, p)
        assert np.array_equal(gpu.accum(), serial_acc)
        assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), serial[frames - 1])
        assert gpu.history_frames() == 58
        for back in (1, 2, 15, 31, 58):
            assert np.array_equal(gpu.readback_history(back), serial[frames - 1 - back]), back
        # more frames than one launch holds, in one go: 64 + 6; the history is the last fold pass's
        _setup(gpu, sc, cam)
        before = gpu.path_launches()
        for s in range(frames):
            gpu.render(s, 1, p)
        assert np.array_equal(gpu.accum(), serial_acc)
        assert gpu.path_launches() - before == 2
        assert gpu.history_frames() == 5
        for back in range(1, 6):
            assert np.array_equal(gpu.readback_history(back), serial[frames - 1 - back]), back
        gpu.resolve(abi.RESOLVE_RGBA8_CUDA)              # anything else that writes the display ends the history
        assert gpu.history_frames() == 0
    finally:
        gpu.set_frames_in_flight(1)


def test_frames_in_flight_launch_boundaries(gpu):
    """What may share a launch and what may not: frames whose parameters, sample count or sample range do not
    continue the ones held back start a new launch, and a camera change launches what is held back with the old
    camera.  Whatever the partition, the buffer is the serial loop's bit for bit."""
    sc, cam = scenes.cuda_default(160, 90)
    _, cam_b = scenes.glass_stress(160, 90)
    cam_dof = abi.make_camera((0.0, 0.5, 3.5), (0.0, 0.0, 0.0), (0, 1, 0), 60.0, 160, 90, abi.CAMERA_CUDA, aspect_ratio=160 / 90,
                              aperture=0.2, focus_dist=3.0, use_dof=1)
    calls = ([("render", s, 1, dict()) for s in range(5)] +                      # five frames of 1 spp
             [("render", 5 + 3 * k, 3, dict()) for k in range(4)] +               # four of 3 spp
             [("render", 17, 3, dict(max_depth=7)), ("render", 20, 3, dict(max_depth=7))] +  # other parameters
             [("render", 40, 3, dict(max_depth=7))] +                             # a gap in the sample range
             [("camera", cam_b)] +
             [("render", 43 + k, 1, dict(fused_rgba8=1)) for k in range(6)] +      # another camera, with display
             [("camera", cam)] +
             [("render", 49 + 2 * k, 2, dict(flags=abi.FLAG_NEE_MIS)) for k in range(3)] +
             [("render", 60 + 8 * k + 2, 2, dict()) for k in range(7)] +          # equally spaced: rank 1 of 4, 2 spp per step
             [("render", 200, 1, dict()), ("render", 190, 1, dict())] +           # going backwards: never one launch
             [("camera", cam_dof)] +
             [("render", 300 + k, 1, dict(fused_rgba8=1)) for k in range(5)])      # thin-lens rays (finished when the ray is taken)
    def run(r):
        _setup(r, sc, cam)
        for c in calls:
            if c[0] == "camera":
                r.set_camera(c[1])
            else:
                r.render(c[1], c[2], abi.render_params(**c[3]))
        return r.accum()
    want = run(gpu)
    assert np.all(want[..., 3] == 5 + 12 + 6 + 3 + 6 + 6 + 14 + 2 + 5)
    for n in (2, 3, 8):
        gpu.set_frames_in_flight(n)
        try:
            got = run(gpu)
        finally:
            gpu.set_frames_in_flight(1)
        assert np.array_equal(got, want), n


@pytest.mark.parametrize("size", [(8200, 8), (8, 8200), (65536, 2)])
def test_frames_in_flight_on_long_thin_frames(gpu, size):
    """A lane keeps (x, y, frame) of its next ray in one register; the bits are dealt out by the frame's shape (14 + 3
    for 8200 x 8, 17 + 1 for 65536 x 2, the rest for the frame number), so frames far from 16:9 are collected into
    launches like any other and give the serial loop's image and displays."""
    sc, cam = scenes.cuda_default(*size)
    q = abi.render_params(max_depth=6, fused_rgba8=1)
    _setup(gpu, sc, cam)
    for s in range(5):
        gpu.render(s, 1, q)
    want, want_rgba = gpu.accum(), gpu.readback(abi.RESOLVE_RGBA8_CUDA)
    gpu.set_frames_in_flight(2)
    try:
        _setup(gpu, sc, cam)
        before = gpu.path_launches()
        for s in range(5):
            gpu.render(s, 1, q)
        assert np.array_equal(gpu.accum(), want)
        assert gpu.path_launches() - before == 1
        assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), want_rgba)
        assert gpu.history_frames() == 4
    finally:
        gpu.set_frames_in_flight(1)
    _setup(gpu, *scenes.cuda_default(64, 36))   # (back to a small frame: the chains' buffers are resized with it)


def test_frames_in_flight_cut_large_calls_into_launches(gpu):
    """One call for many samples per pixel is rendered as frames of 1..4 samples of a multi-frame launch
    (pt_api.cpp, part_size): same samples, each once, whatever the call size, and -- the cut depends on the call
    alone -- the same image bit for bit with and without frames in flight."""
    sc, cam = scenes.cuda_default(320, 180)
    _setup(gpu, sc, cam)
    gpu.render(7, 150, abi.render_params())
    serial = gpu.accum()
    gpu.set_frames_in_flight(3)
    try:
        runs = []
        for rep in range(2):
            _setup(gpu, sc, cam)
            gpu.render(7, 150, abi.render_params())
            runs.append(gpu.accum())
        assert np.array_equal(runs[0], runs[1])
        assert np.all(runs[0][..., 3] == 150)
        assert np.array_equal(runs[0], serial)
        # the same samples asked for in three calls of unequal size (other frames: equal up to float summation order)
        _setup(gpu, sc, cam)
        gpu.render(7, 49, abi.render_params())
        gpu.render(56, 100, abi.render_params())
        gpu.render(156, 1, abi.render_params())
        got = gpu.accum()
        assert np.all(got[..., 3] == 150)
        np.testing.assert_allclose(got[..., :3], serial[..., :3], rtol=1e-5, atol=1e-5)
    finally:
        gpu.set_frames_in_flight(1)


def test_large_serial_calls_are_split_over_the_chains(gpu, monkeypatch):
    """Without frames in flight a call for >= 12 samples per pixel is a burst of its own on three chains,
    folded before the call returns: the same samples as the single launch (PT_AMD_CHUNK=0), reproducible,
    and a fused display that equals the resolve pass."""
    from path_tracer_amd.renderer import Renderer
    sc, cam = scenes.cuda_default(320, 180)
    monkeypatch.setenv("PT_AMD_CHUNK", "0")
    whole = Renderer(0)
    monkeypatch.delenv("PT_AMD_CHUNK")
    try:
        _setup(whole, sc, cam)
        whole.render(3, 40, abi.render_params())
        want = whole.accum()
    finally:
        whole.close()
    runs = []
    for rep in range(2):
        _setup(gpu, sc, cam)
        gpu.render(3, 40, abi.render_params(fused_rgba8=1))
        fused = gpu.readback(abi.RESOLVE_RGBA8_CUDA)
        runs.append(gpu.accum())
        assert np.array_equal(fused, gpu.resolve_to_host(abi.RESOLVE_RGBA8_CUDA))
    assert np.array_equal(runs[0], runs[1])
    assert np.all(runs[0][..., 3] == 40)
    np.testing.assert_allclose(runs[0][..., :3], want[..., :3], rtol=1e-5, atol=1e-5)
    # and the next call continues on the same buffer
    gpu.render(43, 5, abi.render_params())
    assert np.all(gpu.accum()[..., 3] == 45)


def test_kernel_builds_with_and_without_lane_parallel_mode_agree(gpu, oracle, monkeypatch):
    """The host picks the build with or without the lane-parallel closest hit per launch (pt_api.cpp);
    both trace the same samples: their images agree like either agrees with the oracle (the two closest-hit
    forms round differently, so a path in a few thousand takes another branch)."""
    from path_tracer_amd.renderer import Renderer
    for name, size, depth in (("cuda_default", (160, 90), 50), ("random4096", (64, 36), 8)):
        sc, cam = scenes.builtin(name, *size)
        p = abi.render_params(max_depth=depth)
        imgs = []
        for few in ("0", "48"):
            monkeypatch.setenv("PT_AMD_FEW_MAX", few)
            r = Renderer(0)
            try:
                _setup(r, sc, cam)
                for s in range(8):
                    r.render(s, 1, p)
                imgs.append(r.accum())
            finally:
                r.close()
        monkeypatch.delenv("PT_AMD_FEW_MAX")
        assert np.array_equal(imgs[0][..., 3], imgs[1][..., 3])
        _bound(f"lane_parallel_{name}/frac_differing_between_builds", _frac_differing(_mean(imgs[0]), _mean(imgs[1])), 0.002)   # measured 0
        want = oracle.render_path(sc, cam, 0, 8, p, precision=32)
        for few, img in zip(("off", "on"), imgs):
            _bound(f"lane_parallel_{name}/frac_differing_vs_oracle_{few}", _frac_differing(_mean(img), _mean(want)),
                   0.0017 if name == "cuda_default" else 0.06)   # measured 0.00056 / 0.0204


def test_frames_in_flight_random_call_sequences(gpu):
    """Any interleaving of renders with the calls that read, replace or re-target the accumulation buffer keeps
    the books: the buffer always holds exactly the samples rendered since the last reset, each once."""
    rng = np.random.default_rng(20261110)
    sc, cam = scenes.cuda_default(96, 54)
    sc2, cam2 = scenes.glass_stress(96, 54)
    _setup(gpu, sc, cam)
    expected = 0      # samples per pixel the buffer must hold
    next_sample = 0
    try:
        for step in range(300):
            op = rng.integers(0, 12)
            if op <= 4:
                n = int(rng.integers(1, 4))
                gpu.render(next_sample, n, abi.render_params(fused_rgba8=int(rng.integers(0, 2)), max_depth=6))
                next_sample += n; expected += n
            elif op == 5:
                gpu.set_frames_in_flight(int(rng.integers(1, 9)))
            elif op == 6:
                gpu.flush()
            elif op == 7:
                acc = gpu.accum()
                assert np.all(acc[..., 3] == expected), (step, expected)
            elif op == 8:
                gpu.reset(); expected = 0
            elif op == 9:
                gpu.resolve(abi.RESOLVE_RGBA8_CUDA)
            elif op == 10:
                gpu.set_camera(cam2 if rng.integers(0, 2) else cam)        # same size: keeps the samples
            else:
                gpu.set_scene(sc2 if rng.integers(0, 2) else sc)
        acc = gpu.accum()
        assert np.all(acc[..., 3] == expected) and np.isfinite(acc).all()
    finally:
        gpu.set_frames_in_flight(1)


def test_frames_in_flight_full_size_and_reset(gpu):
    sc, cam = scenes.cuda_default(1280, 720)
    _setup(gpu, sc, cam)
    gpu.set_frames_in_flight(3)
    try:
        for s in range(32):
            gpu.render(s, 1, abi.render_params(fused_rgba8=1))
        acc = gpu.accum()
        assert np.all(acc[..., 3] == 32) and np.isfinite(acc).all()
        fused = gpu.readback(abi.RESOLVE_RGBA8_CUDA)      # previews while the launches ran; exact once the chains are folded
        assert np.array_equal(fused, gpu.resolve_to_host(abi.RESOLVE_RGBA8_CUDA))
        gpu.render(32, 1, abi.render_params())
        gpu.reset()                                       # discards the open burst too
        assert not gpu.accum().any()
        gpu.render(0, 1, abi.render_params())
        gpu.render(1, 1, abi.render_params())
        assert np.all(gpu.accum()[..., 3] == 2)
    finally:
        gpu.set_frames_in_flight(1)


def test_frames_in_flight_with_bound_tensor(gpu):
    import torch
    sc, cam = scenes.cuda_default(160, 90)
    _setup(gpu, sc, cam)
    for s in range(6):
        gpu.render(s, 1, abi.render_params())
    want = gpu.accum()
    t = torch.zeros((90, 160, 4), device="cuda", dtype=torch.float32)
    gpu.use_torch_stream()
    gpu.bind_tensor(t)
    gpu.set_frames_in_flight(2)
    try:
        for s in range(6):
            gpu.render(s, 1, abi.render_params())
        gpu.flush()                                       # the tensor is read by torch, not by the library
        torch.cuda.synchronize()
        got = t.cpu().numpy()
        assert np.array_equal(got, want)
    finally:
        gpu.set_frames_in_flight(1)
        gpu.unbind()
        gpu.use_own_stream()


# ------------------------------------------------------------ scripted camera (§8f.2) ---
def test_scripted_camera_resets_accumulation_on_movement(gpu, oracle):
    """The CPU app's frame loop (main.cpp:153-237) with a scripted input sequence: frames add up
    while the camera rests, any movement re-aims it and restarts the accumulation."""
    from path_tracer_amd.renderer import FlyCamera
    sc, cam = scenes.cpu_default(160, 90)
    p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
    fly = FlyCamera(cam)
    gpu.set_scene(sc)
    frame_count = 0
    script = [(0, 0, 0)] * 3 + [(25.0, -10.0, abi.KEY_W)] + [(0, 0, 0)] * 2
    for i, (dx, dy, keys) in enumerate(script):
        if fly.step(dx, dy, keys) or i == 0:
            gpu.set_camera(fly.camera)
            gpu.reset()
            frame_count = 0
        gpu.render(frame_count, 1, p)
        frame_count += 1
    got = gpu.accum()
    assert frame_count == 3 and np.all(got[..., 3] == 3)
    # the same three frames rendered from the final camera in one go
    gpu.set_camera(fly.camera)
    gpu.reset()
    gpu.render(0, 3, p)
    np.testing.assert_allclose(got[..., :3], gpu.accum()[..., :3], rtol=1e-6, atol=1e-6)
    # and the moved camera sees what the oracle sees from there (primary ids bit-exact)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, fly.camera)
    assert np.array_equal(ids, oid) and np.array_equal(mats, omat)


# ------------------------------------------------- composite objects (§8f.4) on the GPU ---
from tests.composite_scene import composite_scene as _composite_scene, rectangle_ids_agree


def test_composite_objects_ids_and_path(gpu, oracle):
    sc, cam, rects = _composite_scene(200, 120)
    _setup(gpu, sc, cam)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, cam)
    assert np.array_equal(ids, oid) and np.array_equal(mats, omat)
    assert {2, 3, 4} <= set(np.unique(ids).tolist())          # both rectangles and the tetrahedron are visible
    # the rectangles as the REFERENCE tests them (Rectangle_t::intersect, Polygon.cuh:130-168: alpha, beta in [0, 1]),
    # against the HIP id pass on the two triangles the host flattens each to
    rid, rmat = oracle.primary_ids_rect(sc, cam, rects)
    differing = rectangle_ids_agree(ids, mats, rid, rmat, [r[3] for r in rects])
    _bound("composite_200x120/pixels_where_rectangle_formula_and_triangles_differ", differing, 3)   # measured 0 (of 24 000)
    p = abi.render_params(max_depth=12)
    gpu.render(0, 16, p)
    got = gpu.accum()
    want = oracle.render_path(sc, cam, 0, 16, p, precision=32)
    assert np.all(got[..., 3] == 16) and np.isfinite(got).all()
    _bound("composite_200x120_16spp/frac_pixels_differing", _frac_differing(_mean(got), _mean(want)), 0.002)   # measured 0
    _bound("composite_200x120_16spp/rel_mean_diff", abs(_mean(got).mean() - _mean(want).mean()) / _mean(want).mean(), 0.0005)   # measured 0


# ------------------------------------------------------- depth of field (§8f.4) ---
def test_depth_of_field_matches_oracle(gpu, oracle):
    """Thin-lens camera rays (Camera_t::getRayDevice, Camera.cu:31-47) in the path integrator."""
    sc, cam = scenes.cuda_default(240, 135)
    p = abi.render_params(max_depth=8)
    _setup(gpu, sc, cam)
    gpu.render(0, 16, p)
    pinhole = gpu.accum()
    cam.use_dof, cam.aperture, cam.focus_dist = 1, 0.0, 2.6      # a zero aperture is the pinhole camera, bit for bit
    _setup(gpu, sc, cam)
    gpu.render(0, 16, p)
    assert np.array_equal(gpu.accum(), pinhole)
    cam.aperture = 0.4
    _setup(gpu, sc, cam)
    gpu.render(0, 16, p)
    got = gpu.accum()
    want = oracle.render_path(sc, cam, 0, 16, p, precision=32)
    assert np.all(got[..., 3] == 16) and np.isfinite(got).all()
    _bound("depth_of_field_240x135_16spp/frac_pixels_differing", _frac_differing(_mean(got), _mean(want)), 0.002)   # measured 0
    # it is a different image from the pinhole one (edges blur) with the same overall brightness
    assert np.abs(_mean(got) - _mean(pinhole)).mean() > 0.01
    _bound("depth_of_field_240x135_16spp/rel_brightness_vs_pinhole", abs(_mean(got).mean() - _mean(pinhole).mean()) / _mean(pinhole).mean(), 0.025)   # measured 0.0084


# ------------------------------------------------ general planes, empty and degenerate scenes ---
def _tilted_scene(w, h):
    """Planes whose normals are not axis-aligned (the general plane test), one degenerate normal,
    more than four of them (two loop trips), and a sphere light."""
    from path_tracer_amd.scenes import Emissive, Glossy, Lambertian, Metal, SceneBuilder
    b = SceneBuilder()
    white, green, mirror = b.material(Lambertian(0.73)), b.material(Lambertian((0.12, 0.45, 0.15))), b.material(Metal((0.8, 0.8, 0.9), 0.1))
    gloss, lamp = b.material(Glossy((0.9, 0.6, 0.2), 0.3, 0.8)), b.material(Emissive((1, 1, 1), 15.0))
    b.addObject(b.Plane((0, -2, 0), (0.1, 1.0, 0.05), white))          # tilted floor
    b.addObject(b.Plane((0, 0, -4), (0.2, -0.1, 1.0), white))          # tilted back wall
    b.addObject(b.Plane((-2.5, 0, 0), (1.0, 0.15, 0.1), green))        # tilted left wall
    b.addObject(b.Plane((2.5, 0, 0), (-1.0, 0.1, -0.2), mirror))       # tilted mirror wall
    b.addObject(b.Plane((0, 3, 0), (0.05, -1.0, 0.1), white))          # tilted ceiling
    b.addObject(b.Plane((0, 0, 0), (0.0, 0.0, 0.0), green))            # degenerate normal: can never be hit
    b.addObject(b.Plane((0, 0, 3), (0, 0, -1), white))                 # one axis-aligned plane among them
    b.addObject(b.Sphere((0.3, -1.2, -1.5), 0.7, gloss))
    b.addObject(b.Sphere((0, 2.2, -1.5), 0.4, lamp).makeEmissive((1, 1, 1), 15.0))
    cam = abi.make_camera((0, 0, 1.6), (0, -0.2, -1), (0, 1, 0), 75.0, w, h, abi.CAMERA_CUDA)
    return b.flatten(), cam


@pytest.mark.parametrize("flags", [0, abi.FLAG_NEE_MIS])
def test_general_planes_ids_and_path(gpu, oracle, flags):
    sc, cam = _tilted_scene(192, 108)
    _setup(gpu, sc, cam)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, cam)
    assert np.array_equal(ids, oid) and np.array_equal(mats, omat)
    assert 5 not in set(np.unique(ids).tolist())               # the degenerate plane is never seen
    p = abi.render_params(max_depth=10, flags=flags)
    gpu.render(0, 16, p)
    got = gpu.accum()
    want = oracle.render_path(sc, cam, 0, 16, p, precision=32)
    assert np.all(got[..., 3] == 16) and np.isfinite(got).all()
    _bound(f"general_planes_flags{flags}_192x108_16spp/frac_pixels_differing", _frac_differing(_mean(got), _mean(want)),
           0.006 if flags else 0.002)   # measured 0.0019 / 0
    _bound(f"general_planes_flags{flags}_192x108_16spp/rel_mean_diff", abs(_mean(got).mean() - _mean(want).mean()) / _mean(want).mean(), 0.0005)   # measured 1.5e-5 / 0


def test_empty_and_lightless_scenes(gpu, oracle):
    from path_tracer_amd.scenes import Lambertian, SceneBuilder
    # no primitives at all is refused at upload with a clear error (nothing to render)
    from path_tracer_amd._lib import PtError
    b = SceneBuilder()
    b.material(Lambertian(0.5))
    cam = abi.make_camera((0, 0, 1.6), (0, 0, -1), (0, 1, 0), 90.0, 64, 36, abi.CAMERA_CUDA)
    with pytest.raises(PtError, match="at least one primitive"):
        gpu.set_scene(b.flatten())
    # a scene nobody looks at: every ray misses, the image is black, ids are all MISS
    b = SceneBuilder()
    b.addObject(b.Sphere((0, 0, 50.0), 0.5, b.material(Lambertian(0.5))))
    sc = b.flatten()
    _setup(gpu, sc, cam)
    ids, _ = gpu.primary_ids()
    assert np.all(ids == abi.MISS)
    for flags in (0, abi.FLAG_NEE_MIS):
        gpu.reset()
        gpu.render(0, 3, abi.render_params(flags=flags))
        acc = gpu.accum()
        assert np.all(acc[..., 3] == 3) and not acc[..., :3].any()
    # geometry but no light: light sampling has nothing to pick, the image stays black and finite
    b = SceneBuilder()
    m = b.material(Lambertian(0.5))
    b.addObject(b.Plane((0, -1, 0), (0, 1, 0), m))
    b.addObject(b.Sphere((0, 0, -2), 0.8, m))
    sc = b.flatten()
    _setup(gpu, sc, cam)
    gpu.render(0, 4, abi.render_params(flags=abi.FLAG_NEE_MIS, max_depth=6))
    acc = gpu.accum()
    assert np.all(acc[..., 3] == 4) and not acc[..., :3].any()
    want = oracle.render_path(sc, cam, 0, 4, abi.render_params(flags=abi.FLAG_NEE_MIS, max_depth=6), precision=32)
    assert not want[..., :3].any()


# ------------------------------------------- the reference's own published render ---
def test_path_against_the_references_own_render(gpu):
    """images/cuda.png of the reference (a screenshot of the CUDA app: default scene, 1920x1080 client area, unknown
    sample count, 8 bits, truncated) as the fixture tests/golden/reference_cuda_screenshot_480x270.npz.  Same
    scene, camera, resolution; 1024 samples; clamp as convertToRGBAKernel does; 4x4 box filter like the fixture.
    Whole image, then object by object (tests/screenshot_regions.py: every wall, sphere, light and the tetrahedron,
    delimited by the bit-exact primary ids), so that one wrong BSDF cannot hide in the average."""
    from tests.screenshot_regions import pure_cells, region_stats
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_cuda_screenshot_480x270.npz"))["rgb"].astype(np.float64) / 255.0
    sc, cam = scenes.cuda_default(1920, 1080)
    _setup(gpu, sc, cam)
    cells = pure_cells(gpu.primary_ids()[0], 4)
    for flags in (0, abi.FLAG_NEE_MIS):             # the MIS estimator has the same expectation
        _setup(gpu, sc, cam)
        for s0 in range(0, 1024, 64):
            gpu.render(s0, 64, abi.render_params(flags=flags))
        acc = gpu.accum()
        disp = np.clip(acc[..., :3] / acc[..., 3:4], 0.0, 1.0).astype(np.float64)
        small = disp.reshape(270, 4, 480, 4, 3).mean(axis=(1, 3))
        d = np.abs(small - ref)
        tag = f"screenshot_flags{flags}"
        _bound(tag + "/mae", d.mean(), SCREENSHOT_LIMITS["mae"])
        _bound(tag + "/rmse", np.sqrt((d ** 2).mean()), SCREENSHOT_LIMITS["rmse"])
        _bound(tag + "/max_rel_channel_mean_diff", np.abs(small.mean(axis=(0, 1)) / ref.mean(axis=(0, 1)) - 1.0).max(), SCREENSHOT_LIMITS["channel_mean"])
        _bound(tag + "/mirrored_image_mae", np.abs(small[:, ::-1] - ref).mean(), 0.1, at_least=True)   # negative control
        regions = region_stats(small, ref, cells)
        assert set(SCREENSHOT_REGION_LIMITS) <= set(regions), sorted(regions)
        for name, st in regions.items():
            _bound(f"{tag}/region/{name}/mean_colour_diff", st["mean_diff"], SCREENSHOT_REGION_LIMITS.get(name, 0.05))
        # the lights are white in the reference's render (its kernel ignores makeEmissive's colour)
        for name in ("main light", "blue light", "orange light"):
            assert min(regions[name]["ref_mean"]) > 0.99


# measured on the MI355X in round 2 (profiles/r2_parity_measured.json); limits <= 3 x measured, floor 0.01 for the
# regions (the screenshot itself is a noisy frame of unknown sample count, truncated to 8 bits)
SCREENSHOT_LIMITS = {"mae": 0.02, "rmse": 0.05, "channel_mean": 0.025}     # measured 0.0124, 0.0319, 0.0125
SCREENSHOT_REGION_LIMITS = {                                               # measured (1024 samples, both estimators)
    "floor": 0.01, "left wall (red)": 0.01, "right wall (green)": 0.01,    # 0.0027, 0.0015, 0.0012
    "ceiling": 0.02, "back wall (mirror)": 0.011,                          # 0.0067, 0.0036
    "glass sphere": 0.01, "glossy sphere": 0.01, "metal sphere": 0.0174,   # 0.0025, 0.0025, 0.0058
    "main light": 0.01, "blue light": 0.01, "orange light": 0.01,          # 0.0001, 0, 0.0010
    "tetrahedron": 0.0195}                                                 # 0.0065


# ----------------------------------------------------- committed golden fixtures ---
def test_gpu_against_committed_golden_frames(gpu):
    """The HIP path against tests/golden/oracle_frames.npz (written by tests/golden/make_golden.py from the
    oracle; the reference itself holds no fixtures) — no oracle call in this test."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_frames.npz"))
    sc, cam = scenes.cuda_default(64, 36)
    _setup(gpu, sc, cam)
    assert np.array_equal(gpu.primary_ids()[0], g["cuda_default_ids"])                   # bit-exact
    sc, cam = scenes.cuda_default(48, 27)
    for key, flags in (("cuda_default_path_f32_8spp", 0), ("cuda_default_path_f32_8spp_mis_colour", abi.FLAG_NEE_MIS | abi.FLAG_LIGHT_COLOUR)):
        _setup(gpu, sc, cam)
        gpu.render(0, 8, abi.render_params(flags=flags))
        got, want = gpu.accum(), g[key]
        assert np.array_equal(got[..., 3], want[..., 3])
        # FP32 with and without FMA contraction: a few per cent of the pixels hold a sample whose path branched differently
        _bound(f"golden_{key}/frac_pixels_differing", _frac_differing(_mean(got), _mean(want)), 0.021 if flags else 0.0025)   # measured 0.0069 / 0.00077 (one pixel)
        _bound(f"golden_{key}/rel_mean_diff", abs(_mean(got).mean() - _mean(want).mean()) / _mean(want).mean(), 0.0043 if flags else 0.0104)   # measured 0.0014 / 0.0035
    sc, cam = scenes.cpu_default(64, 36)
    _setup(gpu, sc, cam)
    gpu.render(0, 4, abi.render_params(integrator=abi.INTEGRATOR_DIRECT))
    got, want = gpu.accum(), g["cpu_default_direct_4spp"]
    assert np.array_equal(got[..., 3], want[..., 3])
    # deterministic terms exact; the light term flips on FP rounding, so compare the frame mean
    _bound("golden_cpu_default_direct_4spp/rel_mean_diff", abs(got[..., :3].mean() - want[..., :3].mean()) / want[..., :3].mean(), 0.0212)   # measured 0.0071


# --------------------------------------------- headless driver and file formats (§8f.3) ---
def test_headless_driver_ppm_checkpoint_resume_and_script(tmp_path, oracle):
    """tools/pt_headless (C++ host over the C ABI): P6 output, raw (sum, count) checkpoint, resume, scripted camera."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "pt_headless")
    W, H = 96, 54

    def run(*args):
        out = subprocess.run([exe, "--scene", "cuda_default", "--width", str(W), "--height", str(H), *args],
                             check=True, capture_output=True, text=True, timeout=120).stdout
        return json.loads(out.strip().splitlines()[-1])

    full, half, rest = tmp_path / "full.f32", tmp_path / "half.f32", tmp_path / "rest.f32"
    ppm = tmp_path / "frame.ppm"
    info = run("--spp", "8", "--spp-per-launch", "4", "--raw", str(full), "--out", str(ppm))
    assert info["spp"] == 8 and info["width"] == W
    data = ppm.read_bytes()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header) and len(data) == len(header) + 3 * W * H      # Camera::saveImage layout, Camera.h:162-180
    a = np.fromfile(full, np.float32).reshape(H, W, 4)
    assert np.all(a[..., 3] == 8)
    # the PPM is the gamma-2.2, +0.5-rounded mean of the same buffer (Color.h:86-91)
    img = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    assert np.array_equal(img, oracle.resolve(a, abi.ACC_SUM, abi.RESOLVE_RGB8_PPM).reshape(H, W, 3))   # bit-exact bytes
    # 4 spp, checkpoint, resume for 4 more == 8 spp in one go, bit for bit
    run("--spp", "4", "--spp-per-launch", "4", "--raw", str(half))
    run("--spp", "4", "--spp-per-launch", "4", "--resume", str(half), "--raw", str(rest))
    assert np.array_equal(np.fromfile(rest, np.float32), np.fromfile(full, np.float32))
    # scripted camera: 2 frames at rest, 1 moving (resets), 3 at rest -> 4 frames in the buffer
    script = tmp_path / "moves.txt"
    script.write_text("2 0 0 -\n1 15 -5 wd\n3 0 0 -\n")
    acc = tmp_path / "fly.f32"
    info = run("--script", str(script), "--raw", str(acc), "--out", str(tmp_path / "fly_%02u.ppm"))
    assert info["frames"] == 6 and info["resets"] == 1 and info["frames_since_reset"] == 4
    assert np.all(np.fromfile(acc, np.float32).reshape(H, W, 4)[..., 3] == 4)
    assert all((tmp_path / f"fly_{i:02d}.ppm").exists() for i in range(3))


# ------------------------------------------------------------- native RCCL reduce ---
def test_pt_reduce_on_a_one_rank_communicator(gpu):
    """pt_reduce hands the accumulation buffer to ncclReduce / ncclAllReduce of the RCCL copy loaded in the
    process.  One GPU here, so the communicator has one rank and the sum is the buffer itself; the
    N-rank arithmetic is covered by the 2-rank gloo test on the CPU and by bench.py --gpus N."""
    import glob
    import os
    import torch
    from path_tracer_amd._lib import PtError
    path = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))[0]
    rccl = C.CDLL(path, mode=C.RTLD_GLOBAL)       # the copy torch uses; global, so the library's dlsym sees it

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        sc, cam = scenes.cuda_default(160, 90)
        _setup(gpu, sc, cam)
        gpu.set_frames_in_flight(2)
        for s in range(4):
            gpu.render(s, 1, abi.render_params())
        gpu.reduce(comm.value, root=0)          # folds the chains, then ncclReduce in place
        after = gpu.accum()
        assert np.all(after[..., 3] == 4)
        gpu.reduce(comm.value, root=-1)         # ncclAllReduce
        assert np.array_equal(gpu.accum(), after)
        with pytest.raises(PtError, match="communicator is null"):
            gpu.reduce(None, 0)
        # the reduce-scatter form (one rank: its slice is the whole frame) followed by the slice's resolve
        assert gpu.reduce_scatter(comm.value, 0, 1) == (0, 160 * 90)
        assert np.array_equal(gpu.accum(), after)
        out = torch.zeros(160 * 90, device="cuda", dtype=torch.int32)
        part = torch.zeros(1000, device="cuda", dtype=torch.int32)
        torch.cuda.synchronize()                # (torch clears them on its own stream; the context's is not ordered with it)
        gpu.resolve_range(abi.RESOLVE_RGBA8_CUDA, 0, 160 * 90, out)
        gpu.sync()
        assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(90, 160), gpu.resolve_to_host(abi.RESOLVE_RGBA8_CUDA))
        gpu.resolve_range(abi.RESOLVE_RGBA8_CUDA, 4321, 1000, part)
        gpu.sync()
        assert np.array_equal(part.cpu().numpy().view(np.uint32), gpu.readback(abi.RESOLVE_RGBA8_CUDA).reshape(-1)[4321:5321])
        with pytest.raises(PtError, match="not divisible"):
            gpu.reduce_scatter(comm.value, 0, 7)
        with pytest.raises(PtError, match="exceed the frame"):
            gpu.resolve_range(abi.RESOLVE_RGBA8_CUDA, 160 * 90 - 10, 11, part)
    finally:
        gpu.set_frames_in_flight(1)
        rccl.ncclCommDestroy(comm)


# --------------------------------------- many primitives of every kind (all loop shapes) ---
def _crowded_scene(w, h, ns=100, np_=70, nt=150):
    """More than 64 primitives of each kind: several trips of every batched loop, padding of odd counts,
    and the global-memory leg of the lane-parallel test (its LDS image holds the first 64 of a kind)."""
    from path_tracer_amd.scenes import Dielectric, Emissive, Glossy, Lambertian, Metal, SceneBuilder, _u01
    F = lambda v: float(np.float32(v))
    b = SceneBuilder()
    mats = [b.material(Lambertian((F(0.2 + 0.6 * _u01(i, 0)), F(0.2 + 0.6 * _u01(i, 1)), F(0.2 + 0.6 * _u01(i, 2))))) for i in range(6)]
    mats += [b.material(Metal(F(0.9), F(0.05))), b.material(Dielectric(F(1.5))), b.material(Glossy(F(0.8), F(0.3), F(0.7)))]
    lamp = b.material(Emissive((1, 1, 1), 15.0))
    # a closed box of axis planes, then planes outside it (never the closest, but tested): tilted and axis-aligned
    for pnt, nrm in (((0, -3, 0), (0, 1, 0)), ((0, 5, 0), (0, -1, 0)), ((-6, 0, 0), (1, 0, 0)), ((6, 0, 0), (-1, 0, 0)),
                     ((0, 0, -12), (0, 0, 1)), ((0, 0, 3), (0, 0, -1))):
        b.addObject(b.Plane(pnt, nrm, mats[0]))
    for i in range(np_ - 6):
        k = 20.0 + i
        if i % 3 == 0:
            b.addObject(b.Plane((0, -k, 0), (0, 1, 0), mats[i % 6]))
        else:
            b.addObject(b.Plane((F(-k), F(-k), F(-k)), (F(0.3 + _u01(i, 3)), F(0.4 + _u01(i, 4)), F(0.2 + _u01(i, 5))), mats[i % 6]))
    for i in range(ns):
        c = (F(-5 + 10 * _u01(100 + i, 0)), F(-2.5 + 6 * _u01(100 + i, 1)), F(-11 + 9 * _u01(100 + i, 2)))
        if i % 16 == 0:
            b.addObject(b.Sphere(c, F(0.35), lamp).makeEmissive((1, 1, 1), 15.0))
        else:
            b.addObject(b.Sphere(c, F(0.2 + 0.4 * _u01(100 + i, 3)), mats[i % len(mats)]))
    for i in range(nt):
        c = np.array([-5 + 10 * _u01(300 + i, 0), -2.5 + 6 * _u01(300 + i, 1), -11 + 9 * _u01(300 + i, 2)])
        e1 = np.array([_u01(300 + i, 3), _u01(300 + i, 4), _u01(300 + i, 5)]) - 0.5
        e2 = np.array([_u01(300 + i, 6), _u01(400 + i, 0), _u01(400 + i, 1)]) - 0.5
        v = [tuple(F(x) for x in c), tuple(F(x) for x in c + e1), tuple(F(x) for x in c + e2)]
        b.addObject(b.Triangle(v[0], v[1], v[2], mats[i % len(mats)]))
    cam = abi.make_camera((0, 0.5, 2.5), (0, 0, -5), (0, 1, 0), 70.0, w, h, abi.CAMERA_CUDA)
    return b.flatten(), cam


@pytest.mark.parametrize("flags", [0, abi.FLAG_NEE_MIS])
def test_crowded_scene_ids_and_path(gpu, oracle, flags):
    sc, cam = _crowded_scene(160, 90)
    assert (sc.num_spheres, sc.num_planes, sc.num_triangles) == (100, 70, 150)
    _setup(gpu, sc, cam)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, cam)
    assert np.array_equal(ids, oid) and np.array_equal(mats, omat)
    p = abi.render_params(max_depth=12, flags=flags | abi.FLAG_COUNTERS)
    gpu.reset_counters()
    gpu.render(0, 8, p)
    got = gpu.accum()
    want, oc = oracle.render_path(sc, cam, 0, 8, p, precision=32, counters=True)
    assert np.all(got[..., 3] == 8) and np.isfinite(got).all()
    # how many pixels hold a sample whose path branches differently under FP rounding is a property of the
    # scene and estimator: the float and double builds of the oracle disagree on 0.4 % (6 % with light
    # sampling) of these pixels, and the GPU's FMA-contracted float arithmetic may not do much worse
    w64 = oracle.render_path(sc, cam, 0, 8, p, precision=64)
    rounding = _frac_differing(_mean(want), _mean(w64))
    _bound(f"crowded_flags{flags}/frac_pixels_differing", _frac_differing(_mean(got), _mean(want)), 1.5 * rounding + 0.01)
    _bound(f"crowded_flags{flags}/rel_mean_diff", abs(_mean(got).mean() - _mean(want).mean()) / _mean(want).mean(), 0.0037)   # measured 0.0012 / 0.0008
    g, o = gpu.counters().as_dict(), oc.as_dict()
    for k in ("rays", "hits", "sphere_roots"):
        assert abs(g[k] - o[k]) <= 0.01 * max(o[k], 1), (k, g[k], o[k])


# -------------------------------------------------------------------------- error paths ---
def test_compiler_workaround_canaries(gpu, oracle):
    """Two hipcc 7.2 miscompiles are worked around in the kernels; these checks fail if either comes back.
    (1) pt_path_kernel.hip, as_id(): __builtin_bit_cast applied directly to an element of an ext_vector read element
        0, so an axis-aligned plane stored in the second half of its float4 (p_k, bits(id), p_k, bits(id)) got the
        bits of p_k as its id.  Six axis-aligned emissive walls, each its own colour, two per axis (both halves of
        each list entry are in use): at depth 1 the FP32 path image shows, pixel by pixel, the colour of the wall the
        FP64 id pass names.
    (2) pt_f64_kernels.hip: a nested select over three base pointers picked the wrong base for its third arm, so a
        triangle's material id was read from the plane table.  The material id of every pixel that sees a triangle
        is the material bound to that triangle (also part of test_ids_bit_exact, on the built-in scenes)."""
    b = scenes.SceneBuilder()
    colours = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1)]
    walls = [((-3, 0, 0), (1, 0, 0)), ((3, 0, 0), (-1, 0, 0)), ((0, -2, 0), (0, 1, 0)), ((0, 2, 0), (0, -1, 0)),
             ((0, 0, -5), (0, 0, 1)), ((0, 0, 5), (0, 0, -1))]
    for (pnt, nrm), c in zip(walls, colours):
        b.addObject(b.Plane(pnt, nrm, b.material(scenes.Emissive(c, 1.0))))
    grey = b.material(scenes.Lambertian((0.5, 0.5, 0.5)))
    tri_mat = b.material(scenes.Lambertian((0.25, 0.5, 0.75)))
    b.addObject(b.Sphere((-1.0, -1.0, -2.0), 0.5, grey))
    b.addObject(b.Triangle((0.5, -1.5, -3.0), (2.0, -1.5, -3.0), (1.2, 0.5, -3.5), tri_mat))
    sc = b.flatten()
    cam = abi.make_camera((0.4, 0.3, 4.0), (0.0, 0.0, -1.0), (0, 1, 0), 75.0, 320, 180, abi.CAMERA_CUDA, aspect_ratio=320 / 180)
    _setup(gpu, sc, cam)
    ids, mats = gpu.primary_ids()
    oid, omat = oracle.primary_ids(sc, cam)
    assert np.array_equal(ids, oid) and np.array_equal(mats, omat)
    # (2) every kind of primitive reports the material bound to it
    tri_obj = int(sc.arrays["tri_obj"][0])
    assert (ids == tri_obj).sum() > 200
    assert np.all(mats[ids == tri_obj] == tri_mat)
    assert np.all(mats[ids == int(sc.arrays["sph_obj"][0])] == grey)
    # (1) the wall the FP32 loop picked, read off its emission
    gpu.render(0, 1, abi.render_params(max_depth=1))
    img = gpu.accum()[..., :3]
    wall_obj = [int(o) for o in sc.arrays["pl_obj"]]
    seen = 0
    wrong = 0
    for o, c in zip(wall_obj, colours):
        m = ids == o
        seen += int(m.sum() > 50)
        wrong += int((np.abs(img[m] - np.array(c, np.float32)).max(axis=-1) > 0).sum())
    assert seen >= 5, "the camera must see walls from both halves of the axis lists"
    # (pixels on a wall's edge: the path's sample is jittered inside the pixel, the id pass looks through its centre)
    _bound("canary_axis_plane_ids/frac_pixels_with_another_walls_colour", wrong / ids.size, 0.02)


def test_error_behaviour():
    from path_tracer_amd._lib import PtError, check, lib
    from path_tracer_amd.renderer import Renderer
    L = lib()
    r = Renderer(0)
    sc, cam = scenes.cuda_default(32, 18)
    with pytest.raises(PtError) as e:
        r.render(0, 1, abi.render_params())
    assert e.value.code == abi.PT_ERR_STATE
    r.set_scene(sc)
    with pytest.raises(PtError) as e:
        r.render(0, 1, abi.render_params())
    assert e.value.code == abi.PT_ERR_STATE
    r.set_camera(cam)
    r.render(0, 0, abi.render_params())  # zero samples: a no-op
    with pytest.raises(PtError) as e:
        r.render(0, 2, abi.render_params(acc_mode=abi.ACC_RUNNING_MEAN))
    assert e.value.code == abi.PT_ERR_INVALID
    with pytest.raises(PtError):
        r.render(0, 1, abi.render_params(max_depth=0))
    with pytest.raises(PtError):
        r.render(0xFFFFFFFF, 2, abi.render_params())
    bad = scenes.cuda_default(32, 18)[0]
    bad.arrays["sph_mat"][0] = 99
    with pytest.raises(PtError) as e:
        r.set_scene(bad)
    assert e.value.code == abi.PT_ERR_INVALID
    with pytest.raises(PtError) as e:
        r.set_scene(abi.SceneArrays())
    assert e.value.code == abi.PT_ERR_INVALID
    cpu_cam = scenes.cpu_default(32, 18)[1]
    r.set_camera(cpu_cam)
    with pytest.raises(PtError):
        r.render(0, 1, abi.render_params())  # the path integrator needs the CUDA camera model
    with pytest.raises(PtError):
        Renderer(99)
    # pt_resize without a new camera: the id pass and the render refuse (the kernels iterate the camera's pixels)
    r.set_camera(cam)
    check(L.pt_resize(r._ctx, 64, 64))
    r.width, r.height = 64, 64
    for call in (lambda: r.primary_ids(), lambda: r.render(0, 1, abi.render_params())):
        with pytest.raises(PtError) as e:
            call()
        assert e.value.code == abi.PT_ERR_STATE and "resized" in str(e.value)
    r.set_camera(scenes.cuda_default(64, 64)[1])
    assert r.primary_ids()[0].shape == (64, 64)
    # the error message is the thread's LAST call's: a failure, then a good launchRenderKernel (which returns void,
    # like the reference's; INTEGRATION.md tests pt_last_error() after it) leaves it empty
    import torch
    out = torch.zeros((64, 64, 4), device="cuda", dtype=torch.float32)
    assert L.pt_set_frames_in_flight(r._ctx, 99) == abi.PT_ERR_INVALID and L.pt_last_error()
    r.launch_render_kernel(out.data_ptr(), 0)
    assert L.pt_last_error() == b""
    r.sync()
    assert float(out[..., 3].min()) == 1.0
    # ... and a launch that fails says so through the same channel
    check(L.pt_make_current(r._ctx))
    L.launchRenderKernel(C.c_void_p(out.data_ptr()), 32, 32, 0, abi.PtDim3(2, 2, 1), abi.PtDim3(16, 16, 1))
    assert b"does not match" in L.pt_last_error()
    r.close()


def test_frames_in_flight_are_capped_by_the_hardware_queues():
    """Chains that share a hardware queue serialise, so pt_set_frames_in_flight grants at most what
    GPU_MAX_HW_QUEUES leaves room for: this process runs with 16 (tests/conftest.py) and gets all 8; a process
    that leaves the variable unset gets 3 of the 8 it asks for, and renders the same samples."""
    import subprocess
    import sys
    from path_tracer_amd.renderer import Renderer
    r = Renderer(0)
    try:
        r.set_frames_in_flight(8)
        assert r.frames_in_flight() == 8
        r.set_frames_in_flight(2)
        assert r.frames_in_flight() == 2
    finally:
        r.close()
    code = (
        "import numpy as np\n"
        "from path_tracer_amd import abi, scenes\n"
        "from path_tracer_amd.renderer import Renderer\n"
        "r = Renderer(0); sc, cam = scenes.cuda_default(160, 90); r.set_scene(sc); r.set_camera(cam)\n"
        "for s in range(7): r.render(s, 1, abi.render_params())\n"
        "serial = r.accum(); r.reset()\n"
        "r.set_frames_in_flight(8); assert r.frames_in_flight() == 3, r.frames_in_flight()\n"
        "r.set_frames_in_flight(2); assert r.frames_in_flight() == 2\n"
        "r.set_frames_in_flight(8)\n"
        "for s in range(7): r.render(s, 1, abi.render_params())\n"
        "got = r.accum(); assert np.all(got[..., 3] == 7); assert np.array_equal(got, serial)\n"
        "r.close(); print('capped at 3')\n")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "capped at 3" in p.stdout, p.stdout + p.stderr[-1500:]


# ------------------------------------------------------------- full-size property checks ---
def test_full_size_720p_properties(gpu):
    """At BASELINE's size the oracle is too slow to referee every pixel; check size-independent
    properties instead: determinism, sample-count channel, energy bounds, split additivity."""
    sc, cam = scenes.cuda_default(1280, 720)
    p = abi.render_params()
    _setup(gpu, sc, cam)
    for s in range(4):
        gpu.render(s, 1, p)
    a = gpu.accum()
    _setup(gpu, sc, cam)
    gpu.render(0, 4, abi.render_params(pixels_per_wave=512))
    b = gpu.accum()
    assert np.all(a[..., 3] == 4) and np.isfinite(a).all() and a.min() >= 0.0
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-5)
    m = _mean(a)
    assert 0.2 < m.mean() < 0.8          # the oracle's converged mean is ~0.38 (lights 15, albedo <= 0.98)
    assert m.max() <= 15.0 * 1.0001      # nothing brighter than an emitter seen directly


def test_full_size_4k_properties(gpu):
    """BASELINE config 3's frame (3840x2160).  The oracle cannot referee 8.3 M pixels of paths in a test, so: the
    primary ids are bit-exact at this size (test_ids_bit_exact), and here the size-independent properties hold —
    every pixel counted once per sample, the image independent of how the samples are cut into launches, the
    sample split across 'GPUs' additive, and the 4K image box-filtered 3x the same picture as the 720p one."""
    sc, cam = scenes.cuda_default(3840, 2160)
    p = abi.render_params()
    _setup(gpu, sc, cam)
    gpu.render(0, 4, p)
    a = gpu.accum()
    assert np.all(a[..., 3] == 4) and np.isfinite(a).all() and a.min() >= 0.0
    assert a[..., :3].max() <= 4 * 15.0 * 1.0001            # nothing brighter than an emitter seen directly
    _setup(gpu, sc, cam)
    gpu.render(0, 2, abi.render_params(pixels_per_wave=256))   # "GPU 0" of a two-way split, another launch shape
    half = gpu.accum()
    _setup(gpu, sc, cam)
    gpu.render(2, 1, p)
    gpu.render(3, 1, p)                                        # "GPU 1", progressive
    other = gpu.accum()
    np.testing.assert_allclose(half + other, a, rtol=2e-5, atol=1e-5)
    m4k = _mean(a).reshape(720, 3, 1280, 3, 3).mean(axis=(1, 3))
    sc7, cam7 = scenes.cuda_default(1280, 720)
    _setup(gpu, sc7, cam7)
    gpu.render(0, 36, p)                                       # as many samples per 720p pixel as its 3x3 4K pixels hold
    m720 = _mean(gpu.accum())
    _bound("full_size_4k/rel_mean_vs_720p", abs(m4k.mean() - m720.mean()) / m720.mean(), 0.0042)   # measured 0.0014
    # both are 36-sample estimates of (nearly) the same pixel footprint: the difference is Monte-Carlo noise
    coarse = lambda m: np.clip(m, 0, 1).reshape(45, 16, 80, 16, 3).mean(axis=(1, 3))
    _bound("full_size_4k/coarse_mae_vs_720p", np.abs(coarse(m4k) - coarse(m720)).mean(), 0.03)   # measured 0.0117
    # frames in flight at this size: a launch holds 16 frames (2^27 pixel-frames), its ticket space a million units --
    # the largest dividends of the kernel's float-quotient divisions -- and 19 frames are two launches.  The image and
    # the event counters are the serial loop's, bit for bit.
    q = abi.render_params(max_depth=6, fused_rgba8=1, flags=abi.FLAG_COUNTERS)
    _setup(gpu, sc, cam)
    gpu.reset_counters()
    for f in range(19):
        gpu.render(f, 1, q)
    serial, serial_rgba, serial_counters = gpu.accum(), gpu.readback(abi.RESOLVE_RGBA8_CUDA), gpu.counters().as_dict()
    gpu.set_frames_in_flight(3)
    try:
        _setup(gpu, sc, cam)
        gpu.reset_counters()
        before = gpu.path_launches()
        for f in range(19):
            gpu.render(f, 1, q)
        assert np.array_equal(gpu.accum(), serial)
        assert gpu.path_launches() - before == 2
        assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), serial_rgba)
        assert gpu.counters().as_dict() == serial_counters
    finally:
        gpu.set_frames_in_flight(1)
    small = scenes.cuda_default(64, 36)
    gpu.set_scene(small[0])
    gpu.set_camera(small[1])


def test_bench_line_contract():
    """bench.py prints one JSON line with the contract's keys, the roofline and the CPU baseline."""
    import json
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "48", "--warmup", "8", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-800:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 48 and d["warmup"] == 8 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["config"]["workload"] == "cornell_720p_progressive_1spp" and d["dtype"] == "f32"
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1 and r["unit"] == "TFLOP/s"
    assert abs(d["value"] - 1280 * 720 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    # the same frames one at a time: strictly serial launches, and the library's default for that loop (the next frame
    # rendered ahead), which must not be slower; both below the headline's frames in flight
    assert 0 < r["frame_ms_render_ahead"] < r["kernel_ms_serial"] * 1.02 and r["frac_serial"] < r["frac_render_ahead"] * 1.02
    assert r["frame_ms"] < r["frame_ms_render_ahead"]
    # the 4K leg (BASELINE config 3 at N = 1: 64 steps of 16 spp = 1024 spp)
    l = d["legs"]["cornell_4k_1024spp_strong_split"]
    assert l["spp_total"] == 1024 and l["width"] == 3840 and l["value"] > 0 and 0 < l["roofline"]["frac"] < 1
    assert abs(l["value"] - 3840 * 2160 * 1024 / (l["wall_ms"] * 1e-3) / 1e6) < 0.01 * l["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"]
    # the FP64 `direct` leg (BASELINE config 1: cpu_default, 1280x720, 64 spp per step), with the oracle's renderRegion beside it
    g = d["legs"]["cpu_default_direct_720p_64spp"]
    assert g["baseline_config"] == 1 and g["dtype"] == "f64" and g["spp_per_step"] == 64 and g["value"] > 0
    assert g["roofline"]["bound"] == "valu_fp64" and 0 < g["roofline"]["frac"] < 1
    assert abs(g["value"] - 1280 * 720 * 64 / (g["ms_per_step"] * 1e-3) / 1e6) < 0.01 * g["value"]
    assert 1.0 <= g["rays_per_sample"] <= 2.0      # a primary ray and at most one shadow ray (one light)
    assert g["cpu_baseline"]["kind"] == "port" and g["cpu_baseline"]["value"] > 0
