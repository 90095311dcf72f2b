"""pt_render_noalloc (include/pt.h): the entry point for callers that cannot allocate — HIP graph capture, serving
loops, harnesses that forbid hipMalloc.  Called through the C ABI and checked against the existing entry point,
pt_render: same kernel, same launch shape, so the float4 buffers must be bit-identical (a call that pt_render cuts
into several launches is compared up to float summation order, tolerance written below).  Two calls in a row on one
GPU, the second with a smaller input: the static scratch array keeps the first call's contents.

Collected last on purpose (the round's end runs `pytest -x`).
"""
import ctypes as C

import numpy as np
import pytest

from path_tracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


def test_no_alloc_entry_point_matches_pt_render():
    import torch
    from path_tracer_amd._lib import PtError, lib
    from path_tracer_amd.renderer import Renderer
    L = lib()
    r = Renderer(0)
    try:
        p = abi.render_params()
        # every buffer the calls below use exists before the first of them
        big = torch.zeros((180, 320, 4), device="cuda", dtype=torch.float32)
        small = torch.zeros((54, 96, 4), device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()

        # ---- first call: 320x180, 8 samples --------------------------------------------------------------
        sc, cam = scenes.cuda_default(320, 180)
        r.set_scene(sc)
        r.set_camera(cam)
        r.render(0, 8, p)                       # the existing entry point (it cuts the call into 8 one-sample frames of one launch and
                                                #  adds them to a zero buffer in sample order: the same float adds as one 8-sample launch)
        want_big = r.accum()
        free_before = torch.cuda.mem_get_info()[0]
        r.render_noalloc(big.data_ptr(), 0, 8, p)
        r.sync()
        assert torch.cuda.mem_get_info()[0] == free_before          # no device memory was taken
        assert np.array_equal(big.cpu().numpy(), want_big)           # bit-identical

        # ---- second call in a row, smaller input: 96x54, 3 samples, another scene, light sampling on ------
        sc2, cam2 = scenes.glass_stress(96, 54)
        p2 = abi.render_params(max_depth=32, flags=abi.FLAG_NEE_MIS | abi.FLAG_LIGHT_COLOUR)
        r.set_scene(sc2)
        r.set_camera(cam2)
        r.render(0, 3, p2)
        want_small = r.accum()
        r.render_noalloc(small.data_ptr(), 0, 3, p2)
        r.sync()
        assert np.array_equal(small.cpu().numpy(), want_small)
        # ... and it accumulates onto what the buffer holds, like pt_render on its own buffer
        r.render(3, 2, p2)
        r.render_noalloc(small.data_ptr(), 3, 2, p2)
        r.sync()
        assert np.array_equal(small.cpu().numpy(), r.accum())
        assert np.all(small.cpu().numpy()[..., 3] == 5)

        # ---- running mean (the reference's launchRenderKernel semantics) and a fused display ---------------
        pm = abi.render_params(max_depth=32, acc_mode=abi.ACC_RUNNING_MEAN, fused_rgba8=1)
        r.reset()
        small.zero_()
        torch.cuda.synchronize()
        for f in range(3):
            r.render(f, 1, pm)
        want_mean, want_rgba = r.accum(), r.readback(abi.RESOLVE_RGBA8_CUDA)
        for f in range(3):
            r.render_noalloc(small.data_ptr(), f, 1, pm)
        r.sync()
        assert np.array_equal(small.cpu().numpy(), want_mean)
        assert np.array_equal(r.readback(abi.RESOLVE_RGBA8_CUDA), want_rgba)

        # ---- a call that pt_render cuts into launches on its chains: equal up to float summation order ------
        r.set_scene(sc)
        r.set_camera(cam)
        r.render(0, 40, p)
        want40 = r.accum()
        big.zero_()
        torch.cuda.synchronize()
        r.render_noalloc(big.data_ptr(), 0, 40, p)
        r.sync()
        got40 = big.cpu().numpy()
        assert np.array_equal(got40[..., 3], want40[..., 3])
        np.testing.assert_allclose(got40[..., :3], want40[..., :3], rtol=1e-5, atol=1e-5)   # 40 float adds in another order

        # ---- limits and refusals, each with a clear message ---------------------------------------------------
        def refused(code, text, *args):
            rc = L.pt_render_noalloc(r._ctx, *args)
            assert rc == code and text in L.pt_last_error().decode(), (rc, L.pt_last_error())
        ptr = C.c_void_p(big.data_ptr())
        refused(abi.PT_ERR_CAPACITY, "exceeds the maximum", ptr, 16384, 8192, 0, 1, C.byref(p))      # > 2^26 pixels
        refused(abi.PT_ERR_CAPACITY, "samples exceed the maximum", ptr, 320, 180, 0, 4097, C.byref(p))
        refused(abi.PT_ERR_INVALID, "does not match the camera", ptr, 96, 54, 0, 1, C.byref(p))
        refused(abi.PT_ERR_INVALID, "16-byte aligned", C.c_void_p(big.data_ptr() + 4), 320, 180, 0, 1, C.byref(p))
        refused(abi.PT_ERR_INVALID, "16-byte aligned", None, 320, 180, 0, 1, C.byref(p))
        r.set_frames_in_flight(2)
        r.render(40, 1, p)                      # opens a burst
        refused(abi.PT_ERR_STATE, "pt_flush first", ptr, 320, 180, 0, 1, C.byref(p))
        r.flush()
        r.render_noalloc(big.data_ptr(), 40, 1, p)
        r.sync()
        assert np.all(big.cpu().numpy()[..., 3] == 41)
        # frames that launchRenderKernel left in flight are refused the same way, before anything is folded or allocated
        out = torch.zeros((180, 320, 4), device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        launches = r.path_launches()
        r.launch_render_kernel(out.data_ptr(), 0)
        r.launch_render_kernel(out.data_ptr(), 1)          # two frames held back on `out`
        refused(abi.PT_ERR_STATE, "pt_flush first", ptr, 320, 180, 0, 1, C.byref(p))
        assert r.path_launches() == launches                # the refusal launched nothing
        r.flush()
        r.sync()
        assert np.all(out.cpu().numpy()[..., 3] == 1.0)     # the running mean of the two frames arrived in `out`
        r.set_frames_in_flight(1)
    finally:
        r.close()
