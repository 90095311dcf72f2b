"""Frames in flight, second set (round 3): the reference's running mean inside multi-frame launches -- through pt_render
and through launchRenderKernel itself --, the chains' buffers sized by what a launch holds, the serial fall-back when
they cannot be allocated, frames with a side beyond 65 536 pixels, and the display history across a resize.
Everything through the C ABI; the bar is bit-identity with the one-launch-per-frame loop."""
import numpy as np
import pytest

from path_tracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from path_tracer_amd.renderer import Renderer
    r = Renderer(0)
    yield r
    r.close()


def _setup(r, scene, cam):
    r.set_scene(scene)
    r.set_camera(cam)
    r.reset()


@pytest.mark.parametrize("n", [2, 3])
def test_running_mean_frames_in_flight_match_serial(gpu, n):
    """PT_ACC_RUNNING_MEAN frames (renderKernel's c*t + prev*(1-t), t = 1/(frame_count+1), raytracer_kernel.cu:297-304)
    are collected into multi-frame launches like sums: the fold pass applies the recurrence frame after frame, so the
    buffer and every frame's display are the serial loop's bit for bit."""
    sc, cam = scenes.cuda_default(192, 108)
    p = abi.render_params(acc_mode=abi.ACC_RUNNING_MEAN, fused_rgba8=1)
    frames = 70   # one launch of 64 frames and one of 6
    _setup(gpu, sc, cam)
    serial_rgba = []
    for f in range(frames):
        gpu.render(f, 1, p)
        serial_rgba.append(gpu.readback(abi.RESOLVE_RGBA8_CUDA))
    serial = gpu.accum()
    assert np.all(serial[..., 3] == 1.0)
    gpu.set_frames_in_flight(n)
    try:
        _setup(gpu, sc, cam)
        before = gpu.path_launches()
        for f in range(frames):
            gpu.render(f, 1, p)
        got = gpu.accum()
        assert gpu.path_launches() - before == 2
        assert np.array_equal(got, serial)
        assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), serial_rgba[-1])
        assert gpu.history_frames() == 5
        for back in range(1, 6):
            assert np.array_equal(gpu.readback_history(back), serial_rgba[frames - 1 - back]), back
        # frame_count == 0 in the middle of a run overwrites (the reference's only "reset"), also inside a launch
        _setup(gpu, sc, cam)
        q = abi.render_params(acc_mode=abi.ACC_RUNNING_MEAN)
        for f in (0, 1, 2, 0, 1, 2, 3):
            gpu.render(f, 1, q)
        got = gpu.accum()
        gpu.set_frames_in_flight(1)
        _setup(gpu, sc, cam)
        for f in (0, 1, 2, 0, 1, 2, 3):
            gpu.render(f, 1, q)
        assert np.array_equal(got, gpu.accum())
        # sums and means alternate on one buffer: every change of mode starts a new launch
        gpu.set_frames_in_flight(n)
        _setup(gpu, sc, cam)
        for f in range(3):
            gpu.render(f, 1, q)
        for f in range(3, 6):
            gpu.render(f, 1, abi.render_params())
        got = gpu.accum()
        gpu.set_frames_in_flight(1)
        _setup(gpu, sc, cam)
        for f in range(3):
            gpu.render(f, 1, q)
        for f in range(3, 6):
            gpu.render(f, 1, abi.render_params())
        assert np.array_equal(got, gpu.accum())
    finally:
        gpu.set_frames_in_flight(1)


def test_launch_render_kernel_collects_frames(gpu):
    """The drop-in call itself: with frames in flight, consecutive launchRenderKernel calls on one `output` that the
    caller does not read in between leave as ONE launch of the path kernel; after pt_flush `output` is the serial
    loop's buffer bit for bit.  A caller that reads between launches (the reference's loop) flushes, and gets serial
    launches."""
    import torch
    sc, cam = scenes.cuda_default(160, 90)
    frames = 9
    _setup(gpu, sc, cam)
    out = torch.full((90, 160, 4), 7.0, device="cuda", dtype=torch.float32)   # frame_count == 0 overwrites
    torch.cuda.synchronize()
    for f in range(frames):
        gpu.launch_render_kernel(out.data_ptr(), f)
    gpu.sync()
    serial = out.cpu().numpy()
    gpu.set_frames_in_flight(2)
    try:
        out2 = torch.full((90, 160, 4), 7.0, device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        before = gpu.path_launches()
        for f in range(frames):
            gpu.launch_render_kernel(out2.data_ptr(), f)
        assert gpu.path_launches() == before          # all held back
        gpu.flush()                                   # launches and folds them into `output`, on the context's stream
        gpu.sync()
        assert gpu.path_launches() - before == 1
        assert np.array_equal(out2.cpu().numpy(), serial)
        # the context's own buffer is bound again: the next pt_render does not touch `output`
        gpu.reset()
        gpu.render(0, 1, abi.render_params())
        assert np.all(gpu.accum()[..., 3] == 1)
        torch.cuda.synchronize()
        assert np.array_equal(out2.cpu().numpy(), serial)
        # two outputs in turn: a call that names another buffer first folds what the old one is owed
        out3 = torch.zeros_like(out2)
        out4 = torch.zeros_like(out2)
        torch.cuda.synchronize()
        for f in range(4):
            gpu.launch_render_kernel(out3.data_ptr(), f)
        for f in range(frames):
            gpu.launch_render_kernel(out4.data_ptr(), f)
        gpu.sync()
        assert np.array_equal(out4.cpu().numpy(), serial)
        assert np.all(out3.cpu().numpy()[..., 3] == 1.0) and not np.array_equal(out3.cpu().numpy(), serial)
    finally:
        gpu.set_frames_in_flight(1)


@pytest.mark.parametrize("size", [(65537, 2), (3, 70001)])
def test_frames_with_a_side_beyond_65536(gpu, size):
    """alloc_frame admits any frame of up to 2^28 pixels; a lane's (x, y) of the next ray are packed with as many bits
    as width and height need, for one-frame launches too (16 + 16 used to alias pixels of such frames)."""
    sc, cam = scenes.cuda_default(*size)
    q = abi.render_params(max_depth=4, fused_rgba8=1)
    _setup(gpu, sc, cam)
    for s in range(3):
        gpu.render(s, 1, q)                       # serial one-frame launches
    want = gpu.accum()
    assert np.all(want[..., 3] == 3) and np.isfinite(want).all()
    # every pixel got its own samples (aliased pixels would hold other pixels' samples too, and leave some untouched);
    # and the multi-frame launch, which always packed by need, gives the same image
    gpu.set_frames_in_flight(2)
    try:
        _setup(gpu, sc, cam)
        for s in range(3):
            gpu.render(s, 1, q)
        assert np.array_equal(gpu.accum(), want)
    finally:
        gpu.set_frames_in_flight(1)
    _setup(gpu, *scenes.cuda_default(64, 36))


def test_display_history_does_not_survive_a_resize(gpu):
    from path_tracer_amd._lib import PtError
    sc, cam = scenes.cuda_default(96, 54)
    _, cam_big = scenes.cuda_default(320, 180)
    p = abi.render_params(fused_rgba8=1)
    gpu.set_frames_in_flight(2)
    try:
        _setup(gpu, sc, cam)
        for s in range(6):
            gpu.render(s, 1, p)
        assert gpu.history_frames() == 5
        gpu.set_camera(cam_big)                   # a larger frame: the history planes were sized for the old one
        assert gpu.history_frames() == 0
        with pytest.raises(PtError, match="not held"):
            gpu.readback_history(1)
        for s in range(4):
            gpu.render(s, 1, p)
        assert gpu.history_frames() == 3
        assert gpu.readback_history(3).shape == (180, 320)
        gpu.set_camera(cam)                       # and a smaller one
        assert gpu.history_frames() == 0
        with pytest.raises(PtError, match="not held"):
            gpu.readback_history(1)
    finally:
        gpu.set_frames_in_flight(1)


def test_chain_buffers_are_sized_by_the_launch():
    """A plain pt_render(ctx, 0, 16, ..) on a default context is one launch of 16 frames on one chain: it takes 16 planes
    (236 MB at 1280x720), not 64 planes on each of three chains (2.8 GB)."""
    import torch
    from path_tracer_amd.renderer import Renderer
    sc, cam = scenes.cuda_default(1280, 720)
    r = Renderer(0)
    try:
        _setup(r, sc, cam)
        r.render(0, 1, abi.render_params())
        r.sync()
        free0 = torch.cuda.mem_get_info()[0]
        r.render(1, 16, abi.render_params(fused_rgba8=1))
        r.sync()
        taken = free0 - torch.cuda.mem_get_info()[0]
        plane = 1280 * 720 * 16
        assert taken <= 16 * plane + 15 * 1280 * 720 * 4 + (64 << 20), taken     # 16 planes + 15 history planes + allocator slack
        assert np.all(r.accum()[..., 3] == 17)
    finally:
        r.close()


def test_serial_fallback_when_the_chain_buffers_cannot_be_allocated(monkeypatch):
    """No room for the planes of an overlapped launch: the frames are rendered by serial launches instead of being
    dropped with PT_ERR_HIP -- same sums, same displays."""
    from path_tracer_amd.renderer import Renderer
    sc, cam = scenes.cuda_default(160, 90)
    p = abi.render_params(fused_rgba8=1)
    want = want_rgba = None
    for fail in ("0", "1"):
        monkeypatch.setenv("PT_AMD_FAIL_CHAIN_ALLOC", fail)
        r = Renderer(0)
        try:
            _setup(r, sc, cam)
            r.set_frames_in_flight(2)
            before = r.path_launches()
            for s in range(7):
                r.render(s, 1, p)
            r.render(7, 40, abi.render_params())        # a large call too (cut into frames)
            acc, rgba = r.accum(), None
            launches = r.path_launches() - before
            if fail == "0":
                want = acc
                assert launches == 2
            else:
                assert launches == 7 + 20              # one per frame: 7 frames, then 20 frames of 2 samples
                assert np.array_equal(acc, want)
        finally:
            r.close()
    monkeypatch.delenv("PT_AMD_FAIL_CHAIN_ALLOC")


def test_path_length_histogram_matches_oracle(gpu, oracle):
    """pt_path_length_histogram (the bounce histogram of SURVEY §8(d).4): samples by the number of rays their path
    traced, against the oracle's on the same samples, and consistent with the event counters."""
    for name, depth, flags in (("cuda_default", 50, 0), ("glass_stress", 32, 0), ("cuda_default", 50, abi.FLAG_NEE_MIS)):
        sc, cam = scenes.builtin(name, 160, 90)
        p = abi.render_params(max_depth=depth, flags=flags | abi.FLAG_COUNTERS)
        _setup(gpu, sc, cam)
        gpu.reset_counters()
        gpu.render(0, 2, p)
        gpu.render(2, 1, p)
        hist = gpu.path_length_histogram().astype(np.int64)
        c = gpu.counters().as_dict()
        assert hist.sum() == c["samples"] == 160 * 90 * 3
        assert (hist * np.arange(64)).sum() == c["rays"] - c["shadow_rays"]
        assert hist[0] == 0 and not hist[depth + 1:].any()
        want = oracle.path_length_histogram(sc, cam, 0, 3, p).astype(np.int64)
        # a path in a few thousand branches differently under FP32 rounding (FMA contraction on the GPU): it moves from one
        # bin to another
        moved = np.abs(hist - want).sum() / 2
        assert moved <= 0.003 * hist.sum() + 8, (name, flags, moved)
        gpu.reset_counters()
        assert not gpu.path_length_histogram().any()


def test_direct_integrator_counters(gpu):
    """The instrumented build of the FP64 `direct` kernel (PT_FLAG_COUNTERS): samples exact, one primary ray per sample
    plus one shadow ray per light sample with a positive cosine, one light sample per light for every hit that is not the
    light itself -- and the same image as the plain build."""
    sc, cam = scenes.cpu_default(320, 180)
    p = abi.render_params(integrator=abi.INTEGRATOR_DIRECT)
    _setup(gpu, sc, cam)
    gpu.render(0, 4, p)
    plain = gpu.accum()
    _setup(gpu, sc, cam)
    gpu.reset_counters()
    gpu.render(0, 4, abi.render_params(integrator=abi.INTEGRATOR_DIRECT, flags=abi.FLAG_COUNTERS))
    assert np.array_equal(gpu.accum(), plain)
    c = gpu.counters().as_dict()
    n = 320 * 180 * 4
    assert c["samples"] == n and c["hits"] == n                  # cpu_default's camera sees no background
    assert c["rays"] == n + c["shadow_rays"]
    assert 0 < c["shadow_rays"] <= c["nee_samples"] <= n          # one light: at most one light sample per sample
    ids, _ = gpu.primary_ids()
    light_share = float((ids == 7).mean())                       # the emissive sphere is object 7
    assert abs(c["nee_samples"] / n - (1.0 - light_share)) < 0.01


def test_bench_two_rank_rehearsal_on_one_gpu():
    """bench.py's N > 1 code path end to end -- schedules, reduce-scatter, per-rank resolve of the own slice
    (pt_resolve_range), all-gather of the pictures, the per-rank sample-count check -- with two gloo ranks that share this
    box's one GPU (the numbers mean nothing; the RCCL run on separate GPUs is the driver's)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--cpu-seconds", "0", "--rehearse-on-one-gpu"], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["reduce_bytes_per_rank"] == 1280 * 720 // 2 * 16 and d["gather_bytes_per_rank"] == 1280 * 720 // 2 * 4
    assert d["reduce_ms"] > 0 and d["display_ms"] > 0
    # the three parts follow one another on rank 0's stream inside the timed region: together they cannot exceed it
    assert d["host_and_barrier_ms"] >= 0, d
    l = d["legs"]["cornell_4k_1024spp_strong_split"]
    assert l["scaling"] == "strong" and l["n_gpus"] == 2 and l["steps_per_rank_max"] == 32 and l["reduce_bytes_per_rank"] == 3840 * 2160 // 2 * 16
    assert l["render_ms"] + l["reduce_ms"] + l["display_ms"] <= l["wall_ms"] * 1.001, l
    assert "cpu_default_direct_720p_64spp" not in d["legs"]      # config 1 is a single-GPU leg


def test_random_mixes_of_sum_and_mean_frames_equal_the_serial_loop(gpu):
    """Any sequence of sum frames, running-mean frames (frame counts that continue, restart at 0 or jump), flushes,
    resolves and changes of the number of chains gives, bit for bit, the buffer and the display that the same calls give
    with one serial launch per frame."""
    sc, cam = scenes.cuda_default(96, 54)
    for seed in (1, 2, 3):
        rng = np.random.default_rng(20261111 + seed)
        ops, sample, frame = [], 0, 0
        for _ in range(120):
            k = int(rng.integers(0, 10))
            if k <= 3:
                n = int(rng.integers(1, 4)); ops.append(("sum", sample, n, int(rng.integers(0, 2)))); sample += n
            elif k <= 6:
                if rng.integers(0, 8) == 0: frame = int(rng.integers(0, 3)) * 5      # restart (0 overwrites) or jump
                ops.append(("mean", frame, 1, int(rng.integers(0, 2)))); frame += 1
            elif k == 7: ops.append(("flush",))
            elif k == 8: ops.append(("resolve",))
            else: ops.append(("flight", int(rng.integers(1, 4))))

        def run(pipelined):
            _setup(gpu, sc, cam)
            gpu.set_frames_in_flight(1)
            for op in ops:
                if op[0] == "sum": gpu.render(op[1], op[2], abi.render_params(max_depth=8, fused_rgba8=op[3]))
                elif op[0] == "mean": gpu.render(op[1], 1, abi.render_params(max_depth=8, fused_rgba8=op[3], acc_mode=abi.ACC_RUNNING_MEAN))
                elif op[0] == "flush": gpu.flush()
                elif op[0] == "resolve": gpu.resolve(abi.RESOLVE_RGBA8_CUDA)
                elif pipelined: gpu.set_frames_in_flight(op[1])
            acc = gpu.accum()
            return acc, gpu.resolve_to_host(abi.RESOLVE_RGBA8_CUDA)
        try:
            want, want_rgba = run(False)
            got, got_rgba = run(True)
        finally:
            gpu.set_frames_in_flight(1)
        assert np.isfinite(want).all()
        assert np.array_equal(got, want), seed
        assert np.array_equal(got_rgba, want_rgba), seed


def test_render_ahead_gives_the_serial_loops_bits(gpu):
    """Render-ahead (pt_set_render_ahead, on by default with one frame in flight): the frame the loop will ask for next is
    launched before it is asked for.  Buffer and display after every frame are those of strictly serial launches, for
    sums and for the running mean, through pt_render and through launchRenderKernel, whether the host reads between
    the frames or not; a call that does not continue the loop drops the plane and renders as usual."""
    import torch
    sc, cam = scenes.cuda_default(192, 108)
    for p in (abi.render_params(fused_rgba8=1), abi.render_params(acc_mode=abi.ACC_RUNNING_MEAN, fused_rgba8=1), abi.render_params(max_depth=8)):
        spp = 1 if p.acc_mode == abi.ACC_RUNNING_MEAN else 2
        # what the loop asks for: 6 frames in a row, a jump back, 4 more, another frame size of samples, 3 more
        calls = [(k * spp, spp) for k in range(6)] + [(2 * spp, spp)] + [((7 + k) * spp, spp) for k in range(4)]
        if spp == 2: calls += [(40, 1), (41, 1), (42, 1), (43, 1)]
        gpu.set_render_ahead(False)
        _setup(gpu, sc, cam)
        want = []
        for s0, n in calls:
            gpu.render(s0, n, p)
            want.append((gpu.accum(), gpu.readback(abi.RESOLVE_RGBA8_CUDA) if p.fused_rgba8 else None))
        for read_between in (True, False):
            gpu.set_render_ahead(True)
            _setup(gpu, sc, cam)
            launched0, used0 = gpu.render_ahead_stats()
            launches0 = gpu.path_launches()
            for k, (s0, n) in enumerate(calls):
                gpu.render(s0, n, p)
                if read_between:
                    assert np.array_equal(gpu.accum(), want[k][0]), (k, s0)
                    if p.fused_rgba8: assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), want[k][1]), (k, s0)
            assert np.array_equal(gpu.accum(), want[-1][0])
            if p.fused_rgba8: assert np.array_equal(gpu.readback(abi.RESOLVE_RGBA8_CUDA), want[-1][1])
            launched, used = gpu.render_ahead_stats()
            assert gpu.path_launches() - launches0 == len(calls)          # one launch counted per frame, as before
            assert used - used0 >= 4 and launched - launched0 > used - used0   # frames came from planes; the jumps wasted theirs
    # a camera change drops the frame rendered ahead for the old camera
    gpu.set_render_ahead(True)
    p = abi.render_params(fused_rgba8=1)
    _setup(gpu, sc, cam)
    for k in range(4): gpu.render(k, 1, p)
    cam2 = abi.make_camera((0.3, 1.0, 3.2), (0, 1, 0), (0, 1, 0), 40.0, 192, 108, abi.CAMERA_CUDA)
    gpu.set_camera(cam2); gpu.reset()
    for k in range(4, 8): gpu.render(k, 1, p)          # sample 4 was rendered ahead with the old camera: must not be used
    got = gpu.accum()
    gpu.set_render_ahead(False)
    gpu.set_camera(cam2); gpu.reset()
    for k in range(4, 8): gpu.render(k, 1, p)
    assert np.array_equal(got, gpu.accum())
    # a frame of more than 2^22 pixels, where a call of any sample count is one launch (and is rendered ahead as such)
    big_sc, big_cam = scenes.cuda_default(2100, 2000)
    q = abi.render_params(max_depth=6, fused_rgba8=1)
    got = {}
    for on in (False, True):
        gpu.set_render_ahead(on)
        _setup(gpu, big_sc, big_cam)
        u0 = gpu.render_ahead_stats()[1]
        for k in range(5): gpu.render(k * 5, 5, q)
        got[on] = (gpu.accum(), gpu.readback(abi.RESOLVE_RGBA8_CUDA), gpu.render_ahead_stats()[1] - u0)
    assert np.array_equal(got[True][0], got[False][0]) and np.array_equal(got[True][1], got[False][1])
    assert got[True][2] == 3 and got[False][2] == 0 and np.all(got[True][0][..., 3] == 25)
    # launchRenderKernel, the reference's loop: launch, read `output`, launch
    _setup(gpu, sc, cam)
    out_a = torch.zeros((108, 192, 4), device="cuda", dtype=torch.float32)
    out_b = torch.zeros_like(out_a)
    torch.cuda.synchronize()
    for out, on in ((out_a, False), (out_b, True)):
        gpu.set_render_ahead(on)
        seen = []
        for f in range(7):
            gpu.launch_render_kernel(out.data_ptr(), f)
            gpu.sync()
            seen.append(out.cpu().numpy().copy())
        if on: assert all(np.array_equal(x, y) for x, y in zip(seen, ref_seen))
        else: ref_seen = seen
    gpu.set_render_ahead(True)


def test_render_ahead_under_random_host_behaviour():
    """Two contexts get the same random sequence of calls -- frames that continue, repeat or jump, 1 to 3 samples, sums and
    running means, light sampling on and off, resets, reads, checkpoints put back, camera moves, resizes, scene changes,
    bound tensors, stream switches, episodes of frames in flight -- one with render-ahead, one without: after every call
    that reads, buffer and display are equal bit for bit."""
    import torch
    from path_tracer_amd.renderer import Renderer
    before = torch.cuda.current_stream()
    torch.cuda.set_stream(torch.cuda.Stream())      # (use_torch_stream below: a stream that is not the null stream)
    a, b = Renderer(0), Renderer(0)
    try:
        b.set_render_ahead(False)
        launched0 = a.render_ahead_stats()
        import os
        for seed in range(5, 5 + int(os.environ.get("PT_AHEAD_FUZZ_SEEDS", "2"))):   # (more seeds for a long run by hand)
            rng = np.random.default_rng(20261111 + seed)
            size = (96, 54)
            scene_name = "cuda_default"
            bound = [None, None]
            def setup():
                for r in (a, b):
                    if r._bound is not None: r.sync(); r.unbind()      # (the previous seed's tensor is about to be freed)
                    sc, cam = scenes.builtin(scene_name, *size)
                    r.set_scene(sc); r.set_camera(cam); r.reset()
            setup()
            prm = abi.render_params(fused_rgba8=1)
            s, n = 0, 1
            log = []
            for step in range(260):
                k = int(rng.integers(0, 100))
                log.append((step, k, s, n, prm.acc_mode, prm.fused_rgba8, prm.flags, prm.max_depth, size, scene_name, bound[0] is not None))
                if k < 62:                       # the loop goes on
                    for r in (a, b): r.render(s, n, prm)
                    s += n if prm.acc_mode == abi.ACC_SUM else 1
                elif k < 66:                     # jump or repeat
                    s = int(rng.integers(0, 50))
                elif k < 70:                     # other parameters
                    mean = rng.integers(0, 3) == 0
                    n = 1 if mean else int(rng.integers(1, 4))
                    prm = abi.render_params(fused_rgba8=int(rng.integers(0, 2)), acc_mode=abi.ACC_RUNNING_MEAN if mean else abi.ACC_SUM,
                                            flags=abi.FLAG_NEE_MIS if rng.integers(0, 3) == 0 else 0, max_depth=int(rng.choice([2, 8, 50])))
                    if mean:
                        for r in (a, b): r.reset()
                        s = 0
                elif k < 74:
                    for r in (a, b): r.reset()
                elif k < 78:                     # camera move (same size) or resize
                    if rng.integers(0, 2): size = [(96, 54), (120, 40), (64, 64)][int(rng.integers(0, 3))]
                    eye = (float(rng.uniform(-0.5, 0.5)), 1.0 + float(rng.uniform(-0.3, 0.3)), 3.0)
                    cam = abi.make_camera(eye, (0, 1, 0), (0, 1, 0), 45.0, size[0], size[1], abi.CAMERA_CUDA)
                    for i, r in enumerate((a, b)):
                        if bound[i] is not None: r.sync(); r.unbind(); bound[i] = None     # (work on a caller's buffer is asynchronous: wait before it is freed)
                        r.set_camera(cam); r.reset()
                elif k < 80:
                    scene_name = ["cuda_default", "glass_stress"][int(rng.integers(0, 2))]
                    for r in (a, b):
                        sc, _ = scenes.builtin(scene_name, *size); r.set_scene(sc); r.reset()
                elif k < 83:                     # a caller-owned buffer, or back to the context's own
                    for i, r in enumerate((a, b)):
                        if bound[i] is None:
                            t = torch.zeros((size[1], size[0], 4), device="cuda", dtype=torch.float32); torch.cuda.synchronize()
                            r.bind_tensor(t); bound[i] = t
                        else:
                            r.sync(); r.unbind(); bound[i] = None; r.reset()
                elif k < 86:
                    for r in (a, b): (r.use_torch_stream if rng.integers(0, 2) else r.use_own_stream)()
                elif k < 89:                     # an episode of frames in flight
                    for r in (a, b): r.set_frames_in_flight(2)
                    for _ in range(int(rng.integers(1, 6))):
                        for r in (a, b): r.render(s, n, prm)
                        s += n if prm.acc_mode == abi.ACC_SUM else 1
                    for r in (a, b): r.set_frames_in_flight(1)
                elif k < 92:                     # checkpoint out and in again
                    for r in (a, b): r.upload_accum(r.accum())
                else:                            # the host looks
                    ga, gb = a.accum(), b.accum()
                    if not np.array_equal(ga, gb) and os.path.isdir("gpurun_out"):      # (the steps that led here, for tools/ahead_replay.py)
                        with open("gpurun_out/ahead_fuzz_fail.txt", "w") as f:
                            f.write("\n".join(repr(x) for x in log[-60:]) + f"\nseed {seed} step {step} w: {np.unique(ga[..., 3])} vs {np.unique(gb[..., 3])}\n")
                    assert np.array_equal(ga, gb), (seed, step)
                    if prm.fused_rgba8 and ga[..., 3].max() > 0:
                        assert np.array_equal(a.resolve_to_host(abi.RESOLVE_RGBA8_CUDA), b.resolve_to_host(abi.RESOLVE_RGBA8_CUDA)), (seed, step)
            assert np.array_equal(a.accum(), b.accum()), seed
        launched, used = a.render_ahead_stats()
        assert used > 100 and launched > used and b.render_ahead_stats() == (0, 0)
    finally:
        a.close(); b.close()
        torch.cuda.set_stream(before)
