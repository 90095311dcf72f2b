"""The N > 1 path on CPU: world_size 2, gloo.  The host logic under test is
path_tracer_amd/distributed.py (partition, progressive schedule, reduce); the local "renderer"
is the CPU oracle, standing in for the HIP kernel so the test runs without a GPU."""
import os
import socket

import numpy as np
import pytest

from path_tracer_amd import abi, distributed, scenes


def test_partition_covers_every_sample_once():
    for total in (0, 1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, n = distributed.partition(total, world, r)
                seen += list(range(b, b + n))
            assert seen == list(range(total))
            counts = [distributed.partition(total, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        distributed.partition(8, 2, 2)


def test_progressive_schedule_is_disjoint_and_dense():
    world, steps, spp = 4, 5, 2
    seen = []
    for r in range(world):
        for s0, n in distributed.progressive_schedule(10, steps, world, r, spp):
            assert n == spp
            seen += list(range(s0, s0 + n))
    assert sorted(seen) == list(range(10, 10 + world * steps * spp))


def test_strong_schedule_divides_the_steps():
    """bench.py's config-3 leg: 64 steps of 16 spp divided among the ranks, uneven shares included."""
    for world in (1, 2, 3, 4, 8, 5):
        for steps, spp, first in ((64, 16, 96), (7, 3, 0), (2, 1, 5)):
            seen, shares = [], []
            for r in range(world):
                sched = distributed.strong_schedule(first, steps, world, r, spp)
                shares.append(len(sched))
                for s0, n in sched:
                    assert n == spp
                    seen += list(range(s0, s0 + n))
            assert sorted(seen) == list(range(first, first + steps * spp)) and len(seen) == len(set(seen))
            assert max(shares) - min(shares) <= 1 and sum(shares) == steps


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_path):
    import torch
    import torch.distributed as dist

    from tests.oracle_binding import load_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = load_oracle()
    sc, cam = scenes.cuda_default(32, 18)
    p = abi.render_params()
    acc = np.zeros((18, 32, 4), np.float32)

    def render(s0, n):
        oracle.render_path(sc, cam, s0, n, p, precision=32, accum=acc)

    # strong scaling: this rank's share of `total` samples, then progressive frames on top, then bench.py's
    # config-3 schedule (5 steps of 2 samples divided among the ranks: uneven for 2, 3 and 8 ranks)
    distributed.render_split(render, total, world, rank, chunk=3)
    for s0, n in distributed.progressive_schedule(total, 2, world, rank, 1):
        render(s0, n)
    for s0, n in distributed.strong_schedule(total + 2 * world, 5, world, rank, 2):
        render(s0, n)
    t = torch.from_numpy(acc)
    distributed.reduce_accum(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sample_split_over_ranks_equals_single_rank(tmp_path, oracle, world):
    """world_size 2, 3 and 8 over gloo (uneven shares for every schedule): the reduced buffer holds every sample once."""
    import torch.multiprocessing as mp
    total = 10
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    got = np.load(out)
    sc, cam = scenes.cuda_default(32, 18)
    n = total + 2 * world + 5 * 2
    want = oracle.render_path(sc, cam, 0, n, abi.render_params(), precision=32)
    assert np.all(got[..., 3] == n)                           # counts combined by the same reduce
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)  # equal up to FP32 summation order


def test_pixel_slices_cover_the_frame_once():
    for npix in (1, 7, 576, 921600, 8294400, 1280 * 720 + 3):
        for world in (1, 2, 3, 8):
            seen = 0
            for r in range(world):
                first, count = distributed.pixel_slice(npix, world, r)
                assert first == seen and 0 <= count <= -(-npix // world)
                seen += count
            assert seen == npix


def _worker_scatter(rank, world, port, size, total, out_path):
    """bench.py's N > 1 combine step on CPU: reduce-scatter of the float4 buffers, every rank resolves its own slice
    (the oracle's pack stands in for pt_resolve_range), all-gather of the RGBA8 words."""
    import torch
    import torch.distributed as dist

    from tests.oracle_binding import load_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = load_oracle()
    W, H = size
    sc, cam = scenes.cuda_default(W, H)
    p = abi.render_params()
    acc = np.zeros((H, W, 4), np.float32)
    for s0, n in distributed.strong_schedule(0, total, world, rank, 1):
        oracle.render_path(sc, cam, s0, n, p, precision=32, accum=acc)
    own = acc.copy()
    t = torch.from_numpy(acc)
    first, count = distributed.reduce_scatter_accum(t)
    npix = W * H
    assert (first, count) == distributed.pixel_slice(npix, world, rank)
    flat = t.numpy().reshape(-1, 4)
    assert np.all(flat[first:first + count, 3] == total)                     # the slice holds every rank's samples ...
    outside = np.ones(npix, bool); outside[first:first + count] = False
    assert np.array_equal(flat[outside], own.reshape(-1, 4)[outside])         # ... the rest is untouched
    chunk = -(-npix // world)
    words = np.zeros(chunk, np.uint32)
    words[:count] = oracle.resolve(flat[first:first + count], abi.ACC_SUM, abi.RESOLVE_RGBA8_CUDA)
    full = distributed.gather_display(torch.from_numpy(words.view(np.int32)), npix)
    np.save(out_path + f".{rank}.npy", full.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (32, 18)), (3, (31, 17)), (8, (32, 18)), (8, (5, 1))])
def test_reduce_scatter_resolve_gather_equals_single_rank(tmp_path, oracle, world, size):
    """SURVEY §8(e)'s cheaper combine at world sizes 2, 3 and 8 over gloo: frames whose pixel count the ranks divide
    and frames they do not (padding; ranks with an empty slice), and the picture every rank ends up with is the one a
    single rank makes from all the samples (to one 8-bit level where the float summation order moves a channel across a
    truncation boundary)."""
    import torch.multiprocessing as mp
    total = 9
    out = str(tmp_path / "display")
    mp.spawn(_worker_scatter, args=(world, _free_port(), size, total, out), nprocs=world, join=True)
    sc, cam = scenes.cuda_default(*size)
    want_acc = np.zeros((size[1], size[0], 4), np.float32)
    # the ranks' float sums are added pairwise by the collective, the single rank adds sample by sample: the means can
    # differ in the last bit, the 8-bit pictures only where a channel sits on a rounding boundary
    oracle.render_path(sc, cam, 0, total, abi.render_params(), precision=32, accum=want_acc)
    want = oracle.resolve(want_acc, abi.ACC_SUM, abi.RESOLVE_RGBA8_CUDA)
    pics = [np.load(out + f".{r}.npy") for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(pics[r], pics[0])                               # every rank holds the same picture
    got = pics[0]
    assert got.shape == want.shape
    chan = lambda w: np.stack([(w >> 16) & 255, (w >> 8) & 255, w & 255], -1).astype(np.int64)
    assert np.abs(chan(got) - chan(want)).max() <= 1 and (got != want).mean() < 0.02


def test_reduce_is_a_noop_without_a_process_group():
    import torch
    t = torch.ones(2, 2, 4)
    assert distributed.reduce_accum(t) is t and float(t.sum()) == 16.0
