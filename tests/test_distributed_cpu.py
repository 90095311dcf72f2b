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


def test_reduce_is_a_noop_without_a_process_group():
    import torch
    t = torch.ones(2, 2, 4)
    assert distributed.reduce_accum(t) is t and float(t.sum()) == 16.0
