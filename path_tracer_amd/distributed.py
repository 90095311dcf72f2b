"""Sample-split multi-GPU rendering: one process per GPU, disjoint sample ranges, one sum-reduce.

The path shards by SAMPLE, not by image region (north_star): every rank keeps the
whole frame, rank r traces the sample indices `partition(...)` hands it, and the float4
accumulation buffers (sum r,g,b + sample count in .w, so counts combine in the same reduce) are
added once per render with a `torch.distributed` reduce — RCCL over xGMI when the tensors live
on GPUs (backend "nccl"), gloo on CPU in the tests.  Because the RNG is keyed on the absolute
sample index, the N-rank image equals the 1-rank image up to FP32 summation order.

Backend-agnostic on purpose: `render_fn(sample_begin, sample_count)` is whatever adds samples to
the local buffer (the HIP renderer in bench.py; the CPU oracle in tests/).
"""
from __future__ import annotations

from typing import Callable, List, Tuple


def partition(total_samples: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous share [begin, begin+count) of `total_samples` for `rank`; shares differ by at most 1."""
    if world_size < 1 or not (0 <= rank < world_size) or total_samples < 0:
        raise ValueError("bad partition request")
    base, rem = divmod(total_samples, world_size)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def progressive_schedule(first_sample: int, steps: int, world_size: int, rank: int, spp_per_step: int = 1) -> List[Tuple[int, int]]:
    """Per-step (sample_begin, sample_count) for a progressive render in which every step adds
    `spp_per_step` samples on EVERY rank (weak scaling): step k covers the global sample block
    [first + k*W*s, first + (k+1)*W*s) and rank r takes the r-th slice of it."""
    out = []
    for k in range(steps):
        block = first_sample + k * world_size * spp_per_step
        out.append((block + rank * spp_per_step, spp_per_step))
    return out


def strong_schedule(first_sample: int, steps: int, world_size: int, rank: int, spp_per_step: int = 1) -> List[Tuple[int, int]]:
    """Per-step (sample_begin, sample_count) of THIS rank when `steps` steps of `spp_per_step` samples, starting at
    sample `first_sample`, are divided among the ranks (strong scaling, BASELINE config 3): rank r gets a contiguous
    run of steps, shares differing by at most one step.  Over all ranks every sample appears exactly once."""
    begin, count = partition(steps, world_size, rank)
    return [(first_sample + (begin + g) * spp_per_step, spp_per_step) for g in range(count)]


def render_split(render_fn: Callable[[int, int], None], total_samples: int, world_size: int, rank: int,
                 chunk: int = 64, first_sample: int = 0) -> int:
    """Strong scaling: trace this rank's share of `total_samples` in launches of at most `chunk`."""
    begin, count = partition(total_samples, world_size, rank)
    done = 0
    while done < count:
        n = min(chunk, count - done)
        render_fn(first_sample + begin + done, n)
        done += n
    return count


def reduce_accum(accum, dst: int = 0, all_ranks: bool = False):
    """Sum the accumulation buffers over ranks (in place).  `accum` is a torch tensor [H, W, 4]."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return accum
    if all_ranks:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
