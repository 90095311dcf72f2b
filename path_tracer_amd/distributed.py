"""Sample-split multi-GPU rendering: one process per GPU, disjoint sample ranges, one sum-reduce.

The path shards by SAMPLE, not by image region (north_star): every rank keeps the
whole frame, rank r traces the sample indices `partition(...)` hands it, and the float4
accumulation buffers (sum r,g,b + sample count in .w, so counts combine in the same reduce) are
added once per render with a `torch.distributed` collective — RCCL over xGMI when the tensors live
on GPUs (backend "nccl"), gloo on CPU in the tests: a reduce to one rank that then resolves the picture
(`reduce_accum`), or a reduce-scatter after which every rank resolves its own slice of the pixels and the
4-byte pixels are gathered (`reduce_scatter_accum` + `gather_display`, what bench.py times).  Because the RNG is keyed on the absolute
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


def pixel_slice(npix: int, world_size: int, rank: int) -> Tuple[int, int]:
    """The pixels [first, first + count) whose sums `reduce_scatter_accum` leaves on `rank`: equal chunks of
    ceil(npix / world) pixels, the last ranks' chunks cut off at the end of the frame (possibly empty)."""
    if world_size < 1 or not (0 <= rank < world_size) or npix < 0:
        raise ValueError("bad slice request")
    chunk = -(-npix // world_size)
    first = min(rank * chunk, npix)
    return first, min(chunk, npix - first)


def reduce_scatter_accum(accum) -> Tuple[int, int]:
    """Sum the accumulation buffers over ranks so that every rank ends up with the sums of ITS slice of the pixels
    (SURVEY §8e's cheaper collective: each GPU receives and adds 1/N of the frame instead of one GPU receiving all of
    it).  `accum` is a contiguous float32 tensor [H, W, 4]; on return its pixels [first, first + count) -- the returned
    pair, `pixel_slice(H * W, world, rank)` -- hold the sums (and sample counts) of all ranks; the other pixels still
    hold this rank's own.  One `reduce_scatter_tensor` (RCCL over xGMI on GPUs, gloo on CPU), enqueued on the current
    stream like every torch collective: nothing here synchronises the host."""
    import torch
    import torch.distributed as dist
    npix = accum.numel() // 4
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0, npix
    world, rank = dist.get_world_size(), dist.get_rank()
    chunk = -(-npix // world)
    flat = accum.view(-1)
    if chunk * world != npix:   # the collective wants equal chunks: pad with zeros (a frame whose pixel count the ranks do not divide)
        padded = torch.zeros(chunk * world * 4, dtype=accum.dtype, device=accum.device)
        padded[:npix * 4] = flat
    else:
        padded = flat
    if accum.is_cuda and dist.get_backend() == "gloo":   # (the one-GPU rehearsal of bench.py: gloo's collectives are staged through the host)
        mine = torch.empty(chunk * 4, dtype=accum.dtype)
        dist.reduce_scatter_tensor(mine, padded.cpu(), op=dist.ReduceOp.SUM)
        mine = mine.to(accum.device)
    else:
        mine = torch.empty(chunk * 4, dtype=accum.dtype, device=accum.device)
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM)
    first, count = pixel_slice(npix, world, rank)
    flat[first * 4:(first + count) * 4] = mine[:count * 4]
    return first, count


def gather_display(display_slice, npix: int):
    """All ranks' resolved slices (int32 words, 0x00RRGGBB, `pixel_slice` order) -> the whole picture [npix] on every
    rank: one all-gather of 4 bytes per pixel.  `display_slice` holds ceil(npix / world) words (the tail beyond the
    rank's pixel count is ignored)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return display_slice[:npix]
    world = dist.get_world_size()
    chunk = -(-npix // world)
    assert display_slice.numel() == chunk
    if display_slice.is_cuda and dist.get_backend() == "gloo":
        full = torch.empty(chunk * world, dtype=display_slice.dtype)
        dist.all_gather_into_tensor(full, display_slice.cpu())
        return full[:npix].to(display_slice.device)
    full = torch.empty(chunk * world, dtype=display_slice.dtype, device=display_slice.device)
    dist.all_gather_into_tensor(full, display_slice)
    return full[:npix]


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
