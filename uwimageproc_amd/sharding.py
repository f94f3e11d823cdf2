"""Frame-batch partitioning across ranks.

Frames are independent units, so the data path has NO collective: each rank
(one process per GPU) takes a contiguous, balanced slice of every batch of frame
indices.  The only exchange is control-plane: gathering the per-frame scalar
results (overlap ratios, chosen CLAHE parameters) back into frame order, and the
max-over-ranks step time of the benchmark.  Works with any torch.distributed
backend ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def frame_slice(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `n_frames` for `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad rank/world/n_frames")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_slices(n_frames: int, world: int) -> List[Tuple[int, int]]:
    return [frame_slice(n_frames, r, world) for r in range(world)]


def gather_in_frame_order(local_values: Sequence[float], n_frames: int, rank: int, world: int, group=None):
    """All ranks receive the per-frame values of the whole batch, in frame order.
    `local_values` has one entry per frame of this rank's slice."""
    import torch
    import torch.distributed as dist

    a, b = frame_slice(n_frames, rank, world)
    if len(local_values) != b - a:
        raise ValueError("local_values does not match this rank's slice")
    if world == 1 or not dist.is_initialized():
        return list(local_values)
    width = max(s[1] - s[0] for s in all_slices(n_frames, world))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = torch.zeros(width, dtype=torch.float64, device=dev)
    buf[: b - a] = torch.as_tensor(list(local_values), dtype=torch.float64)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    res: List[float] = []
    for r, (s0, s1) in enumerate(all_slices(n_frames, world)):
        res += out[r][: s1 - s0].cpu().tolist()
    return res


def max_over_ranks(value: float, group=None) -> float:
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
