"""Multi-GPU host logic: camera sequences shard across ranks, one process per GPU, no data-path
collective.  torch.distributed is used only to agree on the timing window and to sum the work done."""
from __future__ import annotations

from typing import List, Tuple


def sequences_for_rank(n_sequences: int, world: int, rank: int) -> List[int]:
    """Sequence s runs on GPU s mod world (each sequence = one extractor = one HIP stream)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return [s for s in range(n_sequences) if s % world == rank]


def reduce_throughput(local_seconds: float, local_units: float, device=None) -> Tuple[float, float]:
    """(max over ranks of the timed window, sum over ranks of processed units).  Works on gloo (CPU tensors) and
    on nccl/RCCL (pass device='cuda')."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([local_units], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
