"""Multi-GPU host logic: camera sequences shard across ranks, one process per GPU, no data-path
collective.  torch.distributed is used only to agree on the timing window and to sum the work done."""
from __future__ import annotations

from typing import List, Tuple


def sequences_for_rank(n_sequences: int, world: int, rank: int) -> List[int]:
    """Sequence s runs on GPU s mod world (each sequence = one extractor = one HIP stream)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return [s for s in range(n_sequences) if s % world == rank]


def init_host_group(rank: int, world: int) -> None:
    """The process group of the bench: gloo over CPU tensors.  The path has no data-path collective (north_star: "RCCL not
    required"), so nothing but a host-side barrier and two scalar reductions ever crosses ranks -- no communicator, proxy thread or
    device buffer is created on the GPUs that run the extraction."""
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)


def gather_throughput(local_seconds: float, local_units: float) -> Tuple[float, float, List[Tuple[float, float]]]:
    """(max over ranks of the timed window, sum over ranks of processed units, [(seconds, units) of every rank]) over the host group;
    the per-rank pairs make a straggler visible."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([local_seconds, local_units], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        allr = [torch.zeros(2, dtype=torch.float64) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    per = [(float(x[0]), float(x[1])) for x in allr]
    return max(p[0] for p in per), sum(p[1] for p in per), per


def reduce_throughput(local_seconds: float, local_units: float, device=None) -> Tuple[float, float]:
    """(max over ranks of the timed window, sum over ranks of processed units)."""
    t, u, _ = gather_throughput(local_seconds, local_units)
    return t, u
