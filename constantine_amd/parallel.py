"""
Point-sharded multi-GPU MSM: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

MSM is a sum, so MSM(a, P) = sum_g MSM(a_g, P_g) for any partition of the pairs.  This is the reference's own
msm-level parallelism (msmAffine_vartime_parallel_split, ec_multi_scalar_mul_parallel.nim:386-431: balanced
chunks from balancedChunksPrioNumber, threadpool/partitioners.nim:44-77, then `r ~+= partial` at :427-429)
with GPUs in place of threads.  EC addition is not a collective reduction operator, so the only exchange is an
all_gather of one affine point per rank (<= 192 bytes) followed by a local sum on every rank -- latency-bound,
xGMI bandwidth is irrelevant at this size.
"""
import numpy as np

from .curves import CURVES
from .msm import ec_sum_affine


def shard_bounds(n, world_size, rank):
    """balancedChunksPrioNumber (partitioners.nim:44-77): chunk sizes differ by at most one."""
    base, cutoff = divmod(n, world_size)
    start = rank * base + min(rank, cutoff)
    length = base + (1 if rank < cutoff else 0)
    return start, length


def msm_sharded(curve, local_msm, group=None, device=None, coord="aff"):
    """local_msm() -> this rank's partial result as affine bytes (uint8[2*coord]).
    Gathers the partials of all ranks and returns the combined point (same on every rank)."""
    import torch
    import torch.distributed as dist

    info = CURVES[curve]
    part = np.ascontiguousarray(local_msm(), dtype=np.uint8)
    assert part.shape == (info.aff_bytes,)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return ec_sum_affine(curve, part[None, :], coord=coord)
    world = dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    mine = torch.from_numpy(part).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    allp = np.stack([g.cpu().numpy() for g in gathered])
    return ec_sum_affine(curve, allp, coord=coord)
