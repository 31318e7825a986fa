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


class ShardExchange:
    """The exchange of the partial results, split in two so that it stays off a pipeline's critical path: start() hands this
    rank's affine partial to an asynchronous all_gather (RCCL's own stream for "nccl"; two buffer sets, so one exchange may be
    in flight while the next MSM runs), finish() waits for it and sums the world_size partials on the host.  A caller that
    keeps MSMs in flight finishes exchange i after it has started exchange i+1 (bench.py); msm_sharded() below is the
    blocking form."""

    def __init__(self, curve, group=None, device=None, always_collective=False):
        """always_collective: run the all_gather even in a world of one rank (tests: the first RCCL collective this code
        issues should not be on an 8-GPU scaling run)."""
        import torch
        import torch.distributed as dist
        self.curve = curve
        self.info = CURVES[curve]
        self.group = group
        self.dist = dist
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.collective = self.world > 1 or (always_collective and dist.is_available() and dist.is_initialized())
        self.slot = 0
        if self.collective:
            dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
            nb = self.info.aff_bytes
            self.mine = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
            # ONE flat receive buffer per set (all_gather_into_tensor): finish() brings the world_size partials to the host with a
            # single copy -- a list of per-rank tensors cost one device-to-host copy (and its synchronisation, ~15 us) PER RANK, 0.1 ms of
            # an 8-GPU step whose MSM is 0.65 ms (2^20 pairs in total: 2^17 per GPU)
            self.flat = [torch.empty(nb * self.world, dtype=torch.uint8, device=dev) for _ in range(2)]
            self.into_tensor = hasattr(dist, "all_gather_into_tensor")

    def start(self, part):
        import torch
        part = np.ascontiguousarray(part, dtype=np.uint8)
        assert part.shape == (self.info.aff_bytes,)
        if not self.collective:
            return (None, part)
        k = self.slot
        self.slot ^= 1
        self.mine[k].copy_(torch.from_numpy(part))
        work = None
        if self.into_tensor:
            try:
                work = self.dist.all_gather_into_tensor(self.flat[k], self.mine[k], group=self.group, async_op=True)
            except (RuntimeError, NotImplementedError):
                self.into_tensor = False      # (a backend without the flat form: the list form below, same bytes)
        if work is None:
            views = list(self.flat[k].view(self.world, -1).unbind(0))
            work = self.dist.all_gather(views, self.mine[k], group=self.group, async_op=True)
        return (work, k)

    def finish(self, handle, coord="aff"):
        work, k = handle
        if work is None:
            return ec_sum_affine(self.curve, k[None, :], coord=coord)
        work.wait()
        allp = self.flat[k].cpu().numpy().reshape(self.world, self.info.aff_bytes)
        return ec_sum_affine(self.curve, allp, coord=coord)


def msm_sharded(curve, local_msm, group=None, device=None, coord="aff", always_collective=False):
    """local_msm() -> this rank's partial result as affine bytes (uint8[2*coord]).
    Gathers the partials of all ranks and returns the combined point (same on every rank)."""
    x = ShardExchange(curve, group=group, device=device, always_collective=always_collective)
    return x.finish(x.start(local_msm()), coord=coord)
