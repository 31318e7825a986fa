"""RCCL on the driver's single-GPU box: constantine_amd/parallel.py with backend "nccl" (= RCCL on ROCm) in a
world of ONE rank -- the all_gather of the partial result really goes through RCCL (always_collective), the GPU engine computes the
partial, the oracle checks the combined point.  The blocking form and the pipelined form bench.py --gpus N uses.  An 8-GPU run adds
ranks to exactly this code path; what it cannot show is xGMI."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from constantine_amd import DeviceMsm, parallel
    from oracle import cref
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        bits = 254 if "bn254" in name else 255
        start, ln = parallel.shard_bounds(n, world, rank)
        pts = cref.gen_points(name, 77, ln, first=start, nthreads=4)
        sc = cref.synth_scalars(78, ln, bits, first=start)
        eng = DeviceMsm(0)
        d_p, d_s = torch.from_numpy(pts).cuda(), torch.from_numpy(sc).cuda()
        res = parallel.msm_sharded(name, lambda: eng.msm(name, d_s, d_p, ln, coord="aff"), always_collective=True)
        x = parallel.ShardExchange(name, always_collective=True)
        assert x.collective
        outs, prev = [], None
        for k in range(3):   # exchange k completed after exchange k+1 was started (two buffer sets)
            sck = sc.copy()
            sck[:, 0] = (k * 37 + 1) & 0xFF
            h = x.start(eng.msm(name, torch.from_numpy(sck).cuda(), d_p, ln, coord="aff"))
            if prev is not None:
                outs.append(bytes(x.finish(prev)))
            prev = h
        outs.append(bytes(x.finish(prev)))
        # the reduction bench.py uses for its timing (MAX over ranks) and the rank count it reports
        t = torch.tensor([1.5 + rank], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        eng.close()
        q.put((rank, bytes(res), outs, float(t.item()), int(ones.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,n", [("bls12_381_g1", 5000), ("bn254_snarks_g1", 3000)])
def test_sharded_msm_over_rccl_world_of_one(name, n):
    import torch.multiprocessing as mp
    from oracle import cref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    p = ctx.Process(target=_worker, args=(0, 1, port, name, n, q))
    p.start()
    try:
        rank, res, outs, tmax, ranks_seen = q.get(timeout=300)
    finally:
        p.join(60)
    assert p.exitcode == 0
    bits = 254 if "bn254" in name else 255
    pts = cref.gen_points(name, 77, n, nthreads=4)
    sc = cref.synth_scalars(78, n, bits)
    assert res == bytes(cref.msm(name, sc, pts, nthreads=4)[0])
    for k, o in enumerate(outs):
        sck = sc.copy()
        sck[:, 0] = (k * 37 + 1) & 0xFF
        assert o == bytes(cref.msm(name, sck, pts, nthreads=4)[0]), k
    assert tmax == 1.5 and ranks_seen == 1
