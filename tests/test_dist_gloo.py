"""world_size-2 / 3 / 8 gloo tests of the point-sharded MSM (constantine_amd/parallel.py) on CPU.
There is no GPU in this container, so every rank's partial MSM is computed by tests/emu -- the CPU emulator that runs the
engine's own kernel bodies and host orchestration (msm_bodies.h, msm_pipeline.h) -- and the sharding, the all_gather
exchange and the host-side combine (product code, ctt_hip_ec_sum_affine) run as they do on the GPUs.  The oracle only
checks the combined result.  The GPU engine itself runs the same two-rank path in tests/test_gpu_parity.py
(two contexts on one device) and in `bench.py --gpus 2 --all-ranks-on-device 0 --backend gloo`."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, n, steps, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from constantine_amd import parallel
    from oracle import cref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bits = 254 if "bn254" in name else 255
        start, ln = parallel.shard_bounds(n, world, rank)
        pts = cref.gen_points(name, 77, ln, first=start, nthreads=1)
        sc = cref.synth_scalars(78, ln, bits, first=start)
        from tests.emu import emu
        res = parallel.msm_sharded(name, lambda: emu.msm(name, sc, pts, out_kind=0)[0])
        # the pipelined form bench.py uses: exchange i is completed after exchange i+1 has been started (two buffer sets);
        # step k multiplies by k + 1 through the scalars' low word so that a stale buffer would show
        x = parallel.ShardExchange(name)
        outs, prev = [], None
        for k in range(steps):
            sck = sc.copy()
            sck[:, 0] = (k * 37 + 1) & 0xFF
            h = x.start(emu.msm(name, sck, pts, out_kind=0)[0])
            if prev is not None:
                outs.append(bytes(x.finish(prev)))
            prev = h
        outs.append(bytes(x.finish(prev)))
        q.put((rank, bytes(res), outs))
    finally:
        dist.destroy_process_group()


# world sizes 2, 3 and 8 (the node the metric is quoted on); n not divisible by the world size (uneven remainders of
# balancedChunksPrioNumber, partitioners.nim:44-77), n < world size (ranks with an EMPTY shard contribute the neutral element), and
# the pipelined exchange over >= 4 steps (both buffer sets reused at least once)
@pytest.mark.parametrize("name,n,world,steps", [
    ("bls12_381_g1", 101, 2, 3), ("bn254_snarks_g1", 64, 2, 3),
    ("bls12_381_g1", 100, 3, 4),
    ("bls12_381_g1", 203, 8, 5),
    ("bn254_snarks_g1", 5, 8, 4),       # five ranks hold one pair each, three ranks none
    ("bls12_381_g2", 37, 8, 4),
])
def test_sharded_msm_over_gloo_ranks(name, n, world, steps):
    from oracle import cref
    from tests.emu import emu
    emu.lib()                       # built here, before the ranks start (they would otherwise all run `make`)
    cref.lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    out = {r: a for r, a, _ in got}
    piped = {r: b for r, _, b in got}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(out) == list(range(world))
    bits = 254 if "bn254" in name else 255
    pts = cref.gen_points(name, 77, n)
    sc = cref.synth_scalars(78, n, bits)
    expect, _ = cref.msm(name, sc, pts)
    for r in range(world):
        assert out[r] == bytes(expect), r
    for k in range(steps):
        sck = sc.copy()
        sck[:, 0] = (k * 37 + 1) & 0xFF
        expect, _ = cref.msm(name, sck, pts)
        for r in range(world):
            assert piped[r][k] == bytes(expect), (r, k)


def test_shard_bounds_balanced():
    from constantine_amd.parallel import shard_bounds
    for n, w in ((40, 12), (7, 8), (1 << 24, 8), (5, 2), (0, 8), (1, 8), (203, 8), ((1 << 20) + 3, 3)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and sum(l for _, l in spans) == n
        for (s0, l0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + l0 == s1
        ls = [l for _, l in spans]
        assert max(ls) - min(ls) <= 1
