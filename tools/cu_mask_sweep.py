"""Same-box, same-process sweep of the spatial partition of the chip (CU-masked streams, HipBackend::init):

    python tools/cu_mask_sweep.py <curve> <log2n>[,<log2n>...] [reps] [-- name:ENV=V,ENV=V;opt=v,opt=v ...]

Every configuration is a fresh context created under its own environment ($CTT_HIP_CU_TAIL = CUs per XCD reserved for the tail
stream, $CTT_HIP_CU_MAIN = 0 keeps the main stream on the whole chip, $CTT_HIP_MSM_TAIL / $CTT_HIP_MSM_QUAD = the reduction passes'
tail / four-lane thresholds) plus engine options (early_tail = 2: head merge and every reduction pass on the tail stream whenever the
caller pipelines; depth = MSMs kept in flight, default 2).  Measured per configuration: ms per MSM with two in flight (median of `reps` timed loops), median blocking latency,
and that the result equals the baseline configuration's bytes.  One JSON line per (size, configuration)."""
import collections
import json
import os
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from constantine_amd import DeviceMsm  # noqa: E402
from constantine_amd.msm import CURVES  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

DEFAULT_CONFIGS = [
    "base:",
    "tail1:CTT_HIP_CU_TAIL=1",
    "tail2:CTT_HIP_CU_TAIL=2",
    "tail4:CTT_HIP_CU_TAIL=4",
    "tail8:CTT_HIP_CU_TAIL=8",
    "tail4_shared_main:CTT_HIP_CU_TAIL=4,CTT_HIP_CU_MAIN=0",
    "tail4_early:CTT_HIP_CU_TAIL=4;early_tail=2",
    "tail8_early:CTT_HIP_CU_TAIL=8;early_tail=2",
    "tail4_narrow16k:CTT_HIP_CU_TAIL=4,CTT_HIP_MSM_TAIL=16384",
    "tail4_narrow48k:CTT_HIP_CU_TAIL=4,CTT_HIP_MSM_TAIL=49152",
    "tail8_narrow48k:CTT_HIP_CU_TAIL=8,CTT_HIP_MSM_TAIL=49152",
    "base_again:",
]
ENV_KEYS = ("CTT_HIP_CU_TAIL", "CTT_HIP_CU_MAIN", "CTT_HIP_MSM_TAIL", "CTT_HIP_MSM_QUAD")


def parse(cfg):
    name, _, rest = cfg.partition(":")
    envs, _, opts = rest.partition(";")
    env = dict(kv.split("=") for kv in envs.split(",") if kv)
    opt = {k: int(v) for k, v in (kv.split("=") for kv in opts.split(",") if kv)}
    return name, env, opt


def main():
    argv = sys.argv[1:]
    cfgs = DEFAULT_CONFIGS
    if "--" in argv:
        i = argv.index("--")
        cfgs = argv[i + 1:]
        argv = argv[:i]
    curve = argv[0]
    sizes = [int(x) for x in argv[1].split(",")]
    reps = int(argv[2]) if len(argv) > 2 else 3
    info = CURVES[curve]
    gen = DeviceMsm(0)
    # A CU-masked stream is a BLOCKING stream (hipExtStreamCreateWithCUMask takes no flags): it synchronises with the legacy null
    # stream, torch's default.  DeviceMsm orders the engine behind torch's CURRENT stream with an event record on it -- on the null
    # stream that record waits for everything the masked streams hold, and the two MSMs in flight run one after the other (the first
    # collection of this sweep measured exactly that: 6.5 ms per MSM at 2^20).  So the sweep runs under a side stream of torch's pool.
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    for log2n in sizes:
        n = 1 << log2n
        d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
        gen.gen_points(curve, 0x5EED0002, n, d_points)
        d_scal = torch.from_numpy(synth_scalars(0x5EED0003, n, info.scalar_bits)).cuda()
        torch.cuda.synchronize()
        steps = 60 if log2n <= 18 else 24 if log2n <= 20 else 8
        ref = None
        for cfg in cfgs:
            name, env, opt = parse(cfg)
            for k in ENV_KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            eng = DeviceMsm(0)
            for k, v in opt.items():
                if k != "depth":
                    eng.set_option(k, v)
            eng.enable_timings(False)

            depth = opt.pop("depth", 2)     # MSMs kept in flight (the engine has three slots since round 6)

            def run(k):
                r, pend, sub = None, collections.deque(), 0
                for _ in range(k):
                    while sub < k and len(pend) < depth:
                        pend.append(eng.submit(curve, d_scal, d_points, n))
                        sub += 1
                    r = eng.finish(pend.popleft(), coord="aff")
                return r
            run(4)
            eng.sync()
            ms = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = run(steps)
                eng.sync()
                ms.append((time.perf_counter() - t0) / steps * 1e3)
            plan = eng.last_plan()
            lat = []
            for _ in range(9):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rb = eng.msm(curve, d_scal, d_points, n, coord="aff")
                lat.append((time.perf_counter() - t1) * 1e3)
            eng.enable_timings(True)
            eng.msm(curve, d_scal, d_points, n, coord="aff")
            st = eng.last_timings()
            if ref is None:
                ref = bytes(r)
            print(json.dumps({"curve": curve, "log2n": log2n, "config": name, "env": env, "opt": opt, "in_flight": depth, "plan": plan,
                              "ms_per_msm_two_in_flight": round(statistics.median(ms), 4), "runs": [round(x, 4) for x in ms],
                              "blocking_ms": round(statistics.median(lat[2:]), 4),
                              "same_result": bytes(r) == ref and bytes(rb) == ref,
                              "stage_ms_blocking": {k: round(v, 3) for k, v in st.items()}}), flush=True)
            eng.close()
        del d_points, d_scal
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    gen.close()


if __name__ == "__main__":
    main()
