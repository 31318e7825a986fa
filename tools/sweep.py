"""Option sweep of the device-resident engine inside ONE process (one torch import, inputs generated once per size):

    python tools/sweep.py <curve> <log2n> key=v1,v2,... [key=...] [-- <curve> <log2n> key=...]

keys are engine options (c, K, reduce_block, horner_bits, quad_ratio, host_window_sums); the cartesian product of the value
lists is measured: ms per MSM with two in flight, median blocking latency, stage times of one blocking call.  One JSON line
per configuration; `same` = the result equals the first configuration's of that (curve, size)."""
import itertools
import json
import os
import statistics
import zlib
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from constantine_amd import DeviceMsm  # noqa: E402
from constantine_amd.msm import CURVES  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402


def groups(argv):
    g = []
    for a in argv:
        if a == "--":
            if g:
                yield g
            g = []
        else:
            g.append(a)
    if g:
        yield g


def main():
    eng = DeviceMsm(0)
    cache = {}
    for grp in groups(sys.argv[1:]):
        curve, log2n = grp[0], int(grp[1])
        info = CURVES[curve]
        n = 1 << log2n
        if (curve, n) not in cache:
            cache.clear()
            d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
            eng.gen_points(curve, 0x5EED0002, n, d_points)
            d_scal = torch.from_numpy(synth_scalars(0x5EED0003, n, info.scalar_bits)).cuda()
            torch.cuda.synchronize()
            cache[(curve, n)] = (d_points, d_scal)
        d_points, d_scal = cache[(curve, n)]
        keys, lists = [], []
        for kv in grp[2:]:
            k, v = kv.split("=")
            keys.append(k)
            lists.append([int(x) for x in v.split(",")])
        steps = 60 if log2n <= 18 else 20 if log2n <= 20 else 6
        ref = None
        for combo in itertools.product(*lists):
            for k, v in zip(keys, combo):
                eng.set_option(k, v)
            eng.enable_timings(False)

            host = {"submit": 0.0, "finish": 0.0, "n": 0}

            def run(k):
                r = None
                pend = eng.submit(curve, d_scal, d_points, n)
                for i in range(k):
                    ta = time.perf_counter()
                    nxt = eng.submit(curve, d_scal, d_points, n) if i + 1 < k else None
                    tb = time.perf_counter()
                    r = eng.finish(pend, coord="aff")
                    tc = time.perf_counter()
                    host["submit"] += tb - ta
                    host["finish"] += tc - tb
                    host["n"] += 1
                    pend = nxt
                return r
            run(3)
            eng.sync()
            host.update(submit=0.0, finish=0.0, n=0)
            t0 = time.perf_counter()
            r = run(steps)
            eng.sync()
            ms = (time.perf_counter() - t0) / steps * 1e3
            lat = []
            for _ in range(9):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                eng.msm(curve, d_scal, d_points, n, coord="aff")
                lat.append((time.perf_counter() - t1) * 1e3)
            eng.enable_timings(True)
            eng.msm(curve, d_scal, d_points, n, coord="aff")
            st = eng.last_timings()
            if ref is None:
                ref = bytes(r)
            print(json.dumps({"curve": curve, "log2n": log2n, "opt": dict(zip(keys, combo)), "plan": eng.last_plan(),
                              "ms_per_step": round(ms, 4), "Mpairs_s": round(n / ms / 1e3, 1),
                              "blocking_ms": round(statistics.median(lat[2:]), 4), "same": bytes(r) == ref, "crc": zlib.crc32(bytes(r)),
                              "host_ms_in_submit": round(host["submit"] / max(1, host["n"]) * 1e3, 4),
                              "host_ms_in_finish": round(host["finish"] / max(1, host["n"]) * 1e3, 4),
                              "stage_ms_blocking": {k: round(v, 3) for k, v in st.items()}}), flush=True)
        for k in keys:
            eng.set_option(k, 0)
    eng.close()


if __name__ == "__main__":
    main()
