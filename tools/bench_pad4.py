#!/usr/bin/env python3
"""Upper bound of what padding the sorted buckets to multiples of four entries could give the 254/255-bit accumulate kernel (round-5 review
item 6), MEASURED instead of counted:

    python tools/bench_pad4.py [curve=bn254_snarks_g1] [log2n=22]

tools/libctt_msm_hip_pad4.so is the library with ONE change: the accumulate loop looks for a bucket boundary only at positions that are
multiples of four.  It is fed an input on which that is exact -- every scalar appears four times in a row (with four different points), so
every bucket of every window holds a multiple of four entries and no dummy entry is needed: the gain side of the padding with none of its
costs (1.5 dummy additions per bucket, a second entry format through the sort).  Both libraries run the same input with the same plan
(K a multiple of four); the result bytes must agree.  One JSON line per (library, c)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(curve, log2n, repeat):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sys.path.insert(0, ROOT)
    import statistics
    import time
    import zlib
    import numpy as np
    import torch
    from constantine_amd import DeviceMsm
    from constantine_amd.msm import CURVES
    from constantine_amd.synth import synth_scalars
    info = CURVES[curve]
    n = 1 << log2n
    eng = DeviceMsm(0)
    d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(curve, 0x5EED0002, n, d_points)
    sc = np.repeat(synth_scalars(0x5EED0003, n // repeat, info.scalar_bits), repeat, axis=0)
    d_scal = torch.from_numpy(np.ascontiguousarray(sc)).cuda()
    torch.cuda.synchronize()
    for c in (0, 16):
        eng.set_option("c", c)
        eng.set_option("K", 0)
        eng.enable_timings(False)
        eng.msm(curve, d_scal, d_points, n, coord="aff")
        K = eng.last_plan()["K"]
        K4 = (K + 3) // 4 * 4
        eng.set_option("K", K4)
        eng.enable_timings(True)
        acc, res = [], None
        for _ in range(12):
            res = eng.msm(curve, d_scal, d_points, n, coord="aff")
            acc.append(eng.last_timings()["accumulate"])
        eng.enable_timings(False)
        steps = 12

        def run(k):
            r = None
            pend = eng.submit(curve, d_scal, d_points, n)
            for i in range(k):
                nxt = eng.submit(curve, d_scal, d_points, n) if i + 1 < k else None
                r = eng.finish(pend, coord="aff")
                pend = nxt
            return r
        run(3)
        eng.sync()
        t0 = time.perf_counter()
        run(steps)
        eng.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(json.dumps({"lib": os.path.basename(os.environ.get("CTT_MSM_HIP_LIB", "in-tree")), "scalars_repeated": repeat, "curve": curve, "log2n": log2n, "plan": eng.last_plan(),
                          "accumulate_ms_median": round(statistics.median(acc[2:]), 4), "accumulate_ms_min": round(min(acc[2:]), 4),
                          "ms_per_msm_two_in_flight": round(ms, 4), "crc": zlib.crc32(bytes(res))}), flush=True)
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        curve = sys.argv[1] if len(sys.argv) > 1 else "bn254_snarks_g1"
        log2n = sys.argv[2] if len(sys.argv) > 2 else "22"
        pad4 = os.path.join(ROOT, "tools", "libctt_msm_hip_pad4.so")
        for rep in range(2):
            for lib, repeat in ((None, 1), (None, 4), (pad4, 4)):
                if lib and not os.path.exists(lib):
                    continue
                env = dict(os.environ)
                env.pop("CTT_MSM_HIP_LIB", None)
                if lib:
                    env["CTT_MSM_HIP_LIB"] = lib
                subprocess.call([sys.executable, os.path.abspath(__file__), "--child", curve, log2n, str(repeat)], env=env)
