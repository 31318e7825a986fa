"""Window-bits sweep of the table-less engine: python tools/bench_c.py <curve> <log2n> <c ...> (0 = the plan's choice)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from constantine_amd import DeviceMsm  # noqa: E402
from constantine_amd.msm import CURVES  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

curve, log2n = sys.argv[1], int(sys.argv[2])
cs = [int(x) for x in sys.argv[3:]] or [0]
info = CURVES[curve]
n = (1 << log2n) + int(os.environ.get("N_EXTRA", "0"))   # N_EXTRA: sizes that are not a power of two
eng = DeviceMsm(0)
if os.environ.get("HWS"):
    eng.set_option("host_window_sums", int(os.environ["HWS"]))   # 0 auto, 1 host, 2 device
d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
eng.gen_points(curve, 0x5EED0002, n, d_points)
d_scal = torch.from_numpy(synth_scalars(0x5EED0003, n, info.scalar_bits)).cuda()
torch.cuda.synchronize()
ref = None
steps = 60 if log2n <= 18 else 10 if log2n <= 22 else 4
for c in cs:
    eng.set_option("c", c)
    eng.enable_timings(False)

    def run(k):
        r = None
        pend = eng.submit(curve, d_scal, d_points, n)
        for i in range(k):
            nxt = eng.submit(curve, d_scal, d_points, n) if i + 1 < k else None
            r = eng.finish(pend, coord="aff")
            pend = nxt
        return r
    run(2)
    eng.sync()
    t0 = time.perf_counter()
    r = run(steps)
    eng.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    eng.enable_timings(True)
    eng.msm(curve, d_scal, d_points, n, coord="aff")
    st = eng.last_timings()
    if ref is None:
        ref = bytes(r)
    print(json.dumps({"curve": curve, "log2n": log2n, "c_opt": c, "plan": eng.last_plan(), "ms_per_step": round(ms, 4),
                      "Mpairs_s": round(n / ms / 1e3, 1), "same": bytes(r) == ref,
                      "stage_ms_blocking": {k: round(v, 3) for k, v in st.items()}}), flush=True)
