"""Window-table sweep: python tools/bench_table.py [curve] [log2n] [c ...] -- per c: build time, ms per pipelined step, stage times."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from constantine_amd import DeviceMsm  # noqa: E402
from constantine_amd.msm import CURVES, CachedBases  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_381_g1"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cs = [int(x) for x in sys.argv[3:]] or [0]
info = CURVES[curve]
n = 1 << log2n
eng = DeviceMsm(0)
if os.environ.get("TABLE_K"):
    eng.set_option("K", int(os.environ["TABLE_K"]))   # entries per accumulate lane (0 = from the resident lanes)
d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
eng.gen_points(curve, 0x5EED0002, n, d_points)
d_scal = torch.from_numpy(synth_scalars(0x5EED0003, n, info.scalar_bits)).cuda()
torch.cuda.synchronize()
ref = None
for c in [-1] + cs:   # -1: plain records
    t0 = time.perf_counter()
    bases = CachedBases(curve, d_points, ctx=eng.ctx, on_device=True, table=c >= 0, window_bits=max(c, 0))
    build = (time.perf_counter() - t0) * 1e3

    def run(k):
        r = None
        pend = bases.submit(d_scal, n)
        for i in range(k):
            nxt = bases.submit(d_scal, n) if i + 1 < k else None
            r = bases.finish(pend, coord="aff")
            pend = nxt
        return r
    eng.enable_timings(False)
    run(3)
    eng.sync()
    t0 = time.perf_counter()
    r = run(20)
    eng.sync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    eng.enable_timings(True)
    bases.msm(d_scal, coord="aff")
    st = eng.last_timings()
    if ref is None:
        ref = bytes(r)
    print(json.dumps({"curve": curve, "log2n": log2n, "table": c >= 0, "c": bases.window_bits, "build_ms": round(build, 1),
                      "ms_per_step": round(ms, 4), "Mpairs_s": round(n / ms / 1e3, 1), "same": bytes(r) == ref, "K_opt": os.environ.get("TABLE_K", ""),
                      "plan": eng.last_plan(), "stage_ms_blocking": {k: round(v, 3) for k, v in st.items()}}), flush=True)
    bases.close()
