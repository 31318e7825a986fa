"""Host-pointer calls of the Constantine symbol at sizes around the slice-count thresholds (2^18 .. 2^20 pairs): nine calls per size, ms each.
$CTT_MSM_HIP_LIB selects another library (round 6: 2^19 measured 2.87-2.94 ms with this library and with round 5's)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from constantine_amd import CURVES, DeviceMsm, multiScalarMul_vartime_parallel
from constantine_amd.synth import synth_scalars
from constantine_amd import _lib
name = "bls12_381_g1"; info = CURVES[name]
for n in (1 << 18, 3 << 17, 1 << 19, 3 << 18, 1 << 20):
    eng = DeviceMsm(0)
    d = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(name, 5, n, d); pts = d.cpu().numpy(); eng.close()
    sc = synth_scalars(6, n, 255)
    ts = []
    for i in range(9):
        t0 = time.perf_counter(); multiScalarMul_vartime_parallel(None, name, sc, pts, coord="jac"); ts.append((time.perf_counter() - t0) * 1e3)
    print(os.path.basename(os.environ.get("CTT_MSM_HIP_LIB", "in-tree")), "n =", n, "ms:", " ".join(f"{t:.2f}" for t in ts), flush=True)
