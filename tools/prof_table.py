"""rocprof target: a few window-table MSMs (python tools/prof_table.py curve log2n)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from constantine_amd import DeviceMsm
from constantine_amd.msm import CURVES, CachedBases
from constantine_amd.synth import synth_scalars
curve, log2n = sys.argv[1], int(sys.argv[2])
info = CURVES[curve]; n = 1 << log2n
eng = DeviceMsm(0)
d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
eng.gen_points(curve, 0x5EED0002, n, d_points)
d_scal = torch.from_numpy(synth_scalars(0x5EED0003, n, info.scalar_bits)).cuda()
bases = CachedBases(curve, d_points, ctx=eng.ctx, on_device=True, table=True)
for _ in range(6):
    bases.msm(d_scal, coord="aff")
bases.close()
