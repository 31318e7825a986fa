"""Does the chip clock differently with MSMs kept in flight than with one blocking call after another?  (EXPERIMENTS R6.8: Pallas / Vesta at 2^22 are
slower pipelined than blocking.)  Runs each mode for a few seconds while a thread polls rocm-smi for the shader clock and the socket power:

    python tools/clock_probe.py <curve> <log2n> [seconds=6]"""
import collections
import json
import os
import re
import subprocess
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from constantine_amd import DeviceMsm  # noqa: E402
from constantine_amd.msm import CURVES  # noqa: E402
from constantine_amd.synth import synth_scalars  # noqa: E402


def poll(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(txt)
            card = next(iter(d.values()))
            sclk = next((v for k, v in card.items() if k.startswith("sclk")), "")
            m = re.search(r"(\d+)\s*Mhz", str(sclk), re.I)
            pw = next((v for k, v in card.items() if "Power" in k and "W" in k), None)
            out.append((int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None))
        except Exception as e:  # noqa: BLE001
            out.append((None, None))
        time.sleep(0.2)


def main():
    curve, log2n = sys.argv[1], int(sys.argv[2])
    secs = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
    n = 1 << log2n
    info = CURVES[curve]
    eng = DeviceMsm(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    pts = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(curve, 0x5EED0002, n, pts)
    sc = torch.from_numpy(synth_scalars(0x5EED0003, n, info.scalar_bits)).cuda()
    torch.cuda.synchronize()
    eng.enable_timings(False)
    for mode in ("blocking", "pipelined", "blocking", "pipelined"):
        samples, stop = [], threading.Event()
        th = threading.Thread(target=poll, args=(stop, samples))
        th.start()
        t0 = time.perf_counter()
        k = 0
        if mode == "blocking":
            while time.perf_counter() - t0 < secs:
                eng.msm(curve, sc, pts, n, coord="aff")
                k += 1
        else:
            pend = collections.deque()
            while time.perf_counter() - t0 < secs:
                while len(pend) < 2:
                    pend.append(eng.submit(curve, sc, pts, n))
                eng.finish(pend.popleft(), coord="aff")
                k += 1
            while pend:
                eng.finish(pend.popleft(), coord="aff")
                k += 1
        eng.sync()
        dt = time.perf_counter() - t0
        stop.set()
        th.join()
        clk = [s[0] for s in samples[2:] if s[0]]
        pw = [s[1] for s in samples[2:] if s[1]]
        print(json.dumps({"curve": curve, "log2n": log2n, "mode": mode, "ms_per_msm": round(dt / k * 1e3, 4), "msms": k,
                          "sclk_mhz_mean": round(sum(clk) / len(clk)) if clk else None, "sclk_mhz_min_max": [min(clk), max(clk)] if clk else None,
                          "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "samples": len(samples)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
