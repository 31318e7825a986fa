#!/usr/bin/env python3
"""
bench.py -- MSM throughput on MI355X (BASELINE.json metric: MSM points/sec, BLS12-381 G1, 2^20 random pairs).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8 --steps 50 --warmup 5          (starts its own ranks when WORLD_SIZE is unset)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --total-log2n 24                (BASELINE configs[3]: 2^24 pairs over 8 GPUs, strong scaling)

A "step" is one complete MSM (digits -> sort -> bucket accumulation -> bucket reduction -> window combine ->
affine result on the host) over synthetic inputs that are ALREADY RESIDENT IN HBM when the timed region starts.
With N GPUs the job is ONE MSM sharded by points exactly like the reference's msm-level split
(ec_multi_scalar_mul_parallel.nim:386-431); each step ends with an all_gather of one affine point per rank over RCCL and the
host-side sum of the partials.  `--gpus N` with no size flag is the BASELINE metric: 2^20 pairs IN TOTAL (strong scaling,
2^20 / N pairs per rank) as `value`, with the weak form (2^20 pairs per GPU) and, at N = 8, BASELINE configs[3] (2^24 pairs in
total) timed in the same run and reported as extra keys of the same line.  --total-log2n T times 2^T pairs in total only,
--log2n L (with N > 1) 2^L pairs per GPU only (weak scaling).

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).  Beside the pipelined `value` the line
carries, at N = 1, `latency_ms_blocking` (median wall time of single blocking device-resident calls -- the reference
bench's definition, one call per iteration, benchmarks/bench_ec_msm_bls12_381_g1.nim:43) and `hostptr_ms` (the same
through the Constantine symbol on host pointers, PCIe included).
"""
import argparse
import collections
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

# A process gets four hardware queues by default; the engine's context uses three streams (main, tail, copy) beside the null stream, and
# torch / RCCL bring their own.  Streams beyond the fourth share a hardware queue with another one -- measured: a fifth stream cost every
# small pipelined MSM 8 % without a kernel on it (EXPERIMENTS.md II section 5).  Must be set before the HIP runtime initialises; neutral for one rank.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# v_mad_u64_u32 per mixed addition of the accumulate kernel, as executed (ISA count of the hot path, DESIGN.md 3.3 / EXPERIMENTS.md II 4.3):
# 8 products + 2 squares + 9 Montgomery reductions on NL carry-free limbs = 8 NL^2 + NL (NL + 1) + 9 NL^2 (NL = 14 for
# BLS12-381, 9 for BN254); the Pasta primes have three zero limbs and p_0 = 1, which drop products: 1224.
# G2: the same formula over Fp2 -- a product is 4 base products + 2 reductions, a square 2 + 2.
MADS_PER_MIXED_ADD = {"bls12_381_g1": 3542, "bn254_snarks_g1": 1467, "pallas": 1224, "vesta": 1224,
                      "bls12_381_g2": 8 * (4 * 196 + 2 * 196) + 2 * (2 * 196 + 2 * 196)}
# v_mad_u64_u32 lane-ops/s of the whole chip: profiles/microbench_isa_r02.jsonl (tools/microbench_isa.hip: one asm block of 32
# instructions per loop body, 8 waves per SIMD; 33.3 T at 2 waves per SIMD).  Round 1 used 31 T, taken from its multiplier chain.
INT_MAD_PEAK = 34.5e12
INT_MAD_PEAK_R01 = 31.0e12
BYTES_PER_PAIR = {"bls12_381_g1": 128, "bn254_snarks_g1": 96, "pallas": 96, "vesta": 96, "bls12_381_g2": 224}
BASELINE_CONFIG = {("bls12_381_g1", 20, 1): "configs[1]", ("bn254_snarks_g1", 22, 1): "configs[2]",
                   ("bls12_381_g2", 20, 1): "configs[4]", ("pallas", 20, 1): "configs[4]", ("vesta", 20, 1): "configs[4]"}


# What the reference publishes for this path (terminal screenshots linked from README-PERFORMANCE.md:89-97,
# other hardware): printed beside `cpu_baseline` as external reference lines, never as a baseline of this box.
PUBLISHED_REFERENCE = [
    {"curve": "bls12_381_g1", "n": 1 << 10, "points_per_s": 0.617e6, "threads": 16, "hw": "AMD Ryzen 7 7840U (8C/16T laptop)"},
    {"curve": "bls12_381_g1", "n": 1 << 16, "points_per_s": 1.28e6, "threads": 16, "hw": "AMD Ryzen 7 7840U (8C/16T laptop)"},
    {"curve": "bls12_381_g1", "n": 1 << 18, "points_per_s": 1.62e6, "threads": 16, "hw": "AMD Ryzen 7 7840U (8C/16T laptop)"},
    {"curve": "bn254_snarks_g1", "n": 1 << 16, "points_per_s": 3.42e6, "threads": 36, "hw": "Intel i9-9980XE (18C/36T, 4.1 GHz)"},
    {"curve": "bn254_snarks_g1", "n": 1 << 22, "points_per_s": 6.04e6, "threads": 36, "hw": "Intel i9-9980XE (18C/36T, 4.1 GHz)"},
]


def reference_toolchain():
    """probe for Nim.  Constantine itself is only timed when `nim` and `nimble` AND a checkout
    of the reference ($CTT_REFERENCE_DIR, default /root/reference -- absent on the GPU box) are all present; otherwise the port is."""
    import shutil
    nim, nimble = shutil.which("nim"), shutil.which("nimble")
    ref = os.environ.get("CTT_REFERENCE_DIR", "/root/reference")
    have_ref = os.path.exists(os.path.join(ref, "constantine.nimble"))
    return {"nim": nim, "nimble": nimble, "reference_checkout": ref if have_ref else None,
            "can_run_reference": bool(nim and nimble and have_ref)}


def run_reference_bench(ref_dir, curve):
    """`CC=clang nimble bench_ec_msm_<curve>` of the reference itself (constantine.nimble:1109-1113); returns its raw output lines
    that carry a multi-scalar-mul timing.  Only reached when reference_toolchain()["can_run_reference"]."""
    task = {"bls12_381_g1": "bench_ec_msm_bls12_381_g1", "bn254_snarks_g1": "bench_ec_msm_bn254_snarks_g1"}.get(curve)
    if not task:
        return None
    env = dict(os.environ, CC=os.environ.get("CC", "clang"))
    try:
        out = subprocess.run(["nimble", task], cwd=ref_dir, env=env, capture_output=True, text=True, timeout=1800).stdout
    except Exception as e:   # noqa: BLE001
        return [f"nimble {task} failed: {e}"]
    return [ln.strip() for ln in out.splitlines() if "multi-scalar-mul" in ln.lower() or "msm" in ln.lower()][:40]


def cpu_only_leg(args):
    """BASELINE.json configs[0]: bench_ec_msm_bls12_381_g1.nim, 2^10 random (scalar,point) pairs, CPU threadpool reference --
    plumbing, no GPU.  Times the oracle port (oracle/msm_ref.cpp, reference algorithm incl. its window choice) on this host's
    cores and checks it bit-for-bit against the big-integer oracle (oracle/pyoracle.py).  Prints one JSON line."""
    from oracle import cref
    from oracle import pyoracle as po
    curve, n = args.curve, 1 << args.log2n
    info = po.CURVES[curve]
    seed = 0x5EED0000 + 1
    pts = cref.gen_points(curve, seed, n)
    sc = cref.synth_scalars(seed + 1, n, info.scalar_bits)
    budget = host_cpu_budget()
    cores = min(os.cpu_count() or 1, 2 * budget)
    flags = cref.build_native()
    iters = max(1, 10000 // n)           # the reference bench's iteration count (bench_elliptic_parallel_template.nim:157-164)
    runs, runs1 = [], []
    for _ in range(5):
        t1 = time.perf_counter()
        for _ in range(iters):
            exp, c_used = cref.msm(curve, sc, pts, nthreads=cores)
        runs.append((time.perf_counter() - t1) / iters)
        t1 = time.perf_counter()
        for _ in range(iters):
            exp1, _ = cref.msm(curve, sc, pts, nthreads=1)
        runs1.append((time.perf_counter() - t1) / iters)
    m = min(n, 1 << 10)                  # the big-integer oracle is pure Python: bounded sample
    expect = info.msm_pippenger([int.from_bytes(bytes(x), "little") for x in sc[:m]], [info.aff_from_bytes(bytes(x)) for x in pts[:m]])
    got_m, _ = cref.msm(curve, sc[:m], pts[:m], nthreads=cores)
    tool = reference_toolchain()
    out = {
        "metric": f"MSM points/sec, {curve}, 2^{args.log2n} random pairs, CPU only (BASELINE.json configs[0])",
        "value": n / statistics.median(runs), "unit": "points/s", "n_gpus": 0, "higher_is_better": True,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{curve} MSM, 2^{args.log2n} pairs, oracle port on the host cores (no GPU)", "seed": seed,
                   "window_bits": c_used, "iters_per_run": iters},
        "cpu_baseline": {"value": n / statistics.median(runs), "unit": "points/s", "cores": cores, "kind": "port",
                         "serial_value": n / statistics.median(runs1),
                         "sample": f"all 2^{args.log2n} pairs, median of 5 runs of {iters} calls, {cores} threads on a {budget}-CPU quota "
                                   f"({cpu_model()}); {flags}; oracle/msm_ref.cpp is a restatement of Constantine's algorithm, not "
                                   "Constantine (its endomorphism pre-split where the reference applies it -- here it does; 64-bit C++ "
                                   "Montgomery arithmetic instead of its assembly)"},
        "parity_port_vs_bigint_oracle": bool(info.aff_from_bytes(bytes(got_m)) == expect),
        "parity_threads_vs_serial": bool(bytes(exp) == bytes(exp1)),
        "reference_toolchain": tool,
        "published_reference": [r for r in PUBLISHED_REFERENCE if r["curve"] == curve],
    }
    if tool["can_run_reference"]:
        out["reference_bench_output"] = run_reference_bench(tool["reference_checkout"], curve)
    print(json.dumps(out), flush=True)


def kernel_sources_digest():
    """sha256 over the HIP sources of the library (constantine_amd/csrc/*.h, *.hip), in name order: what a counter file in
    profiles/ was measured on (tools/pmc_summary.py records the same digest; the GPU box has no .git to ask)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "constantine_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def host_cpu_budget():
    """CPUs this process may actually use: min(affinity, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(per)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / per))))
        except Exception:
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def in_flight_depth(pairs_per_gpu):
    """MSMs the timed loop keeps in flight: 2, and 3 for small MSMs ($CTT_BENCH_DEPTH overrides; the engine takes at most 3).
    Three pay up to 2^16 pairs on every box measured (2^14 -8 ... -11 %, 2^16 -4.5 ... -5.7 %).  At 2^17-2^18 the third MSM is a race: the previous
    MSM's tail runs beside an accumulation that now always follows at once, and when it loses, the first reduction pass waits for it -- 2^17
    0.62 -> 0.60 ms per MSM on two boxes, 0.63 -> 0.72 and 0.68 -> 0.81 on two others (profiles/cu_mask_r06.txt section 4).  Two there."""
    env = os.environ.get("CTT_BENCH_DEPTH")
    if env:
        return max(1, min(3, int(env)))
    return 3 if pairs_per_gpu <= (1 << 16) else 2


def measure_hbm_copy_peak(torch, gib=1, reps=6):
    """'measure with a device copy kernel on the box and use the measured figure alongside the nominal'.  A device-to-device
    copy of `gib` GiB (16-byte vector loads and stores, torch's copy kernel) reads and writes every byte once: bytes moved = 2 x size.
    Returns GB/s (best of `reps`, HIP events on torch's current stream), or None when the memory is not there."""
    try:
        n = (gib << 30) // 16
        a = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        b = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        a.fill_(1.0)
        b.copy_(a)
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None or ms < best else best
        del a, b
        torch.cuda.empty_cache()
        return 2.0 * (gib << 30) / (best * 1e-3) / 1e9
    except Exception:   # noqa: BLE001
        return None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """--gpus N without a launcher: start the N ranks ourselves (same command line the driver would use)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--curve", default="bls12_381_g1")
    ap.add_argument("--log2n", type=int, default=None, help="pairs per GPU = 2^log2n (default 20; with --gpus N > 1 it selects "
                    "weak scaling -- without it the N-GPU line is the strong form, 2^20 pairs in total)")
    ap.add_argument("--total-log2n", type=int, default=0, help="total pairs = 2^this, split over the GPUs (strong scaling; "
                    "BASELINE configs[3] is --gpus 8 --total-log2n 24)")
    ap.add_argument("--cpu-sample-log2", type=int, default=20, help="pairs timed on the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the blocking-call and host-pointer legs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL); gloo is for "
                    "exercising the multi-rank path on a single-GPU box together with --all-ranks-on-device")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1, help="testing: put every rank on this GPU")
    ap.add_argument("--cpu-only", action="store_true", help="BASELINE.json configs[0]: the CPU port against the big-integer oracle, "
                    "no GPU (use with --log2n 10)")
    args = ap.parse_args()
    explicit_log2n = args.log2n is not None
    if args.log2n is None:
        args.log2n = 20

    if args.cpu_only:
        return cpu_only_leg(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.all_ranks_on_device >= 0:
        local_rank = args.all_ranks_on_device
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from constantine_amd import CURVES, DeviceMsm
    from constantine_amd import parallel
    from constantine_amd.synth import synth_scalars

    curve = args.curve
    info = CURVES[curve]
    # which job is `value`?  N = 1: 2^log2n pairs.  N > 1: --total-log2n T -> 2^T pairs in total (strong); --log2n L -> 2^L pairs per
    # GPU (weak); neither -> the BASELINE metric, 2^20 pairs in total (strong), with the weak form and configs[3] as extra legs
    strong = args.total_log2n > 0 or (world > 1 and not explicit_log2n)
    driver_form = world > 1 and not explicit_log2n and args.total_log2n <= 0
    main_total_log2 = args.total_log2n if args.total_log2n > 0 else 20
    seed = 0x5EED0000 + 2  # fixed seed = 0x5EED_0000 + config index
    eng = DeviceMsm(local_rank)
    # the roofline needs the accumulate kernel's time over the timed region: mode 2 records that stage (and the total) only --
    # every event record is a barrier packet in the queue, and all twelve cost a small pipelined MSM 15 % (2^16: 0.63 ms per
    # step against 0.55); the full stage breakdown comes from one blocking call after the timed region (stage_ms_blocking)
    eng.enable_timings(2)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    # while rank 0 runs a leg of its own (the one-GPU reference, the in-library sharding over every GPU of the node) the other ranks
    # wait on the HOST: an RCCL barrier would park a spinning kernel on the very GPUs rank 0's library is about to use
    cpu_group = None
    if world > 1 and args.backend == "nccl":
        try:
            cpu_group = dist.new_group(backend="gloo")
        except Exception:   # noqa: BLE001  (no usable interface for gloo: the RCCL barrier it is)
            cpu_group = None
        # every rank must wait the same way: the gloo group is used only if ALL ranks got one
        have = torch.tensor([1 if cpu_group is not None else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        if int(have.item()) == 0:
            cpu_group = None

    def idle_barrier():
        if world > 1:
            if cpu_group is not None:
                dist.barrier(group=cpu_group)
            else:
                dist.barrier()

    def timed_leg(leg_strong, lg, exchange=True, solo=False):
        """One timed loop: `steps` complete MSMs with two in flight.  leg_strong: 2^lg pairs in total over the ranks, else 2^lg per
        rank.  solo: this rank alone, all 2^lg pairs, no collective (the same-run one-GPU reference of an N-GPU line).
        exchange=False: every rank its shard, no collective and no barrier (what the exchange and the barrier cost)."""
        if solo:
            leg_first, leg_n, leg_total = 0, 1 << lg, 1 << lg
        elif leg_strong:
            leg_total = 1 << lg
            leg_first, leg_n = parallel.shard_bounds(leg_total, world, rank)     # balanced contiguous slices (partitioners.nim:44-77)
        else:
            leg_n = 1 << lg
            leg_total, leg_first = world * leg_n, rank * leg_n
        # ... and below 2^20 pairs only every fourth MSM carries the events (four records cost a 0.5 ms step 12 %: 2^16 0.53 ms per step
        # against 0.47 without); the kernel's average duration is then over those launches of the timed region
        every = 1 if leg_n >= (1 << 20) else 4
        eng.set_option("timings_every", every)
        # ---- synthetic inputs, resident in HBM -----------------------------------------------------------
        d_points = torch.empty((leg_n, info.aff_bytes), dtype=torch.uint8, device="cuda")
        eng.gen_points(curve, seed, leg_n, d_points, first=leg_first)            # P_i = [s_i]G, uniform over the subgroup
        scal = synth_scalars(seed + 1, leg_n, info.scalar_bits, first=leg_first)  # uniform in [0,2^bits), not reduced
        d_scal = torch.from_numpy(scal).cuda()
        torch.cuda.synchronize()
        collective = exchange and not solo
        # N > 1: the all_gather of the partials is started when a step's local MSM is done and completed one step later, so its
        # latency hides under the next MSM (every exchange still starts and ends inside the timed region)
        xchg = parallel.ShardExchange(curve) if collective else None

        # One step = one complete MSM.  Two steps are kept in flight: the GPU work of step i+1 is enqueued before the
        # host tail of step i (Horner over windows, affine normalisation, partial-sum exchange) runs, so the GPU never
        # waits for the CPU.  Every step's result is produced inside the timed region.
        # In flight: two MSMs -- three up to 2^16 pairs per GPU (round 6: the engine has three slots; there the host side of a step, ~0.1 ms of
        # enqueueing and ~0.15 ms of host tail, is a third of the step, and with two in flight submit(i+2) had to wait for finish(i):
        # profiles/cu_mask_r06.txt has the timeline and the A/B; in_flight_depth has why not above 2^16)
        depth = in_flight_depth(leg_n)

        def run_steps(k, acc=None):
            res = None
            queue, submitted = collections.deque(), 0
            in_exchange = None
            for i in range(k):
                while submitted < k and len(queue) < depth:
                    queue.append(eng.submit(curve, d_scal, d_points, leg_n))
                    submitted += 1
                part = eng.finish(queue.popleft(), coord="aff")
                if xchg is None:
                    res = part
                else:
                    started = xchg.start(part)
                    if in_exchange is not None:
                        res = xchg.finish(in_exchange)
                    in_exchange = started
                # (only the sampled steps are asked for their events: the engine's sample counter is reset right before the timed loop, so
                # submit number i carries events iff i is a multiple of `every` -- a last_timings() call per step is ~15 us of Python and
                # ctypes on the thread that feeds a 0.45 ms step)
                if acc is not None and i % every == 0:
                    t = eng.last_timings()                      # HIP events recorded on the engine's stream (zeros: not a sampled step)
                    if t["total"] > 0.0:
                        for key, v in t.items():
                            acc[key] = acc.get(key, 0.0) + v
                        acc["_launches"] = acc.get("_launches", 0) + 1
            if in_exchange is not None:
                res = xchg.finish(in_exchange)
            return res

        def local_fence():
            torch.cuda.synchronize()
            eng.sync()
        fn_fence = fence if collective else local_fence
        run_steps(args.warmup)
        acc = {}
        fn_fence()
        eng.enable_timings(2)            # (resets the engine's sample counter: the timed loop's submit 0 is a sampled one)
        t0 = time.perf_counter()
        res = run_steps(args.steps, acc)
        fn_fence()
        dt = time.perf_counter() - t0
        if collective and world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        launches = int(acc.pop("_launches", 0))
        return {"n": leg_n, "first": leg_first, "total": leg_total, "dt": dt, "res": res, "plan": eng.last_plan(), "depth": depth,
                "stages": {k: v / max(1, launches) for k, v in acc.items()}, "ev_launches": launches,
                "scal": scal, "d_points": d_points, "d_scal": d_scal}

    def ranks_seen():
        # what the collective library reports, not what the launcher asked for: every rank contributes 1
        ones = torch.ones(1, dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        return int(ones.item())

    def leg_summary(leg, leg_strong, lg):
        return {"value": leg["total"] * args.steps / leg["dt"], "unit": "points/s", "ms_per_step": leg["dt"] / args.steps * 1e3,
                "scaling": "strong" if leg_strong else "weak", "total_pairs": leg["total"], "pairs_per_gpu_rank0": leg["n"],
                "window_bits": leg["plan"]["c"], "windows": leg["plan"]["W"], "entries_per_lane": leg["plan"]["K"],
                "rccl_ranks_seen": ranks_seen(),
                "workload": f"{curve} MSM, 2^{lg} pairs " + ("in total" if leg_strong else "per GPU") + f", {world} ranks, inputs resident in HBM, {leg['depth']} MSMs in flight"}

    hbm_measured = measure_hbm_copy_peak(torch) if rank == 0 else None
    lg = main_total_log2 if strong else args.log2n
    leg = timed_leg(strong, lg)
    n, first, total, dt, res, plan, stages = leg["n"], leg["first"], leg["total"], leg["dt"], leg["res"], leg["plan"], leg["stages"]
    scal, d_points, d_scal, ev_launches = leg["scal"], leg["d_points"], leg["d_scal"], leg["ev_launches"]
    value = total * args.steps / dt
    label = BASELINE_CONFIG.get((curve, lg, world)) if not strong else ("configs[3]" if (curve, lg, world) == ("bls12_381_g1", 24, 8) else None)
    if strong and curve == "bls12_381_g1" and lg == 20:
        label = "the metric: 2^20 pairs at 1/2/4/8 GPUs"

    out = {
        "metric": ("MSM points/sec, BLS12-381 G1, 2^20 random pairs" if (curve == "bls12_381_g1" and lg == 20 and (strong or world == 1))
                   else f"MSM points/sec, {curve}, 2^{lg} random pairs")
                  + (f" in total over {world} GPUs" if strong else f" per GPU ({world} x 2^{lg} pairs in one MSM)" if world > 1 else "")
                  + f"; pipelined throughput, {leg['depth']} MSMs in flight, inputs resident in HBM",
        "value": value,
        "value_kind": f"pipelined: steps / wall time with {leg['depth']} complete MSMs in flight (the host tail of MSM i runs under the GPU work of "
                      "the following ones); `value_blocking` = N / latency of ONE blocking call (SURVEY 8d's definition, the reference bench's), "
                      "`value_hostptr` = the same through the Constantine symbol on pageable host arrays (PCIe included)",
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": ("strong" if strong else "weak") if world > 1 else "none",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": (f"{curve} MSM, 2^{lg} (scalar,point) pairs " + ("in total" if strong else "per GPU")
                         + f", inputs resident in HBM, {leg['depth']} MSMs in flight" + (f" (BASELINE.json {label})" if label else "")),
            "pairs_per_gpu": n, "total_pairs": total, "ranks": world, "scalar_bits": info.scalar_bits,
            "window_bits": plan["c"], "windows": plan["W"], "entries_per_lane": plan["K"],
            "sharding": f"points x{world}, asynchronous all_gather of one affine point per rank ({args.backend}), completed one step later, + host sum" if world > 1 else "none",
            "seed": seed,
        },
        "stage_ms": stages,
        "stage_ms_note": f"HIP events around the accumulate kernel (and the whole MSM) on {ev_launches} of the {args.steps} timed steps",
    }

    if world > 1:
        # what the collective library reports, not what the launcher asked for: the first RCCL/gloo collective of the run was the
        # barrier of the first fence(); here every rank contributes 1 and rank 0 prints the sum next to the backend's own count
        out["collective"] = {"backend": dist.get_backend(), "rccl_ranks_seen": ranks_seen(), "world_size": dist.get_world_size(),
                             "devices": torch.cuda.device_count(), "rank0_device": torch.cuda.get_device_name(local_rank)}
        # ---- what bounds the split (same run, same boxes): (a) every rank its shard with no exchange and no barrier -- what the
        # all_gather of one point per rank and the host sum cost on top of the slowest rank's MSMs; (b) rank 0 alone on ALL the pairs --
        # the one-GPU figure the N-GPU line is a speed-up over (the other ranks wait at the barrier)
        if strong:
            own = timed_leg(True, lg, exchange=False)
            t = torch.tensor([own["dt"]], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            own_ms = float(t.item()) / args.steps * 1e3
            solo_ms = None
            if rank == 0 and lg <= 22:
                solo = timed_leg(True, lg, solo=True)
                solo_ms = solo["dt"] / args.steps * 1e3
            idle_barrier()
            out["strong_bound"] = {
                "ms_per_step_shards_without_exchange": own_ms,
                "ms_per_step_one_gpu_same_run": solo_ms,
                "speedup_vs_one_gpu_same_run": (solo_ms / (dt / args.steps * 1e3)) if solo_ms else None,
                "speedup_bound_from_shard_time": (solo_ms / own_ms) if solo_ms else None,
                "note": "2^%d / %d pairs per rank: a small MSM is a chain of dependent launches (DESIGN.md section 5: 2.8 / 1.7 / 1.1 / 0.70 ms for "
                        "2^20 / 2^19 / 2^18 / 2^17 pairs on one GPU), so the split of a FIXED 2^20-pair job is bounded well below N x" % (lg, world)}
        # ---- the other forms of the same job, same run (driver form only: `--gpus N` with no size flag) --------------------------
        if driver_form:
            w = timed_leg(False, 20)
            out["weak_2pow20_per_gpu"] = leg_summary(w, False, 20)
            if world == 8 and curve == "bls12_381_g1":
                c3 = timed_leg(True, 24)
                out["configs3_2pow24_total"] = leg_summary(c3, True, 24)
                out["configs3_2pow24_total"]["baseline_config"] = "BASELINE.json configs[3]"
        # ---- the in-library form: ONE process calls the Constantine symbol on host arrays, the library shards the call over the GPUs
        # (ctt_hip_msm_set_devices; what a Constantine caller on an 8-GPU node gets without a source change).  Rank 0 only, the other
        # ranks wait; PCIe included, never `value`.
        if strong and not args.no_latency:
            if rank == 0:
                from constantine_amd import multiScalarMul_vartime, multiScalarMul_vartime_parallel
                from constantine_amd.msm import set_devices
                full = 1 << lg
                h_points = torch.empty((full, info.aff_bytes), dtype=torch.uint8, device="cuda")
                eng.gen_points(curve, seed, full, h_points, first=0)
                pts_host = h_points.cpu().numpy()
                sc_host = synth_scalars(seed + 1, full, info.scalar_bits, first=0)
                fn = (lambda s_, p_: multiScalarMul_vartime_parallel(None, curve, s_, p_, coord="jac")) if info.has_parallel \
                    else (lambda s_, p_: multiScalarMul_vartime(curve, s_, p_, coord="jac"))
                res_h = {}
                # (testing form, --all-ranks-on-device d: "all devices" = d listed once per rank -- the library's sharding over `world` contexts of one GPU)
                node = list(range(torch.cuda.device_count())) if args.all_ranks_on_device < 0 else [local_rank] * world
                for tag, devs in (("one_device", [local_rank]), ("all_devices", node)):
                    set_devices(devs if len(devs) > 1 else [])
                    hp = []
                    for _ in range(7):
                        t1 = time.perf_counter()
                        fn(sc_host, pts_host)
                        hp.append((time.perf_counter() - t1) * 1e3)
                    res_h[tag] = statistics.median(hp[2:])
                set_devices([])
                torch.cuda.set_device(local_rank)
                out["hostptr_sharded_ms"] = {"one_device": res_h["one_device"], "all_devices": res_h["all_devices"],
                                             "devices": len(node), "distinct_devices": len(set(node)),
                                             "note": f"median of 5 calls of the Constantine symbol on pageable host arrays, 2^{lg} pairs, from ONE process "
                                                     "(rank 0; the other ranks idle): the library's own sharding over the node's GPUs, PCIe included"}
            idle_barrier()

    if rank == 0:
        # ---- roofline of the dominant kernel (bucket accumulation, k_accum); N > 1: rank 0's launches (every rank runs the same
        # kernel over its own shard) -----------------------------
        t_acc = stages.get("accumulate", 0.0) * 1e-3
        alg_bytes = n * BYTES_PER_PAIR.get(curve, 128)  # N x (scalar + affine point), one launch = all windows
        achieved = alg_bytes / t_acc / 1e9 if t_acc > 0 else 0.0
        # HBM traffic of the kernel: rocprofv3 --pmc passes of the same command (tools/collect_round.sh), kept in profiles/ -- counters
        # cannot be collected from inside the run they annotate.  The file records the digest of the kernel sources it was measured
        # on; a figure measured on other sources is refused (traffic: null) rather than quoted.
        traffic, traffic_src = None, None
        tr_path = os.path.join(ROOT, "profiles", "hbm_traffic_k_accum.json")
        if os.path.exists(tr_path):
            try:
                doc = json.load(open(tr_path))
                key = f"{curve}_2^{int(np.log2(n)) if n & (n - 1) == 0 else -1}"
                have, now = doc.get("_sources_sha256"), kernel_sources_digest()
                if key in doc and have == now:
                    traffic = doc[key]
                    traffic_src = (f"profiles/hbm_traffic_k_accum.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on the "
                                   f"same kernel sources (sha256 {now[:12]}, collected {doc.get('_collected', '?')}), not measured in this run")
                elif key in doc:
                    traffic_src = (f"profiles/hbm_traffic_k_accum.json holds {doc[key]} B per launch, measured on kernel sources {str(have)[:12]} -- "
                                   f"this tree is {now[:12]}: refused as stale")
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_accum", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # the same fraction against what a copy kernel reaches on THIS box, measured at the start of this run
            "peak_measured": hbm_measured, "frac_measured": (achieved / hbm_measured) if hbm_measured else None,
            "peak_measured_note": "device-to-device copy of 1 GiB (16-byte vectors, read + write counted), best of 6, HIP events; nominal peak beside it",
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel_ms": stages.get("accumulate", 0.0), "algorithmic_bytes": alg_bytes,
            "note": "integer-VALU bound by construction (no dense contraction); the binding roof is `int_mad` below",
        }
        if curve in MADS_PER_MIXED_ADD and t_acc > 0:
            mads = plan["W"] * n * MADS_PER_MIXED_ADD[curve]      # one mixed addition per (window, pair)
            out["roofline"]["int_mad"] = {
                "instr": "v_mad_u64_u32", "per_launch": mads, "achieved": mads / t_acc / 1e12, "peak": INT_MAD_PEAK / 1e12,
                "unit": "T lane-ops/s", "frac": mads / t_acc / INT_MAD_PEAK,
                "frac_vs_round1_peak_31T": mads / t_acc / INT_MAD_PEAK_R01,
                "note": "the multiply-adds are ~78 % of the kernel's VALU instructions; every VOP3 instruction issues at the same "
                        "~4.5 cycles per wave (profiles/microbench_isa_r02.jsonl), so the kernel's own roof is its instruction "
                        "count: profiles/pmc_r06_sq_counters_k_accum_*.txt: 4558 VALU instructions per mixed addition for 3542 multiply-adds (DESIGN.md 3.3; EXPERIMENTS.md II 4.3 has the account per class)",
            }

    # ---- the reference bench's own definition: one blocking call per iteration ----------------------------
    if world == 1 and not args.no_latency:
        lat = []
        eng.enable_timings(False)   # the reference's bench times the bare call
        for _ in range(12):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            got_blocking = eng.msm(curve, d_scal, d_points, n, coord="aff")
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = lat[2:]
        out["latency_ms_blocking"] = statistics.median(lat)
        out["value_blocking"] = n / statistics.median(lat) * 1e3
        out["latency_note"] = ("median of 10 single blocking ctt_hip_msm_device calls after 2 warm-ups, inputs resident in HBM "
                               f"(min {min(lat):.3f}, max {max(lat):.3f}); points/s at this latency = {n / statistics.median(lat) * 1e3:.4g}")
        eng.enable_timings(True)
        eng.msm(curve, d_scal, d_points, n, coord="aff")
        out["stage_ms_blocking"] = eng.last_timings()   # of one more call, with the stage events on
        eng.enable_timings(False)
        # the drop-in symbol itself: host pointers in, PCIe included (never `value`)
        from constantine_amd import multiScalarMul_vartime, multiScalarMul_vartime_parallel
        fn = (lambda s, p: multiScalarMul_vartime_parallel(None, curve, s, p, coord="jac")) if info.has_parallel \
            else (lambda s, p: multiScalarMul_vartime(curve, s, p, coord="jac"))
        pts_host = d_points.cpu().numpy()
        hp = []
        for _ in range(7):
            t1 = time.perf_counter()
            fn(scal, pts_host)
            hp.append((time.perf_counter() - t1) * 1e3)
        hp = hp[2:]
        out["hostptr_ms"] = statistics.median(hp)
        out["value_hostptr"] = n / statistics.median(hp) * 1e3
        out["hostptr_note"] = (f"median of 5 calls of ctt_{info.sym}_jac_multi_scalar_mul_big_coefs_vartime"
                               f"{'_parallel' if info.has_parallel else ''} on pageable host arrays after 2 warm-ups "
                               f"(H2D of {n * (32 + info.aff_bytes) >> 20} MiB included): {n / statistics.median(hp) * 1e3:.4g} points/s")

    # ---- cached bases with a window table (the ZAL base descriptor / a KZG SRS: bases reused across MSMs) ----------
    # Not `value`: the headline hands the points over per call, as the reference's bench does.  Here the bases are
    # prepared once (CachedBases(table=True): multiples 2^(c*w) * P of every base resident in HBM) and each step moves
    # nothing but works on the resident table.
    if world == 1 and not args.no_latency and n <= (1 << 22):
        from constantine_amd.msm import CachedBases
        cb = {}
        for label, table in (("records", False), ("window_table", True)):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            bases = CachedBases(curve, d_points, ctx=eng.ctx, on_device=True, table=table)
            build_ms = (time.perf_counter() - t1) * 1e3

            def run_cached(k):
                r = None
                pend = bases.submit(d_scal, n) if k > 0 else None
                for i in range(k):
                    nxt = bases.submit(d_scal, n) if i + 1 < k else None
                    r = bases.finish(pend, coord="aff")
                    pend = nxt
                return r
            run_cached(3)
            fence()
            t1 = time.perf_counter()
            r_cb = run_cached(args.steps)
            fence()
            ms = (time.perf_counter() - t1) / args.steps * 1e3
            plan_cb = eng.last_plan()
            lat_cb = []
            for _ in range(7):
                t1 = time.perf_counter()
                bases.msm(d_scal, coord="aff")
                lat_cb.append((time.perf_counter() - t1) * 1e3)
            cb[label] = {"ms_per_step": ms, "pairs_per_s": n / ms * 1e3, "latency_ms_blocking": statistics.median(lat_cb[2:]),
                         "window_bits": plan_cb["c"], "windows": plan_cb["W"], "build_ms": build_ms,
                         "same_result_as_headline": bool(bytes(r_cb) == bytes(res))}
            bases.close()
        out["cached_bases"] = cb

    if rank == 0:
        # ---- CPU baseline: the oracle port on the host cores, bounded sample; doubles as a parity check -----
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cref   # the checker: imported for this leg only
            m = min(n, 1 << args.cpu_sample_log2)
            budget = host_cpu_budget()           # the GPU box caps the container at a CPU quota
            cores = min(os.cpu_count() or 1, 2 * budget)  # 2 threads per granted CPU balances the window tasks best
            flags = cref.build_native()          # same source rebuilt for this host's CPU (g++ -march=native), else the shipped build
            pts_m = d_points[:m].cpu().numpy()
            runs = []
            for _ in range(3):
                t1 = time.perf_counter()
                exp, c_used = cref.msm(curve, scal[:m], pts_m, nthreads=cores)
                runs.append(time.perf_counter() - t1)
            cpu_dt = statistics.median(runs)
            got = eng.msm(curve, d_scal[:m], d_points[:m], m, coord="aff")
            out["cpu_baseline"] = {
                "value": m / cpu_dt, "unit": "points/s", "cores": cores, "kind": "port",
                "sample": f"first 2^{int(np.log2(m))} pairs of the same workload, oracle/msm_ref.cpp "
                          f"(restatement of Constantine's Pippenger with its batched-affine buckets and its endomorphism pre-split "
                          f"where the reference's dispatch applies it -- not at this size --, not Constantine: C++ instead of its assembly), c={c_used}, median of 3 runs "
                          f"({', '.join(f'{r:.2f}' for r in runs)} s wall), "
                          f"{cores} threads on a {budget}-CPU cgroup quota ({os.cpu_count()} logical CPUs visible, {cpu_model()}); {flags}",
            }
            out["parity_vs_oracle_on_sample"] = bool(bytes(got) == bytes(exp))
            # the WHOLE timed workload, without the port: the points are [s_i]G with known s_i, so the MSM is [sum a_i s_i mod r]G --
            # an exact integer dot product and one scalar multiplication of the big-integer oracle (cref.msm_by_discrete_logs)
            from oracle import pyoracle as po
            t1 = time.perf_counter()
            want = cref.msm_by_discrete_logs(curve, seed, scal, first=first)
            out["parity_full_size_vs_discrete_logs"] = bool(po.CURVES[curve].aff_from_bytes(bytes(res)) == want)
            out["parity_full_size_note"] = (f"the timed loop's result for all 2^{args.log2n} pairs against [sum a_i s_i mod r]G "
                                            f"(big-integer oracle, {time.perf_counter() - t1:.1f} s; no bucket method involved)")
            tool = reference_toolchain()
            out["cpu_baseline"]["reference_toolchain"] = tool
            out["cpu_baseline"]["published_reference"] = [r for r in PUBLISHED_REFERENCE if r["curve"] == curve]
            if tool["can_run_reference"] and os.environ.get("CTT_RUN_REFERENCE_BENCH") == "1":
                # Constantine itself (minutes of Nim compilation): opt-in, so that the default run still finishes in minutes
                out["cpu_baseline"]["reference_bench_output"] = run_reference_bench(tool["reference_checkout"], curve)
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
