#!/usr/bin/env python3
"""
bench.py -- MSM throughput on MI355X (BASELINE.json metric: MSM points/sec, BLS12-381 G1, 2^20 random pairs).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one complete MSM (digits -> sort -> bucket accumulation -> bucket reduction -> window combine ->
affine result on the host) over synthetic inputs that are ALREADY RESIDENT IN HBM when the timed region starts.
With N GPUs every rank owns 2^20 pairs (weak scaling: the job is one MSM over N*2^20 pairs, sharded by points
exactly like the reference's msm-level split, ec_multi_scalar_mul_parallel.nim:386-431); each step ends with an
all_gather of one affine point per rank over RCCL and the host-side sum of the partials.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# v_mad_u64_u32 per mixed addition of the accumulate kernel: 8 products + 2 squares + 9 Montgomery reductions on NL
# carry-free limbs = 8 NL^2 + NL (NL + 1) + 9 NL^2 (NL = 14 for BLS12-381, 9 for the 254/255-bit fields); DESIGN.md 4.3
MADS_PER_MIXED_ADD = {"bls12_381_g1": 3542, "bn254_snarks_g1": 1467, "pallas": 1467, "vesta": 1467}
INT_MAD_PEAK = 31.0e12         # v_mad_u64_u32 lane-ops/s, measured on MI355X (profiles/microbench_fpu_r01.jsonl: 79.2 G products/s x 393)
BYTES_PER_PAIR = {"bls12_381_g1": 128, "bn254_snarks_g1": 96, "pallas": 96, "vesta": 96, "bls12_381_g2": 224}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--curve", default="bls12_381_g1")
    ap.add_argument("--log2n", type=int, default=20, help="pairs per GPU = 2^log2n")
    ap.add_argument("--cpu-sample-log2", type=int, default=20, help="pairs timed on the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL); gloo is for "
                    "exercising the multi-rank path on a single-GPU box together with --all-ranks-on-device")
    ap.add_argument("--all-ranks-on-device", type=int, default=-1, help="testing: put every rank on this GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} (got {world})")
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
    from oracle import cref  # used ONLY for synthetic scalars (numpy helper) and the cpu_baseline / parity legs

    curve = args.curve
    info = CURVES[curve]
    n = 1 << args.log2n
    seed = 0x5EED0000 + 2  # fixed seed = 0x5EED_0000 + config index
    eng = DeviceMsm(local_rank)

    # ---- synthetic inputs, resident in HBM -----------------------------------------------------------
    first = rank * n
    d_points = torch.empty((n, info.aff_bytes), dtype=torch.uint8, device="cuda")
    eng.gen_points(curve, seed, n, d_points, first=first)            # P_i = [s_i]G, uniform over the subgroup
    scal = cref.synth_scalars(seed + 1, n, info.scalar_bits, first=first)  # uniform in [0,2^bits), not reduced
    d_scal = torch.from_numpy(scal).cuda()
    torch.cuda.synchronize()

    # One step = one complete MSM.  Two steps are kept in flight: the GPU work of step i+1 is enqueued before the
    # host tail of step i (Horner over windows, affine normalisation, partial-sum exchange) runs, so the GPU never
    # waits for the CPU.  Every step's result is produced inside the timed region.
    def submit():
        return eng.submit(curve, d_scal, d_points, n)

    def finish(ticket):
        return parallel.msm_sharded(curve, lambda: eng.finish(ticket, coord="aff"))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    def run_steps(k, acc=None):
        res = None
        pending = submit() if k > 0 else None
        for i in range(k):
            nxt = submit() if i + 1 < k else None
            res = finish(pending)
            if acc is not None:
                for key, v in eng.last_timings().items():   # HIP events recorded on the engine's stream
                    acc[key] = acc.get(key, 0.0) + v
            pending = nxt
        return res

    run_steps(args.warmup)
    stage_acc = {}
    fence()
    t0 = time.perf_counter()
    res = run_steps(args.steps, stage_acc)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    plan = eng.last_plan()
    stages = {k: v / args.steps for k, v in stage_acc.items()}
    value = world * n * args.steps / dt

    out = {
        "metric": "MSM points/sec, BLS12-381 G1, 2^20 random pairs" if (curve == "bls12_381_g1" and args.log2n == 20)
                  else f"MSM points/sec, {curve}, 2^{args.log2n} random pairs",
        "value": value,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{curve} MSM, 2^{args.log2n} (scalar,point) pairs per GPU, inputs resident in HBM "
                        f"(BASELINE.json configs[1])",
            "pairs_per_gpu": n, "total_pairs": world * n, "scalar_bits": info.scalar_bits,
            "window_bits": plan["c"], "windows": plan["W"], "entries_per_lane": plan["K"],
            "sharding": f"points x{world}" if world > 1 else "none", "seed": seed,
        },
        "stage_ms": stages,
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (bucket accumulation, k_accum) -----------------------------
        t_acc = stages.get("accumulate", 0.0) * 1e-3
        alg_bytes = n * BYTES_PER_PAIR.get(curve, 128)  # N x (scalar + affine point), one launch = all windows
        achieved = alg_bytes / t_acc / 1e9 if t_acc > 0 else 0.0
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "hbm_traffic_k_accum.json")
        if os.path.exists(tr_path):
            try:
                traffic = json.load(open(tr_path)).get(f"{curve}_2^{args.log2n}")
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_accum", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel_ms": stages.get("accumulate", 0.0), "algorithmic_bytes": alg_bytes,
            "note": "integer-VALU bound by construction (no dense contraction); the binding roof is `int_mad` below",
        }
        if curve in MADS_PER_MIXED_ADD and t_acc > 0:
            mads = plan["W"] * n * MADS_PER_MIXED_ADD[curve]      # one mixed addition per (window, pair)
            out["roofline"]["int_mad"] = {
                "instr": "v_mad_u64_u32", "per_launch": mads, "achieved": mads / t_acc / 1e12, "peak": INT_MAD_PEAK / 1e12,
                "unit": "T lane-ops/s", "frac": mads / t_acc / INT_MAD_PEAK,
            }
        # ---- CPU baseline: the oracle port on the host cores, bounded sample; doubles as a parity check -----
        if world == 1 and not args.no_cpu_baseline:
            m = min(n, 1 << args.cpu_sample_log2)
            budget = host_cpu_budget()           # the GPU box caps the container at a CPU quota
            cores = min(os.cpu_count() or 1, 2 * budget)  # 2 threads per granted CPU balances the window tasks best
            pts_host = d_points[:m].cpu().numpy()
            t1 = time.perf_counter()
            exp, c_used = cref.msm(curve, scal[:m], pts_host, nthreads=cores)
            cpu_dt = time.perf_counter() - t1
            got = eng.msm(curve, d_scal[:m], d_points[:m], m, coord="aff")
            out["cpu_baseline"] = {
                "value": m / cpu_dt, "unit": "points/s", "cores": cores, "kind": "port",
                "sample": f"first 2^{int(np.log2(m))} pairs of the same workload, oracle/msm_ref.cpp "
                          f"(restatement of Constantine's Pippenger, not Constantine), c={c_used}, {cpu_dt:.2f} s wall, "
                          f"{cores} threads on a {budget}-CPU cgroup quota ({os.cpu_count()} logical CPUs visible, {cpu_model()}); "
                          f"g++ -O3 -march=x86-64-v3, one run",
            }
            out["parity_vs_oracle_on_sample"] = bool(bytes(got) == bytes(exp))
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
