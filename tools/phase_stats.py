#!/usr/bin/env python3
"""8-byte phase of the multiply-adds in an accumulate kernel's code object (round 6: the BLS12-381 G2 kernel, one wave per SIMD, runs 10 %
slower whenever a change shifts its instruction stream by an odd number of dwords -- 73 % of its v_mad_u64_u32 at addresses = 0 mod 8 in
the fast builds, 27 % in the slow ones; profiles/g2_gather_r06.txt):

    python tools/phase_stats.py <object.o>[=tag] ...      KEY=<substring of the kernel's mangled name> (default k_accumINS_3Fp2), INTO=1 for the <.., true> form

Prints, per object: instructions in the kernel, v_mad_u64_u32 at 0 / 4 mod 8, all 8-byte instructions at 0 / 4 mod 8."""
import collections
import os
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin"
KEY = os.environ.get("KEY", "k_accumINS_3Fp2")
FORM = "Lb1E" if os.environ.get("INTO") == "1" else "Lb0E"


def stats(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "x.fat"), os.path.join(d, "x.co")
        # (--dump-section with an OUTPUT file: llvm-objcopy rewrites its input in place when none is given)
        subprocess.run([f"{B}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(d, "copy.o")], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([f"{B}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        dis = subprocess.run([f"{B}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout.splitlines()
    start = next((i for i, l in enumerate(dis) if l.endswith(">:") and KEY in l and FORM in l), None)
    if start is None:
        return None
    addrs = []
    for l in dis[start + 1:]:
        if l.endswith(">:"):
            break
        m = re.match(r"\s+(\S+)\s+.*//\s*([0-9A-Fa-f]+):", l)
        if m:
            addrs.append((int(m.group(2), 16), m.group(1)))
    mad, all8 = collections.Counter(), collections.Counter()
    windows = collections.defaultdict(lambda: [0, 0])      # per 1000 instructions: multiply-adds at 0 / 4 mod 8
    for i, ((a, op), (b, _)) in enumerate(zip(addrs, addrs[1:])):
        if b - a == 8:
            all8[a % 8] += 1
            if op == "v_mad_u64_u32":
                mad[a % 8] += 1
                windows[i // 1000][0 if a % 8 == 0 else 1] += 1
    return {"instructions": len(addrs), "mad_at_0": mad[0], "mad_at_4": mad[4], "aligned_frac": mad[0] / max(1, mad[0] + mad[4]),
            "all8_at_0": all8[0], "all8_at_4": all8[4], "windows": [tuple(windows[k]) for k in sorted(windows)],
            "best_by_window": sum(max(v) for v in windows.values()) / max(1, mad[0] + mad[4])}


if __name__ == "__main__":
    for a in sys.argv[1:]:
        obj, _, tag = a.partition("=")
        s = stats(obj)
        if s is None:
            print(f"{tag or obj}: no kernel matching {KEY} / {FORM}")
            continue
        print(f"{(tag or os.path.basename(obj)):28s} {s['instructions']:6d} instructions; v_mad_u64_u32 at 0 mod 8: {s['mad_at_0']:6d} ({100 * s['aligned_frac']:.0f} %), at 4 mod 8: {s['mad_at_4']:6d};"
              f" all 8-byte instructions {s['all8_at_0']} / {s['all8_at_4']}")
        if os.environ.get("WINDOWS") == "1":
            print("    per 1000 instructions, multiply-adds at 0 / 4 mod 8: " + " ".join(f"{a}/{b}" for a, b in s["windows"]) +
                  f"   (choosing the phase per window would give {100 * s['best_by_window']:.0f} %)")
