#!/usr/bin/env python3
"""Two passes over the compiler's own gfx950 device assembly, run between `hipcc -S` and the assembler (with_asm_pass.sh, Makefile):

    python3 asm_pass.py <in.s> <out.s> [--strip-asm-nops <function-substring>]... [--align [<function-substring>]...]

--strip-asm-nops   hipcc pads every inline-asm statement whose result the next instruction reads with one `s_nop 0`: its hazard
    recogniser cannot see inside the statement and assumes a producer with a partial-dword destination (dst_sel forwarding, one wait
    state).  The statements of this library hold full-dword v_mad_u64_u32 / add chains only, so the pad guards nothing.  With two waves
    per SIMD the no-op hides behind the other wave's arithmetic most of the time; removing them measures -1.7 % per MSM for BLS12-381 G1
    at 2^20 and -3.1 % for BN254 at 2^22 (profiles/asm_pass_r06.txt).  Only `s_nop 0` lines that directly follow `;;#ASMEND` inside
    the named functions are removed; no-ops the compiler placed for hazards between its own instructions stay.

--align   keeps every 8-byte instruction on an 8-byte boundary.  With ONE wave per SIMD -- the BLS12-381 G2 kernels, 483 of 512
    registers -- an 8-byte instruction that starts at 4 mod 8 can cost extra issue time; the compiler mixes 4-byte encodings
    (v_mov_b32_e32, v_sub_u32_e32, s_waitcnt ...) into a stream that is 85 % v_mad_u64_u32 (8 bytes), and any unrelated edit that moves
    the kernel's stream by an odd number of dwords made the G2 accumulate kernel 11.5 % slower (EXPERIMENTS.md R6.3).  Wherever a 4-byte
    instruction would leave the next 8-byte instruction (or a label) at 4 mod 8 the pass either re-encodes it in its 8-byte VOP3 form
    (`_e32` -> `_e64`: same operation, same operands) or, when it has none (scalar ops, s_waitcnt, v_accvgpr_mov_b32), puts an
    `s_nop 0` in front of it; two 4-byte instructions in a row are left alone.  Sizes are not guessed from the text: the input is
    assembled once and the sizes are read back from the disassembly, instruction by instruction, mnemonics checked against the text.
    Measured: -1.3 ... -1.5 % per G2 MSM, and the 11.5 % trap is gone (the phase no longer depends on what precedes the kernel).

Both passes change encodings and add / remove no-ops only; the parity suite runs on their output like on any other build, and the
Makefile falls back to the plain compile when any step of the detour fails."""
import os
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin"
ASM = [f"{B}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c"]
# 4-byte VALU encodings without a VOP3 twin that assembles under the same operand text
NO_E64 = {"v_accvgpr_mov_b32", "v_nop", "v_readfirstlane_b32"}


def assemble(src, obj):
    subprocess.run(ASM + [src, "-o", obj], check=True)


def function_sizes(obj):
    """{function: [(mnemonic, size), ...]} from the disassembly of `obj`."""
    dis = subprocess.run([f"{B}/llvm-objdump", "-d", "--no-show-raw-insn", obj], capture_output=True, text=True, check=True).stdout
    out, cur, rows = {}, None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if cur is not None:
                out[cur] = rows
            cur, rows = m.group(1), []
            continue
        m = re.match(r"\s+(\S+)\s*.*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            rows.append((m.group(1), int(m.group(2), 16)))
    if cur is not None:
        out[cur] = rows
    sized = {}
    for name, rows in out.items():
        sized[name] = [(op, b - a) for (op, a), (_, b) in zip(rows, rows[1:])] + ([(rows[-1][0], 4)] if rows else [])
    return sized


LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
INSTR = re.compile(r"^\s+([a-z][a-z0-9_]*)\b")


def is_instr(line):
    s = line.split(";", 1)[0] if not line.lstrip().startswith(";") else ""
    m = INSTR.match(s)
    return m.group(1) if m and not s.lstrip().startswith(".") else None


MODE = os.environ.get("ASM_ALIGN_MODE", "promote")        # promote | padonly (never re-encode) | promoteonly (never pad)  -- experiments
SHIFT = os.environ.get("ASM_ALIGN_PHASE", "0") == "4"      # experiment: everything at 4 mod 8 instead (one s_nop at the function's entry)


def align(lines, sized, only):
    out, stats = [], {"functions": 0, "promoted": 0, "padded": 0, "pairs": 0}
    i, n = 0, len(lines)
    while i < n:
        line = lines[i]
        m = LABEL.match(line)
        name = m.group(1) if m else None
        if name is None or name not in sized or (only and not any(k in name for k in only)):
            out.append(line)
            i += 1
            continue
        # a function: up to its .Lfunc_end label
        j = i + 1
        while j < n and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i + 1:j]
        sizes = sized[name]
        items, k = [], 0                       # (kind, text, size): kind = 'i' instruction, 'l' label, 'o' other
        for b in body:
            op = is_instr(b)
            if op:
                dop = sizes[k][0] if k < len(sizes) else None
                if dop not in (op, op + "_e32", op + "_e64"):      # (inline asm may leave the encoding suffix out)
                    raise SystemExit(f"{name}: instruction {k}: text has {op!r}, disassembly has {dop!r}")
                items.append(["i", b, sizes[k][1], op if dop == op else (op, dop)])
                k += 1
            elif LABEL.match(b):
                items.append(["l", b, 0, None])
            else:
                items.append(["o", b, 0, None])
        out.append(line)
        if SHIFT:
            out.append("\ts_nop 0")
        stats["functions"] += 1
        phase = 0
        idx = 0
        while idx < len(items):
            kind, text, size, op = items[idx]
            if kind != "i":
                out.append(text)
                idx += 1
                continue
            if size % 8 == 0:
                assert phase == 0 or MODE == "promoteonly", (name, idx)
                out.append(text)
                idx += 1
                continue
            assert size == 4, (name, op, size)
            # what executes next in the text: the next instruction or a label
            nxt = next((it for it in items[idx + 1:] if it[0] in "il"), None)
            if phase == 4:
                out.append(text)
                phase = 0
                idx += 1
                continue
            if nxt is not None and nxt[0] == "i" and nxt[2] == 4:
                out.append(text)
                phase = 4
                stats["pairs"] += 1
                idx += 1
                continue
            written, op = (op, op) if isinstance(op, str) else op
            if MODE != "padonly" and op.startswith("v_") and op.endswith("_e32") and op[:-4] not in NO_E64:
                out.append(re.sub(r"\b" + re.escape(written) + r"\b", op[:-4] + "_e64", text, count=1))
                stats["promoted"] += 1
            elif MODE == "promoteonly":
                out.append(text)
                phase = 4
            else:
                out.append("\ts_nop 0")
                out.append(text)
                stats["padded"] += 1
            idx += 1
        i = j
    return out, stats


def strip_asm_nops(lines, keys):
    out, inside, removed = [], False, 0
    for line in lines:
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".") and any(k in m.group(1) for k in keys):
            inside = True
        elif line.startswith(".Lfunc_end"):
            inside = False
        if inside and line.strip() == "s_nop 0" and out and "#ASMEND" in out[-1]:
            removed += 1
            continue
        out.append(line)
    return out, removed


def main():
    src, dst, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
    strip_keys, align_keys, do_align, mode = [], [], False, None
    for a in rest:
        if a in ("--strip-asm-nops", "--align"):
            mode = a
            do_align = do_align or a == "--align"
        elif mode == "--strip-asm-nops":
            strip_keys.append(a)
        elif mode == "--align":
            align_keys.append(a)
        else:
            raise SystemExit(f"asm_pass.py: unexpected argument {a!r}")
    with open(src) as f:
        lines = f.read().split("\n")
    name = os.path.basename(os.path.dirname(os.path.abspath(src))) + "/" + os.path.basename(src)
    if strip_keys:
        lines, removed = strip_asm_nops(lines, strip_keys)
        print(f"[asm_pass] {name}: {removed} s_nop pads behind inline-asm statements removed ({', '.join(strip_keys)})")
    if do_align:
        with tempfile.TemporaryDirectory() as d:
            cur, obj = os.path.join(d, "in.s"), os.path.join(d, "in.o")
            with open(cur, "w") as f:
                f.write("\n".join(lines))
            assemble(cur, obj)
            sized = function_sizes(obj)
        lines, stats = align(lines, sized, align_keys)
        print(f"[asm_pass] {name}: {stats['functions']} functions aligned: {stats['promoted']} instructions re-encoded as VOP3, "
              f"{stats['padded']} padded with s_nop, {stats['pairs']} pairs of 4-byte instructions left alone")
    with open(dst, "w") as f:
        f.write("\n".join(lines))


if __name__ == "__main__":
    main()
