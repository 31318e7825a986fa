#!/bin/bash
# One translation unit through asm_pass.py (see there) between the compiler and the assembler:
#     with_asm_pass.sh <out.o> <src.hip> "<asm_pass.py arguments>" <hipcc> <compile flags ...>
# hipcc -S (device) -> asm_pass.py -> assembler -> lld -> offload bundle -> hipcc --cuda-host-only with that bundle embedded: the steps
# `hipcc -c` runs by itself (`hipcc -### -c` prints them), with the pass in the middle.  Any failure exits non-zero and the Makefile
# compiles the file the plain way instead.
set -euo pipefail
OUT=$1; SRC=$2; PASS=$3; HIPCC=$4; shift 4
HERE=$(cd "$(dirname "$0")" && pwd)
LLVM=$(dirname "$(readlink -f "$HIPCC")")/../lib/llvm/bin
[ -x "$LLVM/clang" ] || LLVM=/opt/rocm/lib/llvm/bin
ARCH=gfx950
W=${OUT%.o}.asm
rm -rf "$W"; mkdir -p "$W"
"$HIPCC" "$@" --cuda-device-only -S "$SRC" -o "$W/dev.s" 2> "$W/compile.log" || { cat "$W/compile.log" >&2; exit 1; }
grep -v 'argument unused during compilation' "$W/compile.log" >&2 || true
# shellcheck disable=SC2086
python3 "$HERE/asm_pass.py" "$W/dev.s" "$W/dev_pass.s" $PASS
"$LLVM/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu=$ARCH -c "$W/dev_pass.s" -o "$W/dev.o"
"$LLVM/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$W/dev.out" "$W/dev.o"
"$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$W/dev.out" -output="$W/dev.hipfb"
"$HIPCC" "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$W/dev.hipfb" -c "$SRC" -o "$OUT"
rm -rf "$W"
