#!/bin/bash
# experiment helper: device assembly of one curve's translation unit -> a library beside the shipped one
#     bash tools/asm_variant.sh <device.s> <curve> <tag>        -> tools/libctt_msm_hip_<tag>.so  (the other objects are the in-tree build's)
set -e
S=$1; CURVE=$2; TAG=$3
B=/opt/rocm/lib/llvm/bin
W=$(mktemp -d)
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$S" -o $W/dev.o
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
cd "$(dirname "$0")/../constantine_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 -Wno-unused-result \
    --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c curve_$CURVE.hip -o $W/curve.o
OBJS=$(ls build/*.o | grep -v "build/curve_$CURVE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $OBJS $W/curve.o -o ../../tools/libctt_msm_hip_$TAG.so
rm -rf $W
echo "tools/libctt_msm_hip_$TAG.so"
