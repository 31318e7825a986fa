// curve_pallas.hip -- instantiates the MSM kernels and engine for PallasEc (one TU per curve keeps builds parallel).
// multiply-add chain form of the device field (fpu.h CTT_FPU_CHAIN), measured per curve (profiles/bench_r02_chain_variants.txt):
// 1.058 ms (8) vs 1.068 (4) vs 1.128 (0) at 2^20
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 8
#endif  // CTT_FPU_CHAIN
// waves per SIMD the accumulate kernel is compiled for: 4 (128 registers + 60 B of scratch) measured 1.5 - 5 % faster than
// 3 (141 registers) for the 9-limb fields, same box (profiles/bench_r02_waves4.txt)
#ifndef CTT_ACCUM_WAVES
#define CTT_ACCUM_WAVES 4
#endif
#include "hip_backend.h"
#ifdef CTT_TU_ACCUM_INTO   // (the second build of this file, into_pallas.o: the accumulate kernel's INTO form only -- hip_backend.h)
template void ctt::launch_accum_into<ctt::PallasEc::FD>(hipStream_t, const ctt::AccumArgs<ctt::PallasEc::FD>&, uint32_t);
#else
extern "C" const ctt::CurveOps* ctt_ops_pallas(void) { return ctt::CurveImpl<ctt::PallasEc>::ops(); }
#endif
