// curve_bn254_snarks_g1.hip -- instantiates the MSM kernels and engine for Bn254G1 (one TU per curve keeps builds parallel).
// multiply-add chain form of the device field (fpu.h CTT_FPU_CHAIN), measured per curve (profiles/bench_r02_chain_variants.txt):
// 4.83 ms (4) vs 4.95 (8) vs 5.13 (0) at 2^22
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 4
#endif  // CTT_FPU_CHAIN
// waves per SIMD the accumulate kernel is compiled for: 4 (128 registers + 60 B of scratch) measured 1.5 - 5 % faster than
// 3 (141 registers) for the 9-limb fields, same box (profiles/bench_r02_waves4.txt)
#ifndef CTT_ACCUM_WAVES
#define CTT_ACCUM_WAVES 4
#endif
#include "hip_backend.h"
#ifdef CTT_TU_ACCUM_INTO   // (the second build of this file, into_bn254_snarks_g1.o: the accumulate kernel's INTO form only -- hip_backend.h)
template void ctt::launch_accum_into<ctt::Bn254G1::FD>(hipStream_t, const ctt::AccumArgs<ctt::Bn254G1::FD>&, uint32_t);
#else
extern "C" const ctt::CurveOps* ctt_ops_bn254_snarks_g1(void) { return ctt::CurveImpl<ctt::Bn254G1>::ops(); }
#endif
