// curve_bls12_381_g2.hip -- instantiates the MSM kernels and engine for Bls12381G2 (one TU per curve keeps builds parallel).
// multiply-add chain form of the device field (fpu.h CTT_FPU_CHAIN), measured per curve (profiles/bench_r02_chain_variants.txt):
// one wave per SIMD: the s_nop hipcc pads asm statements with is not hidden -- 7.92 ms (0) vs 8.63 (8) vs 8.99 (4) at 2^20
// (and 8.52 ms with 4 at two waves per SIMD, accumulator partly in LDS, against 7.88 ms with 0 at one wave: DESIGN.md section 5)
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 0
#endif  // CTT_FPU_CHAIN
#include "hip_backend.h"
#ifdef CTT_TU_ACCUM_INTO   // (the second build of this file, into_bls12_381_g2.o: the accumulate kernel's INTO form only -- hip_backend.h)
template void ctt::launch_accum_into<ctt::Bls12381G2::FD>(hipStream_t, const ctt::AccumArgs<ctt::Bls12381G2::FD>&, uint32_t);
#else
extern "C" const ctt::CurveOps* ctt_ops_bls12_381_g2(void) { return ctt::CurveImpl<ctt::Bls12381G2>::ops(); }
#endif
