// curve_bls12_381_g1.hip -- instantiates the MSM kernels and engine for Bls12381G1 (one TU per curve keeps builds parallel).
// multiply-add chain form of the device field (fpu.h CTT_FPU_CHAIN), measured per curve: round 2 (profiles/bench_r02_chain_variants.txt)
// 2.396 ms (4) vs 2.402 (8) vs 2.417 (0) on the accumulate kernel at 2^20; round 3, with the pipelined gather
// (profiles/bench_r03_chain_variants.txt): a whole column per asm statement (14) -- 2.94 ms per MSM against 2.96 (8) and 3.00 (4)
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 14
#endif  // CTT_FPU_CHAIN
#include "hip_backend.h"
#ifdef CTT_TU_ACCUM_INTO   // (the second build of this file, into_bls12_381_g1.o: the accumulate kernel's INTO form only -- hip_backend.h)
template void ctt::launch_accum_into<ctt::Bls12381G1::FD>(hipStream_t, const ctt::AccumArgs<ctt::Bls12381G1::FD>&, uint32_t);
#else
extern "C" const ctt::CurveOps* ctt_ops_bls12_381_g1(void) { return ctt::CurveImpl<ctt::Bls12381G1>::ops(); }
#endif
