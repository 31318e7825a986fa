// curve_bls12_381_g1.hip -- instantiates the MSM kernels and engine for Bls12381G1 (one TU per curve keeps builds parallel).
// multiply-add chain form of the device field (fpu.h CTT_FPU_CHAIN), measured per curve: round 2 (profiles/bench_r02_chain_variants.txt)
// 2.396 ms (4) vs 2.402 (8) vs 2.417 (0) on the accumulate kernel at 2^20; round 3, with the pipelined gather
// (profiles/bench_r03_chain_variants.txt): a whole column per asm statement (14) -- 2.94 ms per MSM against 2.96 (8) and 3.00 (4)
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 14
#endif  // CTT_FPU_CHAIN
#include "hip_backend.h"
extern "C" const ctt::CurveOps* ctt_ops_bls12_381_g1(void) { return ctt::CurveImpl<ctt::Bls12381G1>::ops(); }
