// curve_bn254_snarks_g2.hip -- instantiates the MSM kernels and engine for Bn254G2 (one TU per curve keeps builds parallel).
// multiply-add chain form of the device field (fpu.h CTT_FPU_CHAIN), measured per curve (profiles/bench_r02_chain_variants.txt):
// saturated-limb field: no effect
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 0
#endif  // CTT_FPU_CHAIN
#include "hip_backend.h"
extern "C" const ctt::CurveOps* ctt_ops_bn254_snarks_g2(void) { return ctt::CurveImpl<ctt::Bn254G2>::ops(); }
