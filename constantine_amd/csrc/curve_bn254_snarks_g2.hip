// curve_bn254_snarks_g2.hip -- instantiates the MSM kernels and engine for Bn254G2 (one TU per curve keeps builds parallel).
// multiply-add chain form of the device field (fpu.h CTT_FPU_CHAIN), measured per curve (profiles/bench_r02_chain_variants.txt):
// saturated-limb field: no effect
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 0
#endif  // CTT_FPU_CHAIN
#include "hip_backend.h"
#ifdef CTT_TU_ACCUM_INTO   // (the second build of this file, into_bn254_snarks_g2.o: the accumulate kernel's INTO form only -- hip_backend.h)
template void ctt::launch_accum_into<ctt::Bn254G2::FD>(hipStream_t, const ctt::AccumArgs<ctt::Bn254G2::FD>&, uint32_t);
#else
extern "C" const ctt::CurveOps* ctt_ops_bn254_snarks_g2(void) { return ctt::CurveImpl<ctt::Bn254G2>::ops(); }
#endif
