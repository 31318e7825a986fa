// curve_bls12_381_g1.hip -- instantiates the MSM kernels and engine for Bls12381G1 (one TU per curve keeps builds parallel).
#include "hip_backend.h"
extern "C" const ctt::CurveOps* ctt_ops_bls12_381_g1(void) { return ctt::CurveImpl<ctt::Bls12381G1>::ops(); }
