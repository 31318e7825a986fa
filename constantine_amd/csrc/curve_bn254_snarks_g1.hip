// curve_bn254_snarks_g1.hip -- instantiates the MSM kernels and engine for Bn254G1 (one TU per curve keeps builds parallel).
#include "hip_backend.h"
extern "C" const ctt::CurveOps* ctt_ops_bn254_snarks_g1(void) { return ctt::CurveImpl<ctt::Bn254G1>::ops(); }
