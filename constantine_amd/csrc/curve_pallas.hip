// curve_pallas.hip -- instantiates the MSM kernels and engine for PallasEc (one TU per curve keeps builds parallel).
#include "hip_backend.h"
extern "C" const ctt::CurveOps* ctt_ops_pallas(void) { return ctt::CurveImpl<ctt::PallasEc>::ops(); }
