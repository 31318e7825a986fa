// curve_vesta.hip -- instantiates the MSM kernels and engine for VestaEc (one TU per curve keeps builds parallel).
#include "hip_backend.h"
extern "C" const ctt::CurveOps* ctt_ops_vesta(void) { return ctt::CurveImpl<ctt::VestaEc>::ops(); }
