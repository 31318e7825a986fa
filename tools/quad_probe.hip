// tools/quad_probe.hip -- the register-resident quad doubling / addition of the window-sum kernel against the one-lane
// formulas, limb for limb (debug probe).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I constantine_amd/csrc tools/quad_probe.hip -o tools/quad_probe.bin
#include "hip_backend.h"
#include <stdio.h>
using namespace ctt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class C>
__global__ void k_probe(XYZZ<typename C::FD>* out) {
  using F = typename C::F;
  using FD = typename C::FD;
  if (threadIdx.x >= 4) return;
  const int role = threadIdx.x;
  Affine<F> g = generator<C>();
  Affine<FD> G{FD::from_sat(g.x), FD::from_sat(g.y)};
  XYZZ<FD> P = XYZZ<FD>::from_affine(G);
  XYZZ<FD> D1 = xyzz_dbl<FD>(P);            // 2G, one lane
  XYZZ<FD> D2 = P;
  xyzz_dbl_quad_reg<FD>(D2, role);          // 2G, quad
  XYZZ<FD> D3 = xyzz_dbl<FD>(D1);           // 4G
  XYZZ<FD> D4 = D1;
  xyzz_dbl_quad_reg<FD>(D4, role);
  XYZZ<FD> A1 = xyzz_add_inl<FD>(D1, P);    // 3G
  XYZZ<FD> A2 = D1;
  xyzz_add_quad_reg<FD>(A2, P, role);
  XYZZ<FD> A3 = xyzz_add_inl<FD>(D3, A1);   // 7G
  XYZZ<FD> A4 = D3;
  xyzz_add_quad_reg<FD>(A4, A1, role);
  {
    // two broadcasts of one per-lane value feeding one subtraction (the pattern of Y3 = A - Bv)
    FD T = FD::mul(G.x, role == 0 ? G.x : role == 1 ? G.y : role == 2 ? D1.x : D1.y);
    FD A = quad_bcast<0>(T), B = quad_bcast<1>(T);
    XYZZ<FD> dbg;
    dbg.x = A; dbg.y = B; dbg.zz = fsub<FD, 2>(A, B); dbg.zzz = T;
    out[32 + role] = dbg;
  }
  XYZZ<FD>* o = out + 8 * role;
  o[0] = D1; o[1] = D2; o[2] = D3; o[3] = D4; o[4] = A1; o[5] = A2; o[6] = A3; o[7] = A4;
}

template <class C>
static void run(const char* name) {
  using FD = typename C::FD;
  XYZZ<FD>* d;
  CK(hipMalloc(&d, 36 * sizeof(XYZZ<FD>)));
  hipLaunchKernelGGL(k_probe<C>, dim3(1), dim3(64), 0, 0, d);
  CK(hipDeviceSynchronize());
  XYZZ<FD> h[36];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  const char* lab[4] = {"dbl(G)", "dbl(2G)", "2G+G", "4G+3G"};
  for (int t = 0; t < 4; t++)
    for (int role = 0; role < 4; role++) {
      const uint32_t* a = (const uint32_t*)&h[8 * role + 2 * t];
      const uint32_t* b = (const uint32_t*)&h[8 * role + 2 * t + 1];
      const int nl = sizeof(XYZZ<FD>) / 4 / 4;
      for (int f = 0; f < 4; f++) {
        int bad = -1;
        for (int i = 0; i < nl; i++) if (a[f * nl + i] != b[f * nl + i]) { bad = i; break; }
        if (bad >= 0) printf("%s %s lane %d field %d differs at limb %d: one-lane %08x quad %08x\n", name, lab[t], role, f, bad, a[f * nl + bad], b[f * nl + bad]);
      }
    }
  for (int k = 0; k < 8; k++) {   // raw limbs of lane 0's eight results, for a group-element comparison on the host
    const int nl = sizeof(XYZZ<FD>) / 4 / 4;
    printf("RAW %s %d %d", name, k, nl);
    for (int i = 0; i < 4 * nl; i++) printf(" %x", ((const uint32_t*)&h[k])[i]);
    printf("\n");
  }
  for (int t = 0; t < 4; t++) {
    const int nl = sizeof(XYZZ<FD>) / 4 / 4;
    printf("%s %s quad y limb0 per lane:", name, lab[t]);
    for (int role = 0; role < 4; role++) printf(" %08x", ((const uint32_t*)&h[8 * role + 2 * t + 1])[nl]);
    printf("   one-lane y limb0 %08x\n", ((const uint32_t*)&h[2 * t])[nl]);
  }
  for (int role = 0; role < 4; role++) {
    const uint32_t* q = (const uint32_t*)&h[32 + role];
    const int nl = sizeof(XYZZ<FD>) / 4 / 4;
    printf("%s lane %d: A %08x %08x  B %08x %08x  A-B+2p %08x %08x  own T %08x %08x\n", name, role, q[0], q[1], q[nl], q[nl + 1], q[2 * nl], q[2 * nl + 1], q[3 * nl], q[3 * nl + 1]);
  }
  printf("%s probe done\n", name);
}
int main() {
  run<Bls12381G1>("bls12_381_g1");
  run<Bn254G1>("bn254_snarks_g1");
  run<PallasEc>("pallas");
  return 0;
}
