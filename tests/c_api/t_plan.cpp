// Plan invariants of the MSM engine over the whole supported size range (host-only; no GPU, no kernels run).
// Build: g++ -std=c++17 -O1 -I constantine_amd/csrc tests/c_api/t_plan.cpp -o t_plan
#include "msm_pipeline.h"
using namespace ctt;

int main() {
  MsmOptions o;
  o.lanes = 131072;
  int bad = 0;
  const unsigned long long sizes[] = {1, 2, 5, 63, 64, 65, 1000, 4096, 16383, 16384, 16385, 65536, 1ull << 20, (1ull << 20) + 7,
                                      1ull << 22, 1ull << 24, 1ull << 26, (1ull << 26) + 1, 1ull << 27, 1ull << 28, 1ull << 29,
                                      1ull << 30, (1ull << 31) - 1};
  for (int bits : {254, 255}) {
    for (unsigned long long n : sizes) {
      for (int c = 0; c <= 20; c++) {
        if (c == 1) continue;
        o.c = c;
        const MsmPlan p = make_plan((uint32_t)n, bits, o);
        // balanced windows over bits + 1 bits: r windows of cb + 1 bits, the others cb; c is the widest
        bool ok = p.c >= 2 && p.c <= (c == 0 ? 18 : 20) && (c == 0 || p.c <= c) && p.B == (1u << (p.c - 1));
        ok = ok && p.lay.cb * p.W + p.lay.r == bits + 1 && p.lay.r >= 0 && p.lay.r < p.W && p.c == p.lay.cmax();
        ok = ok && p.lay.off((uint32_t)p.W - 1) + p.lay.width((uint32_t)p.W - 1) == bits + 1 && p.lay.off(0) == 0;
        ok = ok && p.NG >= 1 && p.NG <= 4096 && p.NG <= p.B && (p.B >> p.gshift) == p.NG && p.B / p.NG <= 1024;
        ok = ok && p.jbits + 1 + p.gshift <= 32 && (1ull << p.jbits) >= n && p.gshift_narrow <= p.gshift;
        ok = ok && (unsigned long long)p.S * p.slice >= n && (unsigned long long)(p.S - 1) * p.slice < n;
        ok = ok && p.K >= 4 && (unsigned long long)p.G * p.K >= n && (unsigned long long)(p.G - 1) * p.K < n;
        // every reachable bucket of a narrower window (2^(cb-1) of them) has a group
        ok = ok && (p.lay.r == 0 || ((1ull << (p.lay.cb - 1)) >> p.gshift_narrow) <= p.NG);
        if (!ok) {
          bad++;
          printf("BAD bits=%d n=%llu c_req=%d -> c=%d W=%d B=%u NG=%u gshift=%u/%u jbits=%u slice=%u S=%u K=%u G=%u\n", bits, n, c,
                 p.c, p.W, p.B, p.NG, p.gshift, p.gshift_narrow, p.jbits, p.slice, p.S, p.K, p.G);
        }
      }
    }
  }
  printf("%s\n", bad ? "FAILED" : "plans ok");
  return bad ? 1 : 0;
}
