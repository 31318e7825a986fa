// protocols.hip -- the MSM's immediate callers behind the reference's own C symbols:
//
//   EIP-4844 KZG    ctt_eth_kzg_blob_to_kzg_commitment / _compute_kzg_proof / _compute_blob_kzg_proof and the context
//                   constructor / destructor   (include/constantine/protocols/ethereum_eip4844_kzg.h:106,126,153,200,238;
//                   constantine/ethereum_eip4844_kzg.nim:297-444, commitments/kzg.nim:186-223)
//   EIP-2537        ctt_eth_evm_bls12381_g1msm / _g2msm   (include/constantine/protocols/ethereum_evm_precompiles.h:386,419;
//                   constantine/ethereum_evm_precompiles.nim:316-389,894-1060)
//
// Host side in C++: wire formats, range / curve checks, the Fiat-Shamir hash, point (de)compression.  GPU side: every MSM
// (cached SRS: ctt_hip_msm_with_bases; precompile inputs: ctt_hip_msm_host), the subgroup checks (ctt_hip_subgroup_check) and
// the quotient polynomial of an opening (ctt_hip_fr_quotient; the branch "z is a root of unity" runs the reference's other
// formula here on the host).  Verification needs pairings and is out of scope, like the PeerDAS cell functions.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "host_fp64.h"
#include "hip_errors.h"
#include "msm_pipeline.h"   // (OutOfDeviceMemory)

// (part 3 of the header declares this file's symbols with the reference's packed enums and byte-array structs; the definitions
// below spell the same ABI with uint8_t and plain pointers, so the typed declarations stay out of this translation unit)
#define CTT_MSM_HIP_NO_PROTOCOLS 1
#include "../../include/ctt_msm_hip.h"

using namespace ctt;

namespace {

using FpH = Fp64<BLS12_381_Fp>;
using FrH = Fp64<BLS12_381_Fr>;
using Fp2H = Fp2<FpH>;
constexpr int N_BLOB = 4096;

// a failed HIP call inside a protocol symbol: recorded as the thread's last error and thrown to the symbol's boundary, which
// returns CTT_HIP_STATUS_GPU_UNAVAILABLE -- rounds 1-4 aborted here (hip_errors.h)
#define PROT_HIP_CHECK(x)                                                                                          \
  do {                                                                                                             \
    hipError_t e_ = (x);                                                                                           \
    if (e_ != hipSuccess) ::ctt::hip_failed(#x, hipGetErrorString(e_), e_ == hipErrorOutOfMemory, __FILE__, __LINE__); \
  } while (0)

// What a protocol symbol returns when the GPU cannot serve the call: a value outside every status enum of the reference
// (ctt_eth_kzg_status 0..9, ctt_eth_trusted_setup_status 0..2, ctt_evm_status 0..6 -- their *_status_to_string print
// "InvalidStatusCode" for it); ctt_hip_last_error() / _message() of the calling thread say why.  Outputs are untouched.
constexpr uint8_t GPU_UNAVAILABLE = CTT_HIP_STATUS_GPU_UNAVAILABLE;

// the body of a protocol symbol with the GPU's failures turned into that status
template <class Fn>
uint8_t gpu_guarded(Fn&& fn) {
  ErrorGuard guard;
  try {
    return (uint8_t)fn();
  } catch (const HipFailure&) {
    return GPU_UNAVAILABLE;
  } catch (const OutOfDeviceMemory& e) {
    set_last_error(ERR_OUT_OF_MEMORY, "out of device memory (%zu bytes requested)", e.bytes);
    return GPU_UNAVAILABLE;
  }
}

// ---- field helpers (host, 64-bit limbs) ---------------------------------------------------------------------------------
template <class F>
F raw_const(const uint32_t* w) {
  F r;
  for (int i = 0; i < F::N; i++) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  return r;
}
// canonical little-endian limbs < modulus?
template <class F>
bool below_modulus(const uint64_t* v) {
  for (int i = F::N - 1; i >= 0; i--) {
    if (v[i] < F::P(i)) return true;
    if (v[i] > F::P(i)) return false;
  }
  return false;  // equal
}
// big-endian bytes (8*N of them) -> canonical limbs
template <class F>
void limbs_from_be(const uint8_t* be, uint64_t* v) {
  for (int i = 0; i < F::N; i++) {
    uint64_t w = 0;
    for (int b = 0; b < 8; b++) w = (w << 8) | be[8 * (F::N - 1 - i) + b];
    v[i] = w;
  }
}
template <class F>
void limbs_to_be(const uint64_t* v, uint8_t* be) {
  for (int i = 0; i < F::N; i++)
    for (int b = 0; b < 8; b++) be[8 * (F::N - 1 - i) + b] = (uint8_t)(v[i] >> (56 - 8 * b));
}
template <class F>
F to_mont(const uint64_t* canon) {
  F a;
  for (int i = 0; i < F::N; i++) a.l[i] = canon[i];
  return F::mul(a, raw_const<F>(F::Params::R2));
}
template <class F>
void from_mont(const F& a, uint64_t* canon) {
  F one_raw = F::zero();
  one_raw.l[0] = 1;
  const F r = F::mul(a, one_raw);
  for (int i = 0; i < F::N; i++) canon[i] = r.l[i];
}
template <class F>
F fpow(const F& base, const uint64_t* e, int nlimbs) {
  F r = F::one();
  bool started = false;
  for (int i = nlimbs * 64 - 1; i >= 0; i--) {
    if (started) r = F::sqr(r);
    if ((e[i >> 6] >> (i & 63)) & 1) {
      r = started ? F::mul(r, base) : base;
      started = true;
    }
  }
  return r;
}
// v > (p-1)/2 for a canonical value
bool fp_is_larger_half(const uint64_t* v) {
  uint64_t h[FpH::N];   // (p-1)/2
  uint64_t carry = 0;
  for (int i = FpH::N - 1; i >= 0; i--) {
    const uint64_t w = FpH::P(i) - (i == 0 ? 1 : 0);   // p is odd: p - 1 only changes limb 0
    h[i] = (w >> 1) | (carry << 63);
    carry = w & 1;
  }
  for (int i = FpH::N - 1; i >= 0; i--) {
    if (v[i] > h[i]) return true;
    if (v[i] < h[i]) return false;
  }
  return false;
}

// ---- SHA-256 (FIPS 180-4), for the Fiat-Shamir challenge of compute_blob_kzg_proof -------------------------------------
struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len = 0;
  size_t fill = 0;
  Sha256() {
    static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h, iv, sizeof h);
  }
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
      const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
      const uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
#if defined(__x86_64__)
  // the same compression function on the SHA extensions (the Fiat-Shamir challenge hashes a whole blob, 128 KiB: 0.35 ms with
  // the portable rounds, ~0.07 ms here); chosen at run time
  __attribute__((target("sha,sse4.1,ssse3"))) void blocks_shani(const uint8_t* p, size_t nblk) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i t = _mm_loadu_si128((const __m128i*)&h[0]);    // a b c d
    __m128i s1 = _mm_loadu_si128((const __m128i*)&h[4]);   // e f g h
    t = _mm_shuffle_epi32(t, 0xB1);                        // c d a b
    s1 = _mm_shuffle_epi32(s1, 0x1B);                      // h g f e
    __m128i s0 = _mm_alignr_epi8(t, s1, 8);                // a b e f
    s1 = _mm_blend_epi16(s1, t, 0xF0);                     // c d g h
    for (size_t b = 0; b < nblk; b++, p += 64) {
      const __m128i save0 = s0, save1 = s1;
      __m128i m[4];
      for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), bswap);
      for (int r = 0; r < 16; r++) {
        __m128i w = m[r & 3];
        if (r >= 4) {   // message schedule: w[4r .. 4r+3] from the previous four vectors
          __m128i x = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
          x = _mm_add_epi32(x, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
          w = _mm_sha256msg2_epu32(x, m[(r + 3) & 3]);
          m[r & 3] = w;
        }
        __m128i wk = _mm_add_epi32(w, _mm_loadu_si128((const __m128i*)&K[4 * r]));
        s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
        wk = _mm_shuffle_epi32(wk, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, wk);
      }
      s0 = _mm_add_epi32(s0, save0);
      s1 = _mm_add_epi32(s1, save1);
    }
    t = _mm_shuffle_epi32(s0, 0x1B);                       // f e b a
    s1 = _mm_shuffle_epi32(s1, 0xB1);                      // d c h g
    s0 = _mm_blend_epi16(t, s1, 0xF0);                     // d c b a
    s1 = _mm_alignr_epi8(s1, t, 8);                        // h g f e
    _mm_storeu_si128((__m128i*)&h[0], s0);
    _mm_storeu_si128((__m128i*)&h[4], s1);
  }
  static bool have_shani() {
    static const bool v = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3") &&
                          !(getenv("CTT_HIP_NO_SHANI") && atoi(getenv("CTT_HIP_NO_SHANI")) != 0);
    return v;
  }
#else
  void blocks_shani(const uint8_t*, size_t) {}
  static bool have_shani() { return false; }
#endif
  void update(const uint8_t* p, size_t n) {
    len += n;
    while (n) {
      if (fill == 0 && n >= 64) {
        if (have_shani()) {
          const size_t nb = n / 64;
          blocks_shani(p, nb);
          p += 64 * nb;
          n -= 64 * nb;
          continue;
        }
        block(p);
        p += 64;
        n -= 64;
        continue;
      }
      const size_t take = n < 64 - fill ? n : 64 - fill;
      memcpy(buf + fill, p, take);
      fill += take;
      p += take;
      n -= take;
      if (fill == 64) {
        if (have_shani()) blocks_shani(buf, 1); else block(buf);
        fill = 0;
      }
    }
  }
  void finish(uint8_t out[32]) {
    const uint64_t bits = len * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; i++) {
      out[4 * i] = (uint8_t)(h[i] >> 24);
      out[4 * i + 1] = (uint8_t)(h[i] >> 16);
      out[4 * i + 2] = (uint8_t)(h[i] >> 8);
      out[4 * i + 3] = (uint8_t)h[i];
    }
  }
};

// ---- status codes of the two reference interfaces (values as in their headers) -----------------------------------------
enum : int { KZG_Success = 0, KZG_VerificationFailure = 1, KZG_InputsLengthsMismatch = 2, KZG_ScalarZero = 3, KZG_ScalarLargerThanCurveOrder = 4,
             KZG_EccInvalidEncoding = 5, KZG_EccCoordinateGreaterThanOrEqualModulus = 6, KZG_EccPointNotOnCurve = 7, KZG_EccPointNotInSubgroup = 8 };
enum : int { TS_Success = 0, TS_MissingOrInaccessibleFile = 1, TS_InvalidFile = 2 };
enum : int { EVM_Success = 0, EVM_InvalidInputSize = 1, EVM_InvalidOutputSize = 2, EVM_IntLargerThanModulus = 3, EVM_PointNotOnCurve = 4,
             EVM_PointNotInSubgroup = 5 };

// ---- BLS12-381 G1, ZCash / IETF compressed encoding (serialization/codecs_bls12_381.nim) ---------------------------------
// -> affine Montgomery {x, y} in the C-API layout (96 bytes; the neutral is (0,0)).  No subgroup check here.
int g1_decompress(uint8_t aff[96], const uint8_t in[48]) {
  if (!(in[0] & 0x80)) return KZG_EccInvalidEncoding;   // only the compressed form
  if (in[0] & 0x40) {                                   // infinity: every other bit zero
    if (in[0] & 0x3f) return KZG_EccInvalidEncoding;
    for (int i = 1; i < 48; i++)
      if (in[i]) return KZG_EccInvalidEncoding;
    memset(aff, 0, 96);
    return KZG_Success;
  }
  uint8_t xb[48];
  memcpy(xb, in, 48);
  xb[0] &= 0x1f;
  uint64_t xc[FpH::N];
  limbs_from_be<FpH>(xb, xc);
  if (!below_modulus<FpH>(xc)) return KZG_EccCoordinateGreaterThanOrEqualModulus;
  const FpH x = to_mont<FpH>(xc);
  FpH four = FpH::one();
  four = FpH::dbl(FpH::dbl(four));
  const FpH y2 = FpH::add(FpH::mul(FpH::sqr(x), x), four);
  // p = 3 (mod 4): y = y2^((p+1)/4)
  uint64_t e[FpH::N];
  {
    uint64_t c = 1;   // p + 1
    for (int i = 0; i < FpH::N; i++) {
      const uint64_t s = FpH::P(i) + c;
      c = (s < c) ? 1 : 0;
      e[i] = s;
    }
    for (int i = 0; i < FpH::N; i++) e[i] = (e[i] >> 2) | (i + 1 < FpH::N ? e[i + 1] << 62 : 0);
  }
  FpH y = fpow<FpH>(y2, e, FpH::N);
  if (!FpH::eq(FpH::sqr(y), y2)) return KZG_EccPointNotOnCurve;
  uint64_t yc[FpH::N];
  from_mont<FpH>(y, yc);
  if (((in[0] & 0x20) != 0) != fp_is_larger_half(yc)) y = FpH::neg(y);
  memcpy(aff, x.l, 48);
  memcpy(aff + 48, y.l, 48);
  return KZG_Success;
}
void g1_compress(uint8_t out[48], const uint8_t aff[96]) {
  FpH x, y;
  memcpy(x.l, aff, 48);
  memcpy(y.l, aff + 48, 48);
  if (x.is_zero() && y.is_zero()) {
    memset(out, 0, 48);
    out[0] = 0xc0;
    return;
  }
  uint64_t xc[FpH::N], yc[FpH::N];
  from_mont<FpH>(x, xc);
  from_mont<FpH>(y, yc);
  limbs_to_be<FpH>(xc, out);
  out[0] |= 0x80 | (fp_is_larger_half(yc) ? 0x20 : 0);
}

// blob -> 4096 canonical little-endian scalars (blob_to_bigint_polynomial: every element < r)
int blob_to_scalars(uint8_t* out_le, const uint8_t* blob) {
  for (int i = 0; i < N_BLOB; i++) {
    uint64_t v[FrH::N];
    limbs_from_be<FrH>(blob + 32 * i, v);
    if (!below_modulus<FrH>(v)) return KZG_ScalarLargerThanCurveOrder;
    memcpy(out_le + 32 * i, v, 32);
  }
  return KZG_Success;
}
// 256-bit big-endian integer reduced mod r (fromDigest, ethereum_eip4844_kzg.nim:103-126; the EVM scalars, :941-952) -> canonical limbs
void reduce_256_mod_r(const uint8_t be[32], uint64_t v[4]) {
  limbs_from_be<FrH>(be, v);
  while (!below_modulus<FrH>(v)) {   // 2^256 < 3r: at most two subtractions
    uint64_t bw = 0;
    for (int i = 0; i < 4; i++) {
      const unsigned __int128 x = (unsigned __int128)v[i] - FrH::P(i) - bw;
      v[i] = (uint64_t)x;
      bw = (uint64_t)(x >> 64) & 1;
    }
  }
}
// fiatShamirChallenge (ethereum_eip4844_kzg.nim:126-148): sha256(domain | 16-byte big-endian degree | blob | commitment) mod r
void fiat_shamir_challenge(uint64_t z[4], const uint8_t* blob, const uint8_t commitment[48]) {
  Sha256 t;
  t.update((const uint8_t*)"FSBLOBVERIFY_V1_", 16);
  uint8_t deg[16] = {0};
  deg[14] = (uint8_t)(N_BLOB >> 8);
  deg[15] = (uint8_t)(N_BLOB & 0xff);
  t.update(deg, 16);
  t.update(blob, (size_t)N_BLOB * 32);
  t.update(commitment, 48);
  uint8_t d[32];
  t.finish(d);
  reduce_256_mod_r(d, z);
}

uint32_t bit_reverse(uint32_t i, int bits) {
  uint32_t r = 0;
  for (int b = 0; b < bits; b++) r |= ((i >> b) & 1u) << (bits - 1 - b);
  return r;
}

// the 4096 roots of unity in bit-reversed order, Montgomery residues (ctx.domain_brp, commitments_setups/ethereum_kzg_srs.nim)
std::vector<FrH> domain_brp() {
  // w = 7^((r-1)/4096)
  uint64_t e[4];
  for (int i = 0; i < 4; i++) e[i] = FrH::P(i);
  e[0] -= 1;
  for (int i = 0; i < 4; i++) e[i] = (e[i] >> 12) | (i + 1 < 4 ? e[i + 1] << 52 : 0);
  const uint64_t seven[4] = {7, 0, 0, 0};
  const FrH w = fpow<FrH>(to_mont<FrH>(seven), e, 4);
  std::vector<FrH> nat(N_BLOB), brp(N_BLOB);
  FrH x = FrH::one();
  for (int i = 0; i < N_BLOB; i++) {
    nat[i] = x;
    x = FrH::mul(x, w);
  }
  for (int i = 0; i < N_BLOB; i++) brp[i] = nat[bit_reverse((uint32_t)i, 12)];
  return brp;
}

// Montgomery's trick: out[i] = 1/v[i]; every v[i] non-zero
void batch_inverse(const std::vector<FrH>& v, std::vector<FrH>& out) {
  const size_t n = v.size();
  out.resize(n);
  FrH run = FrH::one();
  for (size_t i = 0; i < n; i++) {
    out[i] = run;
    run = FrH::mul(run, v[i]);
  }
  FrH inv = FrH::inv(run);
  for (size_t i = n; i-- > 0;) {
    out[i] = FrH::mul(inv, out[i]);
    inv = FrH::mul(inv, v[i]);
  }
}

// getQuotientPoly (math/polynomials/polynomials.nim; kzg_prove, commitments/kzg.nim:204-223) on the host: y = p(z) and
// q = (p - y)/(X - z) in evaluation form over the bit-reversed domain, both branches (z outside / inside the domain).  The
// device runs the first branch (ctt_hip_fr_quotient); this is what serves "z is a root of unity" and what the device form is
// tested against.  poly, q: canonical little-endian scalars; z canonical limbs; y canonical limbs out.
void quotient_host(const std::vector<FrH>& dom, const uint8_t* poly_le, const uint64_t z_c[4], uint8_t* q_le, uint64_t y_c[4]) {
  const int n = N_BLOB;
  std::vector<FrH> p(n);
  for (int i = 0; i < n; i++) {
    uint64_t v[4];
    memcpy(v, poly_le + 32 * i, 32);
    p[i] = to_mont<FrH>(v);
  }
  const FrH z = to_mont<FrH>(z_c);
  int m = -1;
  for (int i = 0; i < n; i++)
    if (FrH::eq(dom[i], z)) {
      m = i;
      break;
    }
  std::vector<FrH> q(n), d, inv;
  FrH y;
  if (m < 0) {
    d.resize(n);
    for (int i = 0; i < n; i++) d[i] = FrH::sub(z, dom[i]);                       // z - w_i
    batch_inverse(d, inv);
    FrH s = FrH::zero();
    for (int i = 0; i < n; i++) s = FrH::add(s, FrH::mul(FrH::mul(p[i], dom[i]), inv[i]));
    const uint64_t nn[4] = {(uint64_t)n, 0, 0, 0};
    const FrH zn = fpow<FrH>(z, nn, 1);                                          // z^n
    const FrH ninv = FrH::inv(to_mont<FrH>(nn));
    y = FrH::mul(FrH::mul(FrH::sub(zn, FrH::one()), ninv), s);                    // barycentric evaluation
    for (int i = 0; i < n; i++) q[i] = FrH::mul(FrH::sub(y, p[i]), inv[i]);       // (p_i - y)/(w_i - z)
  } else {
    y = p[m];
    d.reserve(n - 1);
    for (int i = 0; i < n; i++)
      if (i != m) d.push_back(FrH::sub(dom[i], z));                               // w_i - z
    batch_inverse(d, inv);
    const FrH zinv = FrH::inv(z);
    FrH acc = FrH::zero();
    int k = 0;
    for (int i = 0; i < n; i++) {
      if (i == m) continue;
      q[i] = FrH::mul(FrH::sub(p[i], y), inv[k++]);
      acc = FrH::add(acc, FrH::mul(FrH::mul(q[i], dom[i]), zinv));                // q_m = - sum q_i w_i / z
    }
    q[m] = FrH::neg(acc);
  }
  from_mont<FrH>(y, y_c);
  for (int i = 0; i < n; i++) {
    uint64_t v[4];
    from_mont<FrH>(q[i], v);
    memcpy(q_le + 32 * i, v, 32);
  }
}

}  // namespace

// ---- the KZG context: the Lagrange SRS in bit-reversal order cached on the GPU, the domain next to it ---------------------
struct ctt_eth_kzg_context_struct {
  std::mutex mu;
  ctt_hip_msm_ctx* hip = nullptr;
  bool own_hip = false;
  int device = 0;
  ctt_hip_msm_bases* bases = nullptr;
  std::vector<FrH> domain;      // bit-reversed roots of unity, Montgomery
  void* d_domain = nullptr;     // the same on the device (ctt_hip_fr_quotient's layout)
  void* d_poly = nullptr;       // 4096 canonical scalars of the blob being opened
  void* d_q = nullptr;          // quotient evaluations (never leave the GPU before the MSM)
};

namespace {

// srs: 4096 x 48 compressed G1 Lagrange points in ceremony (file) order
int kzg_context_build(ctt_eth_kzg_context_struct** out, const uint8_t* srs, int device, int table) {
  std::vector<uint8_t> aff((size_t)N_BLOB * 96);
  std::vector<int> status(N_BLOB, 0);
  // 4096 square roots (380 squarings each): a few host threads
  const unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nth; t++)
    th.emplace_back([&, t]() {
      for (int i = (int)t; i < N_BLOB; i += (int)nth)   // the commitment uses the points in bit-reversal order
        status[i] = g1_decompress(aff.data() + (size_t)bit_reverse((uint32_t)i, 12) * 96, srs + (size_t)i * 48);
    });
  for (std::thread& x : th) x.join();
  for (int i = 0; i < N_BLOB; i++)
    if (status[i] != KZG_Success) return TS_InvalidFile;
  ctt_eth_kzg_context_struct* c = new ctt_eth_kzg_context_struct();
  c->device = device;
  auto fail = [&](int st) {
    if (c->bases) ctt_hip_msm_bases_destroy(c->hip, c->bases);
    if (c->d_domain) (void)hipFree(c->d_domain);
    if (c->d_poly) (void)hipFree(c->d_poly);
    if (c->d_q) (void)hipFree(c->d_q);
    if (c->hip) ctt_hip_msm_ctx_destroy(c->hip);
    delete c;
    return st;
  };
  c->hip = ctt_hip_msm_ctx_create(device);
  if (!c->hip) return fail(GPU_UNAVAILABLE);            // no usable device: ctt_hip_last_error() == -3
  c->own_hip = true;
  std::vector<uint8_t> ok(N_BLOB);
  if (ctt_hip_subgroup_check(c->hip, CTT_HIP_BLS12_381_G1, ok.data(), aff.data(), N_BLOB, 0) != 0) return fail(GPU_UNAVAILABLE);
  for (int i = 0; i < N_BLOB; i++)
    if (!ok[i]) return fail(TS_InvalidFile);
  c->bases = table ? ctt_hip_msm_bases_create_table(c->hip, CTT_HIP_BLS12_381_G1, aff.data(), N_BLOB, 0, 0)
                   : ctt_hip_msm_bases_create(c->hip, CTT_HIP_BLS12_381_G1, aff.data(), N_BLOB, 0);
  if (!c->bases) return fail(GPU_UNAVAILABLE);
  c->domain = domain_brp();
  try {
    DeviceScope device_scope(device);
    PROT_HIP_CHECK(hipMalloc(&c->d_domain, (size_t)N_BLOB * 32));
    PROT_HIP_CHECK(hipMalloc(&c->d_poly, (size_t)N_BLOB * 32));
    PROT_HIP_CHECK(hipMalloc(&c->d_q, (size_t)N_BLOB * 32));
    PROT_HIP_CHECK(hipMemcpy(c->d_domain, c->domain.data(), (size_t)N_BLOB * 32, hipMemcpyHostToDevice));
  } catch (const HipFailure&) {
    return fail(GPU_UNAVAILABLE);
  }
  *out = c;
  return TS_Success;
}

// [proof]_1 and y = p(z) for a validated polynomial (canonical scalars) and challenge
// -> KZG_Success, or GPU_UNAVAILABLE when the GPU refused (nothing written)
int kzg_prove(ctt_eth_kzg_context_struct* c, uint8_t proof[48], uint64_t y_c[4], const uint8_t* poly_le, const uint64_t z_c[4]) {
  std::lock_guard<std::mutex> lock(c->mu);
  DeviceScope device_scope(c->device);
  hipStream_t s = (hipStream_t)ctt_hip_msm_stream(c->hip);
  if (!s) return GPU_UNAVAILABLE;                       // the context was lost to an earlier HIP failure
  PROT_HIP_CHECK(hipMemcpyAsync(c->d_poly, poly_le, (size_t)N_BLOB * 32, hipMemcpyHostToDevice, s));
  uint8_t r_aff[96];
  uint64_t y_tmp[4];
  const int rc = ctt_hip_fr_quotient(c->hip, CTT_HIP_BLS12_381_G1, c->d_q, y_tmp, c->d_poly, c->d_domain, z_c, N_BLOB);
  int mrc;
  if (rc == 0) {
    mrc = ctt_hip_msm_with_bases(c->hip, c->bases, CTT_HIP_COEF_BIG, CTT_HIP_OUT_AFF, r_aff, c->d_q, N_BLOB, 1);
  } else if (rc == -2) {   // z is one of the roots of unity: the reference's other formula, on the host
    std::vector<uint8_t> q((size_t)N_BLOB * 32);
    quotient_host(c->domain, poly_le, z_c, q.data(), y_tmp);
    mrc = ctt_hip_msm_with_bases(c->hip, c->bases, CTT_HIP_COEF_BIG, CTT_HIP_OUT_AFF, r_aff, q.data(), N_BLOB, 0);
  } else {
    return GPU_UNAVAILABLE;
  }
  if (mrc != 0) return GPU_UNAVAILABLE;
  memcpy(y_c, y_tmp, 32);
  g1_compress(proof, r_aff);
  return KZG_Success;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

// ---- host-only helpers (no GPU): what tests/ check against the spec restatement, and what a binding may reuse -------------
void ctt_hip_sha256(uint8_t out[32], const uint8_t* data, size_t len) {
  Sha256 t;
  t.update(data, len);
  t.finish(out);
}
// -> ctt_eth_kzg_status; aff = affine Montgomery {x, y} (C-API layout), (0,0) for the neutral; no subgroup check
int ctt_hip_bls12_381_g1_decompress(uint8_t aff[96], const uint8_t in[48]) { return g1_decompress(aff, in); }
void ctt_hip_bls12_381_g1_compress(uint8_t out[48], const uint8_t aff[96]) { g1_compress(out, aff); }
// blob (131072 bytes) -> 4096 canonical little-endian scalars; -> ctt_eth_kzg_status
int ctt_hip_eth_kzg_blob_to_scalars(uint8_t* scalars_le, const uint8_t* blob) { return blob_to_scalars(scalars_le, blob); }
// Fiat-Shamir challenge of compute_blob_kzg_proof, 32 big-endian bytes
void ctt_hip_eth_kzg_challenge(uint8_t z_be[32], const uint8_t* blob, const uint8_t commitment[48]) {
  uint64_t z[4];
  fiat_shamir_challenge(z, blob, commitment);
  limbs_to_be<FrH>(z, z_be);
}
// quotient polynomial on the host, both branches: poly / q = 4096 canonical little-endian scalars, z / y = 32 little-endian bytes
void ctt_hip_eth_kzg_quotient_host(uint8_t* q_le, uint8_t y_le[32], const uint8_t* poly_le, const uint8_t z_le[32]) {
  static const std::vector<FrH> dom = domain_brp();
  uint64_t z[4], y[4];
  memcpy(z, z_le, 32);
  quotient_host(dom, poly_le, z, q_le, y);
  memcpy(y_le, y, 32);
}

// ---- context (ethereum_eip4844_kzg.h:200-238) -------------------------------------------------------------------------
// From memory: the 4096 x 48 bytes of the Lagrange-form G1 SRS in ceremony order (what the file's first 4096 lines hold).
// device: the GPU the SRS is cached on; table != 0 caches it as a window table (ctt_hip_msm_bases_create_table).
// -> ctt_eth_trusted_setup_status
int ctt_hip_eth_kzg_context_from_srs(ctt_eth_kzg_context_struct** ctx, const uint8_t* g1_lagrange_compressed, size_t n_points,
                                     int device, int table) {
  if (!ctx || !g1_lagrange_compressed || n_points != (size_t)N_BLOB) return TS_InvalidFile;
  return gpu_guarded([&]() { return kzg_context_build(ctx, g1_lagrange_compressed, device, table); });
}
// The c-kzg text format of the Ethereum ceremony ("4096\n65\n", one hex point per line; the reference ships it as
// constantine/commitments_setups/trusted_setup_ethereum_kzg4844_reference.dat): format must be cttEthTSFormat_ckzg4844 (0).
// The reader follows load_ckzg4844 (constantine/commitments_setups/ethereum_kzg_srs.nim:242-350): "%4u" counts, then 4096 Lagrange
// G1 lines of exactly 96 hex characters, 65 monomial G2 lines of exactly 192, 4096 monomial G1 lines of 96, LF or CRLF line ends;
// anything else -- a short file, a long line, a character that is not a hex digit -- is cttEthTS_InvalidFile.  The Lagrange points
// are decompressed and subgroup-checked (they are the commitment key).  The G2 and monomial-G1 sections serve verification and
// the PeerDAS proofs, which are out of scope here: their lines are checked for length, hex digits, the compression flag and
// coordinates below p, not decompressed (round 4 did not read them at all: a truncated file loaded -- ADVICE r4).
static bool srs_hex_line(FILE* f, size_t hex_chars, uint8_t* out) {
  char buf[200];
  size_t got = 0;
  int ch;
  while ((ch = fgetc(f)) != EOF && ch != '\n' && ch != '\r') {
    if (got >= hex_chars || !isxdigit(ch)) return false;
    buf[got++] = (char)ch;
  }
  if (got != hex_chars) return false;
  while (ch == '\r' || ch == '\n') {          // the line end (and blank lines, as fscanf's "%*[\r\n]" skips them)
    ch = fgetc(f);
  }
  if (ch != EOF) ungetc(ch, f);
  auto nib = [](char c) { return (uint8_t)(c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10); };
  for (size_t i = 0; i < hex_chars / 2; i++) out[i] = (uint8_t)(nib(buf[2 * i]) << 4 | nib(buf[2 * i + 1]));
  return true;
}
// a compressed point's x coordinate(s): flag bit set, every 48-byte big-endian coordinate below p (the neutral: flags only)
static bool srs_compressed_shape(const uint8_t* in, int coords) {
  if (!(in[0] & 0x80)) return false;
  if (in[0] & 0x40) {
    if (in[0] & 0x3f) return false;
    for (int i = 1; i < 48 * coords; i++)
      if (in[i]) return false;
    return true;
  }
  for (int k = 0; k < coords; k++) {
    uint8_t be[48];
    memcpy(be, in + 48 * k, 48);
    if (k == 0) be[0] &= 0x1f;
    uint64_t v[FpH::N];
    limbs_from_be<FpH>(be, v);
    if (!below_modulus<FpH>(v)) return false;
  }
  return true;
}
uint8_t ctt_eth_kzg_context_new(ctt_eth_kzg_context_struct** ctx, const char* filepath, uint8_t format) {
  if (!ctx || !filepath || format != 0) return TS_InvalidFile;
  FILE* f = fopen(filepath, "rb");
  if (!f) return TS_MissingOrInaccessibleFile;
  std::vector<uint8_t> srs((size_t)N_BLOB * 48);
  unsigned n1 = 0, n2 = 0;
  int st = TS_Success;
  if (fscanf(f, "%4u\n", &n1) != 1 || n1 != (unsigned)N_BLOB || fscanf(f, "%4u\n", &n2) != 1 || n2 != 65u) st = TS_InvalidFile;
  for (int i = 0; st == TS_Success && i < N_BLOB; i++)
    if (!srs_hex_line(f, 96, &srs[(size_t)i * 48])) st = TS_InvalidFile;
  for (int i = 0; st == TS_Success && i < 65; i++) {
    uint8_t g2[96];
    if (!srs_hex_line(f, 192, g2) || !srs_compressed_shape(g2, 2)) st = TS_InvalidFile;
  }
  for (int i = 0; st == TS_Success && i < N_BLOB; i++) {
    uint8_t g1[48];
    if (!srs_hex_line(f, 96, g1) || !srs_compressed_shape(g1, 1)) st = TS_InvalidFile;
  }
  fclose(f);
  if (st != TS_Success) return (uint8_t)st;
  const char* dv = getenv("CTT_HIP_DEVICE");
  // the SRS as a window table: 10 MB for 4096 points, commitments 0.38 instead of 0.52 ms (profiles/kzg_timing_r04.txt)
  return gpu_guarded([&]() { return kzg_context_build(ctx, srs.data(), dv ? atoi(dv) : 0, 1); });
}
// ethereum_eip4844_kzg.h:232: the reference's constructor with PrecomputedMSM lookup tables (t base groups, b bits per window -- CPU
// tables for the FK20 proofs of PeerDAS).  The same context as above: the SRS is cached on the GPU as a window table whatever t and b
// say; the PeerDAS functions those tables serve are not part of this library.
uint8_t ctt_eth_kzg_context_new_with_precompute(ctt_eth_kzg_context_struct** ctx, const char* filepath, uint8_t format, int /*t*/, int /*b*/) {
  return ctt_eth_kzg_context_new(ctx, filepath, format);
}
void ctt_eth_kzg_context_delete(ctt_eth_kzg_context_struct* c) {
  if (!c) return;
  (void)gpu_guarded([&]() {
    std::lock_guard<std::mutex> lock(c->mu);
    DeviceScope device_scope(c->device);
    ctt_hip_msm_sync(c->hip);
    if (c->bases) ctt_hip_msm_bases_destroy(c->hip, c->bases);
    if (c->d_domain) (void)hipFree(c->d_domain);
    if (c->d_poly) (void)hipFree(c->d_poly);
    if (c->d_q) (void)hipFree(c->d_q);
    if (c->own_hip) ctt_hip_msm_ctx_destroy(c->hip);
    return 0;
  });
  delete c;
}

// ---- EIP-4844 (ethereum_eip4844_kzg.h:106,126,153) -----------------------------------------------------------------------
uint8_t ctt_eth_kzg_blob_to_kzg_commitment(const ctt_eth_kzg_context_struct* ctx, uint8_t dst[48], const uint8_t* blob) {
  ctt_eth_kzg_context_struct* c = const_cast<ctt_eth_kzg_context_struct*>(ctx);
  std::vector<uint8_t> poly((size_t)N_BLOB * 32);
  const int st = blob_to_scalars(poly.data(), blob);
  if (st != KZG_Success) return (uint8_t)st;
  uint8_t r_aff[96];
  {
    std::lock_guard<std::mutex> lock(c->mu);
    const int rc = ctt_hip_msm_with_bases(c->hip, c->bases, CTT_HIP_COEF_BIG, CTT_HIP_OUT_AFF, r_aff, poly.data(), N_BLOB, 0);
    if (rc != 0) return GPU_UNAVAILABLE;     // the GPU refused (ctt_hip_last_error says why); dst untouched
  }
  g1_compress(dst, r_aff);
  return KZG_Success;
}

uint8_t ctt_eth_kzg_compute_kzg_proof(const ctt_eth_kzg_context_struct* ctx, uint8_t proof[48], uint8_t y_be[32], const uint8_t* blob,
                                      const uint8_t z_be[32]) {
  ctt_eth_kzg_context_struct* c = const_cast<ctt_eth_kzg_context_struct*>(ctx);
  uint64_t z[4], y[4];
  limbs_from_be<FrH>(z_be, z);
  if (!below_modulus<FrH>(z)) return KZG_ScalarLargerThanCurveOrder;   // bytes_to_bls_field (:158-166) comes first
  std::vector<uint8_t> poly((size_t)N_BLOB * 32);
  const int st = blob_to_scalars(poly.data(), blob);
  if (st != KZG_Success) return (uint8_t)st;
  const uint8_t pst = gpu_guarded([&]() { return kzg_prove(c, proof, y, poly.data(), z); });
  if (pst != KZG_Success) return pst;
  limbs_to_be<FrH>(y, y_be);
  return KZG_Success;
}

uint8_t ctt_eth_kzg_compute_blob_kzg_proof(const ctt_eth_kzg_context_struct* ctx, uint8_t proof[48], const uint8_t* blob,
                                           const uint8_t commitment[48]) {
  ctt_eth_kzg_context_struct* c = const_cast<ctt_eth_kzg_context_struct*>(ctx);
  // bytes_to_kzg_commitment: a validated point (encoding, range, on the curve, in the subgroup; the neutral is allowed)
  uint8_t aff[96];
  int st = g1_decompress(aff, commitment);
  if (st != KZG_Success) return (uint8_t)st;
  {
    std::lock_guard<std::mutex> lock(c->mu);
    uint8_t ok = 0;
    if (ctt_hip_subgroup_check(c->hip, CTT_HIP_BLS12_381_G1, &ok, aff, 1, 0) != 0) return GPU_UNAVAILABLE;
    if (!ok) return KZG_EccPointNotInSubgroup;
  }
  std::vector<uint8_t> poly((size_t)N_BLOB * 32);
  st = blob_to_scalars(poly.data(), blob);
  if (st != KZG_Success) return (uint8_t)st;
  uint64_t z[4], y[4];
  fiat_shamir_challenge(z, blob, commitment);
  return gpu_guarded([&]() { return kzg_prove(c, proof, y, poly.data(), z); });
}

// the _parallel forms (ethereum_eip4844_kzg_parallel.h:40,61,73): the thread pool is not used, as in the MSM's _parallel symbols
uint8_t ctt_eth_kzg_blob_to_kzg_commitment_parallel(const void* /*tp*/, const ctt_eth_kzg_context_struct* ctx, uint8_t dst[48],
                                                    const uint8_t* blob) {
  return ctt_eth_kzg_blob_to_kzg_commitment(ctx, dst, blob);
}
uint8_t ctt_eth_kzg_compute_kzg_proof_parallel(const void* /*tp*/, const ctt_eth_kzg_context_struct* ctx, uint8_t proof[48],
                                               uint8_t y_be[32], const uint8_t* blob, const uint8_t z_be[32]) {
  return ctt_eth_kzg_compute_kzg_proof(ctx, proof, y_be, blob, z_be);
}
uint8_t ctt_eth_kzg_compute_blob_kzg_proof_parallel(const void* /*tp*/, const ctt_eth_kzg_context_struct* ctx, uint8_t proof[48],
                                                    const uint8_t* blob, const uint8_t commitment[48]) {
  return ctt_eth_kzg_compute_blob_kzg_proof(ctx, proof, blob, commitment);
}

// ---- EIP-2537 BLS12_G1MSM / BLS12_G2MSM (ethereum_evm_precompiles.h:386,419) -------------------------------------------
// parseRawUint: a 64-byte big-endian field element, top 16 bytes zero, below p
static bool evm_fp(const uint8_t* b64, FpH& out) {
  for (int i = 0; i < 16; i++)
    if (b64[i]) return false;
  uint64_t v[FpH::N];
  limbs_from_be<FpH>(b64 + 16, v);
  if (!below_modulus<FpH>(v)) return false;
  out = to_mont<FpH>(v);
  return true;
}
static void evm_fp_out(const FpH& a, uint8_t* b64) {
  uint64_t v[FpH::N];
  from_mont<FpH>(a, v);
  memset(b64, 0, 16);
  limbs_to_be<FpH>(v, b64 + 16);
}
// The reference handles the pairs in order, each point fully (coordinates, curve, subgroup) before the next one
// (fromRawCoords, ethereum_evm_precompiles.nim:316-389): the status is that of the FIRST offending pair.  parse[i] is the
// host-side verdict on pair i; the subgroup checks of all points (or of those in front of the first offender) are one call of
// ctt_hip_subgroup_check.
// When every pair parsed, the MSM is issued at once and the subgroup checks run beside it on host threads (up to 256 points; above
// that the check is a launch, and it goes first): the checks of a call of 16-256 pairs take as long as its MSM (profiles/evm_timing_r04.txt),
// and a call that fails them only discards a result.  r_aff is written in every case, the caller looks at it on EVM_Success only.
static int evm_validate_and_msm(int curve, void* r_aff, const void* coefs, const uint8_t* pts, const std::vector<int>& parse) {
  const size_t n = parse.size();
  size_t bad = n;
  for (size_t i = 0; i < n; i++)
    if (parse[i] != EVM_Success) {
      bad = i;
      break;
    }
  std::vector<uint8_t> ok(bad ? bad : 1);
  if (bad < n) {   // the call fails anyway: which status -- a point in front of the first malformed pair outside the subgroup?
    if (bad > 0 && ctt_hip_subgroup_check(nullptr, curve, ok.data(), pts, bad, 0) != 0) return GPU_UNAVAILABLE;
    for (size_t i = 0; i < bad; i++)
      if (!ok[i]) return EVM_PointNotInSubgroup;
    return parse[bad];
  }
  int check_rc = 0;
  bool beside = n <= 256;   // (the host side of ctt_hip_subgroup_check; above it the check is a launch on the context the MSM uses:
                                  //  measured slower side by side than one after the other, 512 G2 pairs 9.0 against 4.8 ms)
  std::thread check;
  if (beside) {
    try {
      check = std::thread([&]() { check_rc = ctt_hip_subgroup_check(nullptr, curve, ok.data(), pts, n, 0); });
    } catch (const std::system_error&) {
      beside = false;   // (no thread to be had: one after the other)
    }
  }
  if (!beside) {
    check_rc = ctt_hip_subgroup_check(nullptr, curve, ok.data(), pts, n, 0);
    if (check_rc != 0) return GPU_UNAVAILABLE;
    for (size_t i = 0; i < n; i++)
      if (!ok[i]) return EVM_PointNotInSubgroup;
  }
  // The MSM.  A refusal because all in-flight slots of the context are held by another thread's device-resident tickets (ERR_BUSY)
  // passes: wait for it (up to ~2 s) instead of failing a consensus call; anything else (bad arguments, no device, out of device
  // memory, a HIP failure) comes back as GPU_UNAVAILABLE at once with the thread's last error set -- rounds 1-4 aborted the process here.
  int rc = ctt_hip_msm_host(curve, CTT_HIP_COEF_BIG, CTT_HIP_OUT_AFF, r_aff, coefs, pts, n);
  for (int spin = 0; rc == -1 && ctt_hip_last_error() == ERR_BUSY && spin < 20000; spin++) {
    std::this_thread::sleep_for(std::chrono::microseconds(100));
    rc = ctt_hip_msm_host(curve, CTT_HIP_COEF_BIG, CTT_HIP_OUT_AFF, r_aff, coefs, pts, n);
  }
  if (beside) check.join();
  if (rc != 0 || check_rc != 0) return GPU_UNAVAILABLE;
  for (size_t i = 0; i < n; i++)
    if (!ok[i]) return EVM_PointNotInSubgroup;
  return EVM_Success;
}

uint8_t ctt_eth_evm_bls12381_g1msm(uint8_t* r, size_t r_len, const uint8_t* inputs, size_t inputs_len) {
  if (inputs_len == 0 || inputs_len % 160 != 0) return EVM_InvalidInputSize;
  if (r_len != 128) return EVM_InvalidOutputSize;
  const size_t n = inputs_len / 160;
  std::vector<uint8_t> pts(n * 96, 0), coefs(n * 32);
  std::vector<int> parse(n, EVM_Success);
  FpH four = FpH::dbl(FpH::dbl(FpH::one()));
  for (size_t i = 0; i < n; i++) {
    const uint8_t* rec = inputs + i * 160;
    FpH x, y;
    if (!evm_fp(rec, x) || !evm_fp(rec + 64, y)) {
      parse[i] = EVM_IntLargerThanModulus;
    } else if (!(x.is_zero() && y.is_zero())) {
      if (!FpH::eq(FpH::sqr(y), FpH::add(FpH::mul(FpH::sqr(x), x), four))) parse[i] = EVM_PointNotOnCurve;
      else {
        memcpy(&pts[i * 96], x.l, 48);
        memcpy(&pts[i * 96 + 48], y.l, 48);
      }
    }
    uint64_t s[4];
    reduce_256_mod_r(rec + 128, s);   // the spec allows any s < 2^256; the group is cyclic of order r
    memcpy(&coefs[i * 32], s, 32);
  }
  uint8_t aff[96];
  const int st = evm_validate_and_msm(CTT_HIP_BLS12_381_G1, aff, coefs.data(), pts.data(), parse);
  if (st != EVM_Success) return (uint8_t)st;
  FpH x, y;
  memcpy(x.l, aff, 48);
  memcpy(y.l, aff + 48, 48);
  evm_fp_out(x, r);         // the neutral is (0,0) in both encodings
  evm_fp_out(y, r + 64);
  return EVM_Success;
}

uint8_t ctt_eth_evm_bls12381_g2msm(uint8_t* r, size_t r_len, const uint8_t* inputs, size_t inputs_len) {
  if (inputs_len == 0 || inputs_len % 288 != 0) return EVM_InvalidInputSize;
  if (r_len != 256) return EVM_InvalidOutputSize;
  const size_t n = inputs_len / 288;
  std::vector<uint8_t> pts(n * 192, 0), coefs(n * 32);
  std::vector<int> parse(n, EVM_Success);
  const FpH four = FpH::dbl(FpH::dbl(FpH::one()));
  const Fp2H b2{four, four};   // b' = 4 (1 + i)
  for (size_t i = 0; i < n; i++) {
    const uint8_t* rec = inputs + i * 288;
    Fp2H x, y;
    if (!evm_fp(rec, x.c0) || !evm_fp(rec + 64, x.c1) || !evm_fp(rec + 128, y.c0) || !evm_fp(rec + 192, y.c1)) {
      parse[i] = EVM_IntLargerThanModulus;
    } else if (!(x.is_zero() && y.is_zero())) {
      if (!Fp2H::eq(Fp2H::sqr(y), Fp2H::add(Fp2H::mul(Fp2H::sqr(x), x), b2))) parse[i] = EVM_PointNotOnCurve;
      else {
        memcpy(&pts[i * 192], x.c0.l, 48);
        memcpy(&pts[i * 192 + 48], x.c1.l, 48);
        memcpy(&pts[i * 192 + 96], y.c0.l, 48);
        memcpy(&pts[i * 192 + 144], y.c1.l, 48);
      }
    }
    uint64_t s[4];
    reduce_256_mod_r(rec + 256, s);
    memcpy(&coefs[i * 32], s, 32);
  }
  uint8_t aff[192];
  const int st = evm_validate_and_msm(CTT_HIP_BLS12_381_G2, aff, coefs.data(), pts.data(), parse);
  if (st != EVM_Success) return (uint8_t)st;
  for (int k = 0; k < 4; k++) {
    FpH v;
    memcpy(v.l, aff + 48 * k, 48);
    evm_fp_out(v, r + 64 * k);
  }
  return EVM_Success;
}

#pragma GCC visibility pop
}  // extern "C"
