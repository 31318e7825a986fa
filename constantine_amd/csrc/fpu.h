// fpu.h -- carry-free ("unsaturated") prime field for the hot kernels: NL limbs of LB < 32 bits.
//
// Why: on gfx950 the carry instruction of a saturated 32-bit-limb multiplier (v_addc_co_u32) costs almost
// as much issue time as the multiplier itself (v_mad_u64_u32), see profiles/microbench_r01.jsonl.  With
// LB-bit limbs a whole product-scanning column fits a 64-bit accumulator, so the Montgomery multiplication
// (same algorithm as the reference's, limbs_montgomery.nim:268-310, radix 2^LB instead of 2^64) is nothing but
// v_mad_u64_u32, plus one mask/shift per column: 392 multiplies and no carries for BLS12-381 (LB = 28,
// NL = 14) against 288 multiplies + 288 carries.
//
// Representation: value v = sum l[i] * 2^(LB*i) with v == x * R' (mod p), R' = 2^(LB*NL).  Elements are NOT
// kept in [0,p): every operation returns limbs normalised to < 2^LB (top limb takes the excess) and a value
// bounded by a small multiple of p that the caller tracks statically:
//     mul, sqr   : result < 2p   provided  a*b < R'*p   (R'/p >= 2^RP_OVER_P_LOG2: 2^11 for BLS12-381, 2^7 for
//                                                        the 254/255-bit fields; operands up to ~11p are fine)
//     add        : bound(a) + bound(b)
//     sub<B>     : a - b + B*p, needs b < B*p;  result < bound(a) + B
// Conversions to/from the reference's representation (Montgomery R = 2^(64L), saturated limbs) happen once per
// point on the way in (from_sat) and once per returned point on the host (HostField, host_fp64.h).
#pragma once
#include <type_traits>

#include "fp.h"

namespace ctt {

template <class UP>
struct FpU {
  static constexpr int HEADROOM_LOG2 = UP::RP_OVER_P_LOG2;  // log2(R'/p): a product of operands < k1*p, k2*p needs k1*k2 < R'/p
  using Params = UP;
  using Sat = Fp<typename UP::Sat>;
  static constexpr int NL = UP::NL;
  static constexpr int LB = UP::LB;
  static constexpr uint32_t MASK = UP::MASK;
  static constexpr int N = NL;        // limb count (for generic helpers)
  static constexpr int MULB = 2;      // mul/sqr outputs are < MULB * p
  static constexpr bool UNSAT = true;
  uint32_t l[NL];

  CTT_HD static FpU zero() {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
    return r;
  }
  CTT_HD static FpU one() {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = UP::ONE[i];
    return r;
  }
  // raw test: every limb zero.  Exact for values we set to zero ourselves (neutral flags, (0,0) inputs);
  // a value that is merely == 0 (mod p) needs is_zero_modp.
  CTT_HD bool is_zero() const {
    uint32_t a = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) a |= l[i];
    return a == 0;
  }
  CTT_HD static FpU select(bool c, const FpU& a, const FpU& b) {
    FpU r;
#if !defined(CTT_FPU_SELECT_CNDMASK)
    // one v_bfi_b32 per limb; v_cndmask_b32 (lane mask read from an SGPR pair) measured 0.3-0.6 % slower on every curve
    // (profiles/bench_r02_select_bfi.txt) and 22 cycles per instruction in isolation (profiles/microbench_isa_r02.jsonl)
    const uint32_t m = 0u - (uint32_t)c;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = (a.l[i] & m) | (b.l[i] & ~m);
#else
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
#endif
    return r;
  }

  // limbs of k*p in canonical radix-2^LB form
  struct KP { uint32_t l[NL]; };
  static constexpr KP kp(int k) {
    KP r{};
    uint64_t c = 0;
    for (int i = 0; i < NL; i++) {
      uint64_t v = (uint64_t)UP::P[i] * (uint64_t)k + c;
      r.l[i] = (i == NL - 1) ? (uint32_t)v : (uint32_t)(v & MASK);
      c = v >> LB;
    }
    return r;
  }

  // value == 0 (mod p), for an element known to be < B*p (B < 2^LB); the limbs may be lazy (sub_lazy).
  // v = k*p  =>  v * (-1/p) == -k (mod 2^LB), and M0INV is -1/p mod 2^LB: one multiplication of the low limb gives the
  // only k the value could be a multiple by; k >= B rejects (all but B in 2^LB values), else compare against k*p.
  template <int B>
  CTT_HD bool is_zero_modp() const {
    const uint32_t k = (0u - l[0] * UP::M0INV) & MASK;
    if (k >= (uint32_t)B) return false;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");  // keep the full comparison behind a real branch (hipcc otherwise speculates it)
#endif
    return equals_kp(*this, k);
  }
  // rare path: a == k*p exactly (a is normalised here first)
  static CTT_HD bool equals_kp(FpU a, uint32_t k) {
    a.normalise();
    uint32_t d = 0;
    uint64_t c = 0;
    for (int i = 0; i < NL; i++) {
      uint64_t v = (uint64_t)UP::P[i] * (uint64_t)k + c;
      uint32_t li = (i == NL - 1) ? (uint32_t)v : (uint32_t)(v & MASK);
      c = v >> LB;
      d |= a.l[i] ^ li;
    }
    return d == 0;
  }

  // carry propagation: limbs < 2^LB afterwards (the top limb keeps the excess)
  CTT_HD void normalise() {
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
      uint32_t c = l[i] >> LB;
      l[i] &= MASK;
      l[i + 1] += c;
    }
  }

  CTT_HD static FpU add(const FpU& a, const FpU& b) {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    r.normalise();
    return r;
  }
  CTT_HD static FpU dbl(const FpU& a) {
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] << 1;
    r.normalise();
    return r;
  }

  // a - b + B*p  (b normalised and < B*p).  B*p is spread over the limbs as
  // (c0 + 2^LB, c1 + 2^LB - 1, ..., c_top - 1) so that no limb goes negative before the carries are propagated.
  template <int B>
  CTT_HD static FpU sub(const FpU& a, const FpU& b) {
    constexpr KP c = kp(B);
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t bias = (i == 0) ? c.l[0] + (1u << LB) : (i == NL - 1) ? c.l[i] - 1u : c.l[i] + (1u << LB) - 1u;
      r.l[i] = a.l[i] + bias - b.l[i];
    }
    r.normalise();
    return r;
  }
  template <int B>
  CTT_HD static FpU neg(const FpU& a) { return sub<B>(zero(), a); }
  // conditional negation; the negated branch is B*p - a (a < B*p).  A raw zero stays a raw zero.
  template <int B>
  CTT_HD static FpU cneg(const FpU& a, bool c) {
    FpU n = sub<B>(zero(), a);
    return select(c, n, a);
  }

  // ---- lazy forms: no carry propagation --------------------------------------------------------------------------
  // sub_lazy<B>(a, b) = a - b + (B+1)*p with every limb left as a_i + bias_i - b_i (< 3*2^LB): half the instructions of
  // sub<B>.  Valid ONLY as an operand of a product (never stored, never an operand of add/sub); a and b normalised,
  // b < B*p.  The bias is (B+1)*p, one multiple more than sub<B> takes, so that the top limb cannot go negative
  // before the carries that normalise() would have delivered.  Value < bound(a) + B + 1.
  // Which products may take lazy operands is a matter of the 64-bit column sums (limbs l_a, l_b; NL terms each):
  //   LAZY_ONE   (lazy x normal) + (lazy x normal) + reduction:   (3 + 3 + 1) * NL * 2^(2 LB) < 2^64
  //   LAZY_BOTH  (lazy x lazy)   + (lazy x normal) + reduction:   (9 + 3 + 1) * NL * 2^(2 LB) < 2^64
  static constexpr bool LAZY_ONE = 7ull * NL <= (1ull << (64 - 2 * LB)) - 1;
  static constexpr bool LAZY_BOTH = 13ull * NL <= (1ull << (64 - 2 * LB)) - 1;
  static_assert(UP::P[NL - 1] >= 8u, "top limb of p too small for the lazy bias arithmetic");
  template <int B>
  CTT_HD static FpU sub_lazy(const FpU& a, const FpU& b) {
    constexpr KP c = kp(B + 1);
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t bias = (i == 0) ? c.l[0] + (1u << LB) : (i == NL - 1) ? c.l[i] - 1u : c.l[i] + (1u << LB) - 1u;
      r.l[i] = a.l[i] + bias - b.l[i];
    }
    return r;
  }
  // conditional negation, lazy in the negated branch: (B+1)*p - a  (a < B*p, normalised)
  template <int B>
  CTT_HD static FpU cneg_lazy(const FpU& a, bool c) {
    return select(c, sub_lazy<B>(zero(), a), a);
  }
  // The same two conditional negations with the condition as a lane MASK WORD m (0 or 0xffffffff, e.g. the sign bit of a sorted
  // entry spread by an arithmetic shift) instead of a bool: bias - a = (a ^ ~0) + (bias + 1) in two's complement, so
  //     r_i = (a_i ^ m) + ((bias_i + 1) & m)
  // is a_i or bias_i - a_i with no compare and no v_cndmask_b32 -- which hipcc makes of every bool select, bfi idiom or not, and
  // which costs 22 cycles per wave on gfx950 against 2.6 for the xor / and / add (profiles/microbench_isa_r02.jsonl; round 4:
  // 14 of them sat in every iteration of the accumulate loop, 14 more in its bucket-start block).
  template <int B>
  CTT_HD static FpU cneg_lazy_m(const FpU& a, uint32_t m) {
    constexpr KP c = kp(B + 1);
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t bias = (i == 0) ? c.l[0] + (1u << LB) : (i == NL - 1) ? c.l[i] - 1u : c.l[i] + (1u << LB) - 1u;
      r.l[i] = (a.l[i] ^ m) + ((bias + 1u) & m);
    }
    return r;
  }
  template <int B>
  CTT_HD static FpU cneg_m(const FpU& a, uint32_t m) {
    constexpr KP c = kp(B);
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t bias = (i == 0) ? c.l[0] + (1u << LB) : (i == NL - 1) ? c.l[i] - 1u : c.l[i] + (1u << LB) - 1u;
      r.l[i] = (a.l[i] ^ m) + ((bias + 1u) & m);
    }
    r.normalise();   // (a no-op on the lanes that keep a: its limbs are normalised already)
    return r;
  }
  CTT_HD static FpU norm(const FpU& a) {
    FpU r = a;
    r.normalise();
    return r;
  }
  // a - b - 2c + K*p, normalised, in one pass (the X3 of the addition formulas: RR - PPP - 2Q).
  // Needs b + 2c < (K-1)*p; every limb gets 3*2^LB on loan from the next one.  Value < bound(a) + K.
  template <int K>
  CTT_HD static FpU sub3(const FpU& a, const FpU& b, const FpU& c) {
    constexpr KP kpv = kp(K);
    FpU r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint32_t bias = (i == 0) ? kpv.l[0] + (3u << LB) : (i == NL - 1) ? kpv.l[i] - 3u : kpv.l[i] + (3u << LB) - 3u;
      r.l[i] = a.l[i] + bias - b.l[i] - (c.l[i] << 1);
    }
    r.normalise();
    return r;
  }

  // ---- column accumulation --------------------------------------------------------------------------------
  // A product-scanning column is a chain acc = a_i*b_j + acc of v_mad_u64_u32 whose 64-bit addend is the running
  // sum, the previous column's carry (acc >> LB) included.  Written as plain C++ the compiler re-associates every
  // column (its Reassociate pass ranks the carry last): it starts a fresh chain from 0 and adds the carry at the end
  // with a separate 64-bit addition (v_lshl_add_u64, as expensive as a multiply) -- one extra quarter-rate
  // instruction per column, 243 per mixed addition.  CTT_FPU_CHAIN != 0 spells the chain out as inline asm so that
  // the carry IS the first addend, with up to CTT_FPU_CHAIN (1..14) multiply-adds per asm statement.
  // The host build (tests/emu, window combine) always takes the C++ form.
#ifndef CTT_FPU_CHAIN
#define CTT_FPU_CHAIN 4
#endif
#if defined(__HIP_DEVICE_COMPILE__) && CTT_FPU_CHAIN
#define CTT_FPU_ASM 1
#define CTT_MADU(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\t"
  // 1..14 dependent multiply-adds into acc per statement (hipcc pads every asm statement with one s_nop), generated: the asm text, the
  // parameter list and the operand list of a chain of N grow by one multiply-add per line; madv takes both factors in VGPRs, mads the
  // second factor -- a constant of the field -- in an SGPR (gfx9 VOP3 reads one SGPR per instruction)
#define CTT_MAD_ASM_1 CTT_MADU(1, 2)
#define CTT_MAD_ASM_2 CTT_MAD_ASM_1 CTT_MADU(3, 4)
#define CTT_MAD_ASM_3 CTT_MAD_ASM_2 CTT_MADU(5, 6)
#define CTT_MAD_ASM_4 CTT_MAD_ASM_3 CTT_MADU(7, 8)
#define CTT_MAD_ASM_5 CTT_MAD_ASM_4 CTT_MADU(9, 10)
#define CTT_MAD_ASM_6 CTT_MAD_ASM_5 CTT_MADU(11, 12)
#define CTT_MAD_ASM_7 CTT_MAD_ASM_6 CTT_MADU(13, 14)
#define CTT_MAD_ASM_8 CTT_MAD_ASM_7 CTT_MADU(15, 16)
#define CTT_MAD_ASM_9 CTT_MAD_ASM_8 CTT_MADU(17, 18)
#define CTT_MAD_ASM_10 CTT_MAD_ASM_9 CTT_MADU(19, 20)
#define CTT_MAD_ASM_11 CTT_MAD_ASM_10 CTT_MADU(21, 22)
#define CTT_MAD_ASM_12 CTT_MAD_ASM_11 CTT_MADU(23, 24)
#define CTT_MAD_ASM_13 CTT_MAD_ASM_12 CTT_MADU(25, 26)
#define CTT_MAD_ASM_14 CTT_MAD_ASM_13 CTT_MADU(27, 28)
#define CTT_MAD_PAR_1 uint32_t a0, uint32_t b0
#define CTT_MAD_PAR_2 CTT_MAD_PAR_1, uint32_t a1, uint32_t b1
#define CTT_MAD_PAR_3 CTT_MAD_PAR_2, uint32_t a2, uint32_t b2
#define CTT_MAD_PAR_4 CTT_MAD_PAR_3, uint32_t a3, uint32_t b3
#define CTT_MAD_PAR_5 CTT_MAD_PAR_4, uint32_t a4, uint32_t b4
#define CTT_MAD_PAR_6 CTT_MAD_PAR_5, uint32_t a5, uint32_t b5
#define CTT_MAD_PAR_7 CTT_MAD_PAR_6, uint32_t a6, uint32_t b6
#define CTT_MAD_PAR_8 CTT_MAD_PAR_7, uint32_t a7, uint32_t b7
#define CTT_MAD_PAR_9 CTT_MAD_PAR_8, uint32_t a8, uint32_t b8
#define CTT_MAD_PAR_10 CTT_MAD_PAR_9, uint32_t a9, uint32_t b9
#define CTT_MAD_PAR_11 CTT_MAD_PAR_10, uint32_t a10, uint32_t b10
#define CTT_MAD_PAR_12 CTT_MAD_PAR_11, uint32_t a11, uint32_t b11
#define CTT_MAD_PAR_13 CTT_MAD_PAR_12, uint32_t a12, uint32_t b12
#define CTT_MAD_PAR_14 CTT_MAD_PAR_13, uint32_t a13, uint32_t b13
#define CTT_MAD_OPS_1(BC) "v"(a0), BC(b0)
#define CTT_MAD_OPS_2(BC) CTT_MAD_OPS_1(BC), "v"(a1), BC(b1)
#define CTT_MAD_OPS_3(BC) CTT_MAD_OPS_2(BC), "v"(a2), BC(b2)
#define CTT_MAD_OPS_4(BC) CTT_MAD_OPS_3(BC), "v"(a3), BC(b3)
#define CTT_MAD_OPS_5(BC) CTT_MAD_OPS_4(BC), "v"(a4), BC(b4)
#define CTT_MAD_OPS_6(BC) CTT_MAD_OPS_5(BC), "v"(a5), BC(b5)
#define CTT_MAD_OPS_7(BC) CTT_MAD_OPS_6(BC), "v"(a6), BC(b6)
#define CTT_MAD_OPS_8(BC) CTT_MAD_OPS_7(BC), "v"(a7), BC(b7)
#define CTT_MAD_OPS_9(BC) CTT_MAD_OPS_8(BC), "v"(a8), BC(b8)
#define CTT_MAD_OPS_10(BC) CTT_MAD_OPS_9(BC), "v"(a9), BC(b9)
#define CTT_MAD_OPS_11(BC) CTT_MAD_OPS_10(BC), "v"(a10), BC(b10)
#define CTT_MAD_OPS_12(BC) CTT_MAD_OPS_11(BC), "v"(a11), BC(b11)
#define CTT_MAD_OPS_13(BC) CTT_MAD_OPS_12(BC), "v"(a12), BC(b12)
#define CTT_MAD_OPS_14(BC) CTT_MAD_OPS_13(BC), "v"(a13), BC(b13)
#define CTT_MAD_DEF(N)                                                                                                              \
  CTT_HD static void madv(uint64_t& acc, CTT_MAD_PAR_##N) { asm(CTT_MAD_ASM_##N : "+v"(acc) : CTT_MAD_OPS_##N("v") : "vcc"); }    \
  CTT_HD static void mads(uint64_t& acc, CTT_MAD_PAR_##N) { asm(CTT_MAD_ASM_##N : "+v"(acc) : CTT_MAD_OPS_##N("s") : "vcc"); }
  CTT_MAD_DEF(1) CTT_MAD_DEF(2) CTT_MAD_DEF(3) CTT_MAD_DEF(4) CTT_MAD_DEF(5) CTT_MAD_DEF(6) CTT_MAD_DEF(7)
  CTT_MAD_DEF(8) CTT_MAD_DEF(9) CTT_MAD_DEF(10) CTT_MAD_DEF(11) CTT_MAD_DEF(12) CTT_MAD_DEF(13) CTT_MAD_DEF(14)
#else
#define CTT_FPU_ASM 0
#endif

  // acc += sum_{i = LO}^{HI-1} a[i] * b[K-i]    (all bounds compile-time: the columns are unrolled by templates)
  static constexpr int GROUP = CTT_FPU_CHAIN < 1 ? 1 : CTT_FPU_CHAIN > 14 ? 14 : CTT_FPU_CHAIN;  // multiply-adds per asm statement
  template <int K, int LO, int HI>
  CTT_HD static void col_ab(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
    if constexpr (LO < HI) {
#if CTT_FPU_ASM
      constexpr int R = HI - LO < GROUP ? HI - LO : GROUP;
#define CTT_AB(j) a[LO + j], b[K - LO - j]
      if constexpr (R == 14) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6), CTT_AB(7), CTT_AB(8), CTT_AB(9), CTT_AB(10), CTT_AB(11), CTT_AB(12), CTT_AB(13));
      else if constexpr (R == 13) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6), CTT_AB(7), CTT_AB(8), CTT_AB(9), CTT_AB(10), CTT_AB(11), CTT_AB(12));
      else if constexpr (R == 12) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6), CTT_AB(7), CTT_AB(8), CTT_AB(9), CTT_AB(10), CTT_AB(11));
      else if constexpr (R == 11) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6), CTT_AB(7), CTT_AB(8), CTT_AB(9), CTT_AB(10));
      else if constexpr (R == 10) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6), CTT_AB(7), CTT_AB(8), CTT_AB(9));
      else if constexpr (R == 9) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6), CTT_AB(7), CTT_AB(8));
      else if constexpr (R == 8) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6), CTT_AB(7));
      else if constexpr (R == 7) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5), CTT_AB(6));
      else if constexpr (R == 6) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4), CTT_AB(5));
      else if constexpr (R == 5) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3), CTT_AB(4));
      else if constexpr (R == 4) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2), CTT_AB(3));
      else if constexpr (R == 3) madv(acc, CTT_AB(0), CTT_AB(1), CTT_AB(2));
      else if constexpr (R == 2) madv(acc, CTT_AB(0), CTT_AB(1));
      else madv(acc, CTT_AB(0));
#undef CTT_AB
      col_ab<K, LO + R, HI>(acc, a, b);
#else
      acc += (uint64_t)a[LO] * b[K - LO];
      col_ab<K, LO + 1, HI>(acc, a, b);
#endif
    }
  }
  // number of leading non-zero limbs P[K-LO], P[K-LO-1], ... (at most GROUP, at most HI-LO)
  template <int K, int LO, int HI>
  static constexpr int nz_run() {
    int n = 0;
    for (int i = LO; i < HI && n < GROUP; i++) {
      if (UP::P[K - i] == 0u) break;
      n++;
    }
    return n;
  }
  // acc += sum_{i = LO}^{HI-1} m[i] * P[K-i]   (zero limbs of P are skipped: the Pasta primes have three)
  template <int K, int LO, int HI>
  CTT_HD static void col_mp(uint64_t& acc, const uint32_t* m) {
    if constexpr (LO < HI) {
      if constexpr (UP::P[K - LO] == 0u) {
        col_mp<K, LO + 1, HI>(acc, m);
      } else {
#if CTT_FPU_ASM
        constexpr int R = nz_run<K, LO, HI>();
#define CTT_MP(j) m[LO + j], UP::P[K - LO - j]
        if constexpr (R == 14) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6), CTT_MP(7), CTT_MP(8), CTT_MP(9), CTT_MP(10), CTT_MP(11), CTT_MP(12), CTT_MP(13));
        else if constexpr (R == 13) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6), CTT_MP(7), CTT_MP(8), CTT_MP(9), CTT_MP(10), CTT_MP(11), CTT_MP(12));
        else if constexpr (R == 12) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6), CTT_MP(7), CTT_MP(8), CTT_MP(9), CTT_MP(10), CTT_MP(11));
        else if constexpr (R == 11) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6), CTT_MP(7), CTT_MP(8), CTT_MP(9), CTT_MP(10));
        else if constexpr (R == 10) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6), CTT_MP(7), CTT_MP(8), CTT_MP(9));
        else if constexpr (R == 9) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6), CTT_MP(7), CTT_MP(8));
        else if constexpr (R == 8) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6), CTT_MP(7));
        else if constexpr (R == 7) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5), CTT_MP(6));
        else if constexpr (R == 6) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4), CTT_MP(5));
        else if constexpr (R == 5) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3), CTT_MP(4));
        else if constexpr (R == 4) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2), CTT_MP(3));
        else if constexpr (R == 3) mads(acc, CTT_MP(0), CTT_MP(1), CTT_MP(2));
        else if constexpr (R == 2) mads(acc, CTT_MP(0), CTT_MP(1));
        else mads(acc, CTT_MP(0));
#undef CTT_MP
        col_mp<K, LO + R, HI>(acc, m);
#else
        acc += (uint64_t)m[LO] * UP::P[K - LO];
        col_mp<K, LO + 1, HI>(acc, m);
#endif
      }
    }
  }
  CTT_HD static void mad1(uint64_t& acc, uint32_t a, uint32_t b) {
#if CTT_FPU_ASM
    madv(acc, a, b);
#else
    acc += (uint64_t)a * b;
#endif
  }
  CTT_HD static void mad1k(uint64_t& acc, uint32_t a, uint32_t k) {
#if CTT_FPU_ASM
    mads(acc, a, k);
#else
    acc += (uint64_t)a * k;
#endif
  }
  // products of column K of a*b
  template <int K>
  CTT_HD static void col_prod(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
    col_ab<K, (K < NL ? 0 : K - NL + 1), (K < NL ? K + 1 : NL)>(acc, a, b);
  }
  // column K of a square: cross products once, sum_{i < K-i} (2 a_i) a_{K-i}, then the diagonal term when K is even
  template <int K>
  CTT_HD static void col_sq(uint64_t& acc, const uint32_t* a, const uint32_t* a2) {
    col_ab<K, (K - NL + 1 > 0 ? K - NL + 1 : 0), ((K + 1) >> 1)>(acc, a2, a);
    if constexpr ((K & 1) == 0) mad1(acc, a[K >> 1], a[K >> 1]);
  }
  // Montgomery part of column K: the products of the quotient digits known so far with P; for K < NL the new digit
  // m[K] and its product with P[0] (the low LB bits of the column then vanish); for K >= NL the result limb.
  // Ends with the carry into the next column.
  template <int K>
  CTT_HD static void col_finish(uint64_t& acc, uint32_t* m, uint32_t* t) {
    col_mp<K, (K < NL ? 0 : K - NL + 1), (K < NL ? K : NL)>(acc, m);
    if constexpr (K < NL) {
      m[K] = ((uint32_t)acc * UP::M0INV) & MASK;
      mad1k(acc, m[K], UP::P[0]);
    } else {
      t[K - NL] = (uint32_t)acc & MASK;
    }
    acc >>= LB;
  }
  // compile-time loop over the columns
  template <int K, int END, class Fn>
  CTT_HD static void columns(Fn&& f) {
    if constexpr (K < END) {
      f(std::integral_constant<int, K>{});
      columns<K + 1, END>(f);
    }
  }
#define CTT_COL_LAMBDA(kc) [&](auto kc) __attribute__((always_inline))

  // Montgomery product a*b/R' (mod p), radix 2^LB, product scanning; limbs of a, b < 2^30, a*b < R'*p.
  CTT_HD static FpU mul(const FpU& a, const FpU& b) {
    uint64_t acc = 0;
    uint32_t m[NL];
    FpU t;
    columns<0, 2 * NL - 1>(CTT_COL_LAMBDA(kc) {
      constexpr int k = decltype(kc)::value;
      col_prod<k>(acc, a.l, b.l);
      col_finish<k>(acc, m, t.l);
    });
    t.l[NL - 1] = (uint32_t)acc;
    return t;
  }

  // (a*b + c*d)/R' (mod p): two products, one Montgomery reduction.  Needs a*b + c*d < R'*p.
  CTT_HD static FpU mul2(const FpU& a, const FpU& b, const FpU& c, const FpU& d) {
    uint64_t acc = 0;
    uint32_t m[NL];
    FpU t;
    columns<0, 2 * NL - 1>(CTT_COL_LAMBDA(kc) {
      constexpr int k = decltype(kc)::value;
      col_prod<k>(acc, a.l, b.l);
      col_prod<k>(acc, c.l, d.l);
      col_finish<k>(acc, m, t.l);
    });
    t.l[NL - 1] = (uint32_t)acc;
    return t;
  }

  // Two independent products written column by column side by side: a single product is one long dependency
  // chain through its column accumulator; a second, independent chain gives the in-order issue logic something to
  // do while a v_mad_u64_u32 result is in flight.  hipcc is left to schedule the two chains.
  CTT_HD static void mul_pair(const FpU& a, const FpU& b, const FpU& c, const FpU& d, FpU& r1, FpU& r2) {
    uint64_t acc1 = 0, acc2 = 0;
    uint32_t m1[NL], m2[NL];
    columns<0, 2 * NL - 1>(CTT_COL_LAMBDA(kc) {
      constexpr int k = decltype(kc)::value;
      col_prod<k>(acc1, a.l, b.l);
      col_prod<k>(acc2, c.l, d.l);
      col_finish<k>(acc1, m1, r1.l);
      col_finish<k>(acc2, m2, r2.l);
    });
    r1.l[NL - 1] = (uint32_t)acc1;
    r2.l[NL - 1] = (uint32_t)acc2;
  }

  // two independent squares, interleaved
  CTT_HD static void sqr_pair(const FpU& a, const FpU& c, FpU& r1, FpU& r2) {
    uint64_t acc1 = 0, acc2 = 0;
    uint32_t m1[NL], m2[NL], a2[NL], c2[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
      a2[i] = a.l[i] << 1;
      c2[i] = c.l[i] << 1;
    }
    columns<0, 2 * NL - 1>(CTT_COL_LAMBDA(kc) {
      constexpr int k = decltype(kc)::value;
      col_sq<k>(acc1, a.l, a2);
      col_sq<k>(acc2, c.l, c2);
      col_finish<k>(acc1, m1, r1.l);
      col_finish<k>(acc2, m2, r2.l);
    });
    r1.l[NL - 1] = (uint32_t)acc1;
    r2.l[NL - 1] = (uint32_t)acc2;
  }

  // square: cross products once with a doubled operand (2*a_i < 2^31 fits)
  CTT_HD static FpU sqr(const FpU& a) {
    uint64_t acc = 0;
    uint32_t m[NL], a2[NL];
    FpU t;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.l[i] << 1;
    columns<0, 2 * NL - 1>(CTT_COL_LAMBDA(kc) {
      constexpr int k = decltype(kc)::value;
      col_sq<k>(acc, a.l, a2);
      col_finish<k>(acc, m, t.l);
    });
    t.l[NL - 1] = (uint32_t)acc;
    return t;
  }

  // ---- conversions --------------------------------------------------------------------------------
  // bits [pos, pos+LB) of a little-endian 32-bit limb array of n words
  CTT_HD static uint32_t bits_at(const uint32_t* w, int n, int pos) {
    const int word = pos >> 5, sh = pos & 31;
    uint32_t v = word < n ? w[word] >> sh : 0u;
    if (sh + LB > 32 && word + 1 < n) v |= w[word + 1] << (32 - sh);
    return v & MASK;
  }
  // reference representation (Montgomery R, saturated, < p)  ->  this one (Montgomery R'); (raw) zero stays zero
  CTT_HD static FpU from_sat(const Sat& s) {
    FpU r, c;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      r.l[i] = bits_at(s.l, Sat::N, LB * i);
      c.l[i] = UP::C_IN[i];
    }
    return mul(r, c);
  }
};

// ---------------------------------------------------------------------------------------------
// Fp2 = Fp[i]/(i^2+1) over the carry-free base field.  Every component is normalised and bounded like a base
// element (MULB = 2: products come out of ONE Montgomery reduction each, < 2p):
//   mul:  c0 = a0*b0 + (Kp - a1)*b1,  c1 = a0*b1 + a1*b0        (two sum-of-products, towers.nim:852-878 prod2x)
//   sqr:  c0 = (a0 + a1)*(a0 - a1 + Kp),  c1 = a0*a1 + a0*a1    (square_complex, towers.nim:758-796)
// K = 12 covers every operand the EC formulas produce (components < 12p, ec.h); needs 2*K^2 < R'/p, true for
// BLS12-381 (R'/p = 2^11), not for the 29x9 fields.  No lazy operands here: two sum-of-products per column already
// use the 64-bit column budget.
// ---------------------------------------------------------------------------------------------
template <class UP>
struct Fp2<FpU<UP>> {
  using F = FpU<UP>;
  using Base = F;
  static constexpr int MULB = 2;
  static constexpr bool UNSAT = true;
  static constexpr int KNEG = 12;
  static constexpr int HEADROOM_LOG2 = UP::RP_OVER_P_LOG2;  // log2(R'/p): operand bounds k1*k2 must stay below it
  static constexpr int LB = F::LB;
  static constexpr int NL = 2 * F::NL;   // limbs of the whole element (c0 then c1)
  static_assert(2 * KNEG * KNEG < (1 << UP::RP_OVER_P_LOG2), "Fp2 over this base field needs more Montgomery headroom");
  F c0, c1;

  CTT_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
  CTT_HD static Fp2 one() { return {F::one(), F::zero()}; }
  CTT_HD bool is_zero() const { return c0.is_zero() & c1.is_zero(); }
  template <int B>
  CTT_HD bool is_zero_modp() const {
    return c0.template is_zero_modp<B>() && c1.template is_zero_modp<B>();
  }
  CTT_HD static Fp2 select(bool c, const Fp2& a, const Fp2& b) {
    return {F::select(c, a.c0, b.c0), F::select(c, a.c1, b.c1)};
  }
  CTT_HD static Fp2 add(const Fp2& a, const Fp2& b) { return {F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
  CTT_HD static Fp2 dbl(const Fp2& a) { return {F::dbl(a.c0), F::dbl(a.c1)}; }
  template <int B>
  CTT_HD static Fp2 sub(const Fp2& a, const Fp2& b) {
    return {F::template sub<B>(a.c0, b.c0), F::template sub<B>(a.c1, b.c1)};
  }
  template <int B>
  CTT_HD static Fp2 cneg(const Fp2& a, bool c) {
    return {F::template cneg<B>(a.c0, c), F::template cneg<B>(a.c1, c)};
  }
  template <int K>
  CTT_HD static Fp2 sub3(const Fp2& a, const Fp2& b, const Fp2& c) {
    return {F::template sub3<K>(a.c0, b.c0, c.c0), F::template sub3<K>(a.c1, b.c1, c.c1)};
  }
  // Karatsuba on unreduced columns (3 limb products, 2 reductions) was tried: the six live operands plus two sets
  // of Montgomery quotients spill the accumulate kernel to scratch (G2 accumulate 8.7 -> 126 ms); two sums of
  // products it is.
  // The negated / combined operands only feed products: they stay lazy (no carry propagation, FpU "lazy forms"); the
  // column budget of a sum of two products with one lazy operand each is LAZY_ONE, of (lazy sum) x (lazy difference) < LAZY_BOTH.
  static_assert(F::LAZY_BOTH, "Fp2 over this base field needs the 28-bit-limb column budget");
  CTT_HD static F add_lazy(const F& a, const F& b) {
    F r;
#pragma unroll
    for (int i = 0; i < F::NL; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
  }
  CTT_HD static Fp2 mul(const Fp2& a, const Fp2& b) {
    F na1 = F::template sub_lazy<KNEG - 1>(F::zero(), a.c1);   // KNEG*p - a1
    return {F::mul2(a.c0, b.c0, na1, b.c1), F::mul2(a.c0, b.c1, a.c1, b.c0)};
  }
  CTT_HD static Fp2 sqr(const Fp2& a) {
    F s = add_lazy(a.c0, a.c1);
    F d = F::template sub_lazy<KNEG - 1>(a.c0, a.c1);          // a0 - a1 + KNEG*p
    return {F::mul(s, d), F::mul(add_lazy(a.c0, a.c0), a.c1)};   // (a0+a1)(a0-a1), 2 a0 a1: two products, two reductions
  }
  // a*b + c*d: two products (not fused further: four base sum-of-products)
  CTT_HD static Fp2 mul2(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) { return add(mul(a, b), mul(c, d)); }
  CTT_HD static void mul_pair(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d, Fp2& r1, Fp2& r2) {
    r1 = mul(a, b);
    r2 = mul(c, d);
  }
  CTT_HD static void sqr_pair(const Fp2& a, const Fp2& c, Fp2& r1, Fp2& r2) {
    r1 = sqr(a);
    r2 = sqr(c);
  }
  CTT_HD static Fp2 from_sat(const Fp2<typename F::Sat>& s) { return {F::from_sat(s.c0), F::from_sat(s.c1)}; }
};

// Field-generic spellings used by ec.h: saturated fields ignore the bias/bound parameters.
template <class F, int B> CTT_HD F fsub(const F& a, const F& b) {
  if constexpr (F::UNSAT) return F::template sub<B>(a, b); else return F::sub(a, b);
}
template <class F, int B> CTT_HD F fcneg(const F& a, bool c) {
  if constexpr (F::UNSAT) return F::template cneg<B>(a, c); else return F::cneg(a, c);
}
// lazy operands (fpu.h "lazy forms"): which products of field F may take them
template <class F> struct LazyOps { static constexpr bool ONE = false, BOTH = false; };
template <class UP> struct LazyOps<FpU<UP>> { static constexpr bool ONE = FpU<UP>::LAZY_ONE, BOTH = FpU<UP>::LAZY_BOTH; };
// a - b + (B+1)*p, lazy when LZ (the caller names the product rule it relies on) -- else the normalised a - b + B*p.
// Either way the value is < bound(a) + B + 1.
template <class F, int B, bool LZ> CTT_HD F fsub_lz(const F& a, const F& b) {
  if constexpr (LZ) return F::template sub_lazy<B>(a, b); else return fsub<F, B>(a, b);
}
template <class F, int B, bool LZ> CTT_HD F fcneg_lz(const F& a, bool c) {
  if constexpr (LZ) return F::template cneg_lazy<B>(a, c); else return fcneg<F, B>(a, c);
}
// the condition as a lane mask word (0 / 0xffffffff): the carry-free base fields negate in arithmetic (FpU::cneg_m), every other
// field takes it as the bool it stands for
struct SignMask {
  uint32_t m;
  CTT_HD explicit SignMask(uint32_t entry_word) : m((uint32_t)((int32_t)entry_word >> 31)) {}   // sign = bit 31 of a sorted entry
};
template <class F> struct HasMaskNeg { static constexpr bool value = false; };
template <class UP> struct HasMaskNeg<FpU<UP>> { static constexpr bool value = true; };
template <class F, int B> CTT_HD F fcneg(const F& a, SignMask s) {
  if constexpr (HasMaskNeg<F>::value) return F::template cneg_m<B>(a, s.m); else return fcneg<F, B>(a, s.m != 0u);
}
template <class F, int B, bool LZ> CTT_HD F fcneg_lz(const F& a, SignMask s) {
  if constexpr (HasMaskNeg<F>::value && LZ) return F::template cneg_lazy_m<B>(a, s.m); else return fcneg<F, B>(a, s);
}
template <class F, bool LZ> CTT_HD F fnorm(const F& a) {
  if constexpr (LZ) return F::norm(a); else return a;
}
// a - b - 2c (+ K*p), normalised: b + 2c < (K-1)*p
template <class F, int K> CTT_HD F fsub3(const F& a, const F& b, const F& c) {
  if constexpr (F::UNSAT) return F::template sub3<K>(a, b, c); else return F::sub(F::sub(a, b), F::dbl(c));
}
// a*b - c*d with one reduction where the field supports it; B bounds d's partner c (c < B*p)
template <class F> struct IsFp2 { static constexpr bool value = false; };
template <class B> struct IsFp2<Fp2<B>> { static constexpr bool value = true; };
template <class F, int B> CTT_HD F fmul_sub(const F& a, const F& b, const F& c, const F& d) {
  if constexpr (F::UNSAT && IsFp2<F>::value) {
    return F::template sub<F::MULB>(F::mul(a, b), F::mul(c, d));   // < 2*MULB*p, the bound the callers assume
  } else if constexpr (F::UNSAT) {
    F nc = F::template sub<B>(F::zero(), c);   // B*p - c  (< B*p), so a*b + nc*d == a*b - c*d (mod p)
    return F::mul2(a, b, nc, d);
  } else {
    return F::sub(F::mul(a, b), F::mul(c, d));
  }
}
// a*b - c*d (c < B*p) with c's negation lazy when LZ
template <class F, int B, bool LZ> CTT_HD F fmul_sub_lz(const F& a, const F& b, const F& c, const F& d) {
  if constexpr (LZ) return F::mul2(a, b, F::template sub_lazy<B>(F::zero(), c), d); else return fmul_sub<F, B>(a, b, c, d);
}
// two independent products / squares: interleaved where the field supports it (fpu.h mul_pair)
template <class F> CTT_HD void fmul_pair(const F& a, const F& b, const F& c, const F& d, F& r1, F& r2) {
  if constexpr (F::UNSAT) {
    F::mul_pair(a, b, c, d, r1, r2);
  } else {
    r1 = F::mul(a, b);
    r2 = F::mul(c, d);
  }
}
template <class F> CTT_HD void fsqr_pair(const F& a, const F& c, F& r1, F& r2) {
  if constexpr (F::UNSAT) {
    F::sqr_pair(a, c, r1, r2);
  } else {
    r1 = F::sqr(a);
    r2 = F::sqr(c);
  }
}
template <class F, int B> CTT_HD bool fis_zero_modp(const F& a) {
  if constexpr (F::UNSAT) return a.template is_zero_modp<B>(); else return a.is_zero();
}

}  // namespace ctt
