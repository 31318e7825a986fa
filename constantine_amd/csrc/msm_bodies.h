// msm_bodies.h -- per-thread bodies of the MSM kernels + curve descriptors.
//
// The MI355X engine replaces the reference's CPU Pippenger
// (constantine/math/elliptic/ec_multi_scalar_mul.nim:204-296, ..._parallel.nim:148-431) with a
// sort-then-segmented-reduce pipeline:
//
//   sort     Booth signed window digits of every scalar (bigints.nim:806-859), then (point index, sign) entries
//            grouped by bucket per window: partition by bucket group + LDS sort per group  [kernels in msm_engine.hip]
//   accum    every lane sums K consecutive sorted entries into XYZZ accumulators
//            (mixed add = the reference's `accumulate`, ec_multi_scalar_mul.nim:177-184);
//            runs fully inside a lane's range go straight to the bucket array, runs that
//            straddle a lane boundary go to head/tail partial slots
//   merge    partial slots of one bucket are combined (tree over lanes, log steps)
//   reduce   sum_k k*B_k per window re-associated by the bits of k: one shared pairwise-sum pyramid,
//            depth c-1 (log-depth form of bucketReduce, ec_multi_scalar_mul.nim:186-197)
//   combine  Horner over (window, bit) on the host (ec_multi_scalar_mul.nim:250-254)
//
// Bodies are __host__ __device__ so that tests/emu can execute exactly this code on the CPU
// (logic check without a GPU); the product path runs them only as HIP kernels.
#pragma once
#include "ec.h"

namespace ctt {

static constexpr uint32_t DIGIT_NONE = 0xffffffffu;
static constexpr uint32_t KEY_NONE = 0xffffffffu;

// ---------------------------------------------------------------------------------------------
// Curve descriptors (a = 0 everywhere; only the field, scalar field and scalar width matter for MSM)
// constantine/named/config_fields_and_curves.nim:116-133,214-229,269-287
// ---------------------------------------------------------------------------------------------
// F  = coordinate field in the reference's representation (what the C API hands over and takes back)
// FD = coordinate field the kernels compute in (carry-free limbs where that is faster, see fpu.h)
// ACC_NS / RED_NS: ns per mixed addition of the accumulate kernel and per full addition of the reduction passes with the whole
// chip busy, measured on MI355X (profiles/): what the window-size cost model (msm_pipeline.h choose_window_bits) ranks plans by.
// NARROW_PRIO_LOG2N: the narrow reduction passes run at raised wave priority (hip_backend.h k_pyr_quad) for MSMs of up to 2^this pairs, 0 = never;
// measured per curve (profiles/wave_priority_r06.txt: it helps BLS12-381 G1 and BN254 up to 2^19, does nothing for G2 and costs Pallas 3 % at 2^16).
// WHOLE_TAIL_LOG2N: from 2^this pairs on the accumulation of a pipelined MSM waits for the WHOLE tail of the previous one instead of its wide passes only
// (msm_pipeline.h accumulate_pairs), 0 = never: the accumulate kernels of the 9-limb curves own every register of the chip (4 waves x 128), a narrow pass
// beside them crawls (2 ms instead of 7 us) and costs the accumulation 19 % -- Pallas / Vesta 2^22 5.58 -> 5.02 ms per MSM, BN254 2^21 3.03 -> 2.76
// (profiles/whole_tail_wait_r06.txt); BLS12-381 G1 leaves registers free and loses 5-19 % when made to wait.
struct Bls12381G1 { using F = Fp<BLS12_381_Fp>; using FD = FpU<BLS12_381_Fp_U>; using Fr = Fp<BLS12_381_Fr>; static constexpr int BITS = 255; static constexpr int ID = 0; static constexpr double ACC_NS = 0.142, RED_NS = 0.26; static constexpr int NARROW_PRIO_LOG2N = 19; static constexpr int WHOLE_TAIL_LOG2N = 0; };
struct Bls12381G2 { using F = Fp2<Fp<BLS12_381_Fp>>; using FD = Fp2<FpU<BLS12_381_Fp_U>>; using Fr = Fp<BLS12_381_Fr>; static constexpr int BITS = 255; static constexpr int ID = 1; static constexpr double ACC_NS = 0.467, RED_NS = 1.1; static constexpr int NARROW_PRIO_LOG2N = 0; static constexpr int WHOLE_TAIL_LOG2N = 0; };
struct Bn254G1 { using F = Fp<BN254_Fp>; using FD = FpU<BN254_Fp_U>; using Fr = Fp<BN254_Fr>; static constexpr int BITS = 254; static constexpr int ID = 2; static constexpr double ACC_NS = 0.0685, RED_NS = 0.12; static constexpr int NARROW_PRIO_LOG2N = 19; static constexpr int WHOLE_TAIL_LOG2N = 21; };
struct Bn254G2 { using F = Fp2<Fp<BN254_Fp>>; using FD = F; using Fr = Fp<BN254_Fr>; static constexpr int BITS = 254; static constexpr int ID = 3; static constexpr double ACC_NS = 0.5, RED_NS = 1.2; static constexpr int NARROW_PRIO_LOG2N = 0; static constexpr int WHOLE_TAIL_LOG2N = 0; };
struct PallasEc { using F = Fp<Pallas_Fp>; using FD = FpU<Pallas_Fp_U>; using Fr = Fp<Vesta_Fp>; static constexpr int BITS = 255; static constexpr int ID = 4; static constexpr double ACC_NS = 0.056, RED_NS = 0.10; static constexpr int NARROW_PRIO_LOG2N = 0; static constexpr int WHOLE_TAIL_LOG2N = 21; };
struct VestaEc { using F = Fp<Vesta_Fp>; using FD = FpU<Vesta_Fp_U>; using Fr = Fp<Pallas_Fp>; static constexpr int BITS = 255; static constexpr int ID = 5; static constexpr double ACC_NS = 0.056, RED_NS = 0.10; static constexpr int NARROW_PRIO_LOG2N = 0; static constexpr int WHOLE_TAIL_LOG2N = 21; };

// ---------------------------------------------------------------------------------------------
// Booth signed digits
// ---------------------------------------------------------------------------------------------
// c+1 bits of the 256-bit scalar k starting at bit `pos` (bits >= 256 read as 0)
CTT_HD uint32_t scalar_bits_at(const uint32_t* k, int pos, int nb) {
  int word = pos >> 5, sh = pos & 31;
  uint32_t lo = word < 8 ? k[word] >> sh : 0u;
  if (sh + nb > 32 && word + 1 < 8) lo |= k[word + 1] << (32 - sh);
  return lo & ((1u << nb) - 1u);
}

// Window layout (round 3).  W windows cover bits + 1 bits of the scalar -- the spare top bit is always zero, so the top digit
// never carries out and no window exists only for a carry -- with widths that differ by at most one: the first r windows (the
// low bits) are cb + 1 bits wide, the others cb (r = 0: all cb).  Rounds 1-2 cut the scalar into bits/c windows of c bits plus
// whatever remained (the reference's layout, ec_multi_scalar_mul.nim:278-289: 255 = 19 x 13 + 8): a top window of 8, 3 or 0 bits
// sends its N digits into 128, 4 or 1 buckets -- head chains of hundreds of lanes for the merge tree (6 tree steps at 2^16
// pairs, c = 13), one sort group that swallows N records, and window tables restricted to the few c with a wide remainder.
// Any layout yields the same group element: sum_w 2^off(w) * sum_b digit_b B_b.
struct WinLayout {
  int cb, r;
  CTT_HD int off(uint32_t w) const { return (int)w * cb + ((int)w < r ? (int)w : r); }
  CTT_HD int width(uint32_t w) const { return cb + ((int)w < r ? 1 : 0); }
  CTT_HD int cmax() const { return cb + (r > 0 ? 1 : 0); }
  CTT_HD bool is_wide(uint32_t w) const { return r == 0 || (int)w < r; }   // as wide as the widest window
};
// W windows of at most c bits over bits + 1 bits
CTT_HD WinLayout window_layout(int bits, int c, int* W) {
  const int T = bits + 1;
  const int nw = (T + c - 1) / c;
  WinLayout L;
  L.cb = T / nw;
  L.r = T - L.cb * nw;
  *W = nw;
  return L;
}

// digit of window w -> packed ((val-1)<<1 | neg) or DIGIT_NONE when val == 0 (Booth signed digits, bigints.nim:806-859, over
// the window's own width)
CTT_HD uint32_t booth_digit_packed(const uint32_t* k, int w, const WinLayout& L) {
  const int i = L.off((uint32_t)w), c = L.width((uint32_t)w);
  uint32_t d;
  if (i == 0) {
    d = (k[0] << 1) & ((1u << (c + 1)) - 1u);
  } else {
    d = scalar_bits_at(k, i - 1, c + 1);
  }
  uint32_t neg = d >> c;
  uint32_t e = (d + 1u) >> 1;
  uint32_t val = neg ? (1u << c) - e : e;
  val &= (1u << c) - 1u;
  return val ? (((val - 1u) << 1) | neg) : DIGIT_NONE;
}

// The digits of windows [w0, w0 + nw) of NS scalars held in registers, window by window: fn(w, d[NS]) with the packed digit
// of every scalar.  Same values as booth_digit_packed; the walk is organised by the 32-bit WORD a window starts in (an
// unrolled loop), so that the limbs are addressed with compile-time indices -- indexing k[] with the (wave-uniform, but
// run-time) word of a window costs the GPU an 8-way select chain per access (measured: k_part_count 129 -> 78 us at 2^22).
template <int NS, class Fn>
CTT_HD void for_each_digit(const uint32_t (&k)[NS][8], uint32_t w0, uint32_t nw, const WinLayout& L, Fn&& fn) {
  uint32_t w = w0;
  const uint32_t wend = w0 + nw;
#pragma unroll
  for (int word = 0; word < 8; word++) {
    while (w < wend) {
      const uint32_t i = (uint32_t)L.off(w);
      const int c = L.width(w);
      const uint32_t mask = (1u << (c + 1)) - 1u, vmask = (1u << c) - 1u;
      const uint32_t pos = i ? i - 1u : 0u;
      if ((pos >> 5) != (uint32_t)word) break;
      const uint32_t sh = pos & 31u;
      uint32_t dg[NS];
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const uint32_t lo = k[s][word], hi = word < 7 ? k[s][word < 7 ? word + 1 : 7] : 0u;
        uint32_t d = i ? (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) : lo << 1;
        d &= mask;
        const uint32_t neg = d >> c;
        const uint32_t e = (d + 1u) >> 1;
        const uint32_t val = (neg ? (1u << c) - e : e) & vmask;
        dg[s] = val ? (((val - 1u) << 1) | neg) : DIGIT_NONE;
      }
      fn(w, dg);
      w++;
    }
  }
}

// Digits + sort by bucket.  Output contract (what the accumulation consumes): for every bucket set r (one per window;
// a single one for all windows in the window-table form below), entries[r][0..bucket_start[r][B]) = (point index | sign << 31)
// of every non-zero digit, grouped by bucket in increasing bucket order (any order inside a bucket), bucket_start[r][b] =
// first position of bucket b, and maxcount[0] = size of the largest bucket over all sets.
//
// Window-table form (`merged`): the points are a table T[w][j] = 2^(c*w) * P_j (MsmEngine::prepare_table), so that the
// digit of window w of scalar j is a multiple of the table entry w*id_stride + j and ALL windows share ONE bucket set:
// W = 1, nent = Wd*n entries, the "point index" of an entry is the table row.  The partition pass then has one set of NG
// group regions that every digit window of a block fills (the runs of a block are Wd times longer than without a table),
// and its records are 64-bit: low bucket bits << 32 | sign << 31 | table row.
struct SortArgs {
  const uint32_t* scalars;  // [n][8] canonical
  uint32_t n;
  int c;                    // bits of the widest window: B = 2^(c-1)
  WinLayout lay;            // widths and offsets of the digit windows
  uint32_t W, B;            // bucket sets, buckets per set
  uint32_t Wd;              // digit windows per scalar (== W unless merged)
  uint32_t merged;          // 1: every digit window goes to bucket set 0
  uint32_t nent;            // capacity of one set's entry list: n, or Wd*n when merged
  uint32_t id_stride;       // merged: table entries per window (>= n: a prefix of the cached bases may be used)
  uint32_t NG, gshift;      // bucket groups per set; group = bucket >> gshift
  uint32_t gshift_narrow;   // same for the windows one bit narrower than c, whose digits only reach B/2 buckets (== gshift when merged)
  uint32_t slice, nblk;     // partition pass: scalars per block, number of blocks
  uint32_t jbits;           // bits of a point index: 32-bit record = low bucket bits << (jbits+1) | sign << jbits | index
  uint32_t* part;           // [W][nent] packed records partitioned by group (merged: 64-bit records, [nent])
  uint32_t* cntA;           // [nblk][W*NG] per-block group counts -> block offsets inside the group
  uint32_t* gtot;           // [W*NG] group sizes
  uint32_t* gbase;          // [W][NG+1] group start inside the set
  uint32_t* bstart;         // [W][B+1]
  uint32_t* entries;        // [W][nent]
  uint32_t* maxcount;       // [4]: [0] largest bucket, [2] the head merge's queue count; zeroed by the sort's first kernel (k_part_count)
  uint32_t cap, big;        // k_group_sort: entries per LDS tile; buckets above `big` bypass the LDS image
  uint32_t xcd_map = 1;     // partition kernels: neighbouring slices on one XCD (msm_engine.hip part_slice_of_block); 0 = slice b to block b
  uint32_t staged = 1;      // pass A sweep 2 through an LDS image of the block's output (k_part_scatter_staged); 0 = one store per record
  // Round 5: the sort clears the EMPTY buckets of the set it sorts for (zero_bytes = bytes of one bucket, [W][B] buckets at
  // zero_base; 0 = leave them alone).  The accumulation and the head merge write every non-empty bucket, so this is all of the
  // "bucket set = neutral" that the accumulation needs -- rounds 1-4 cleared the whole set with a fill launch between the sort and
  // the accumulation (8 us + a launch gap of a 2^16-pair MSM's chain).
  void* zero_base = nullptr;
  uint32_t zero_bytes = 0;
};

// Fr Montgomery -> canonical (batchFromField, finite_fields.nim:915-920)
template <class Fr>
CTT_HD void fr_from_mont_body(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t j) {
  if (j >= n) return;
  Fr a;
#pragma unroll
  for (int i = 0; i < Fr::N; i++) a.l[i] = in[(uint64_t)Fr::N * j + i];
  a = Fr::from_mont(a);
#pragma unroll
  for (int i = 0; i < Fr::N; i++) out[(uint64_t)Fr::N * j + i] = a.l[i];
}

// Converted points are stored one per 128-byte line: after the sort every lane gathers whole points by index,
// and a 112-byte (or 72-byte) record at its natural stride would straddle two cache lines most of the time.
// The word behind the coordinates is a flag: 1 = the affine neutral (0,0) -- the accumulate kernel tests one word
// instead of all the limbs of x and y.
template <class FD>
constexpr uint32_t gather_stride() {
  return sizeof(Affine<FD>) + 4 <= 128 ? 128u : sizeof(Affine<FD>) + 4 <= 256 ? 256u : 512u;
}
template <class FD>
constexpr uint32_t gather_flag_offset() { return (uint32_t)sizeof(Affine<FD>); }   // right behind the coordinates: same 16-byte chunk or the next
// 16-byte chunks of a record that carry data (coordinates + flag word)
template <class FD>
constexpr uint32_t gather_chunks() { return (gather_flag_offset<FD>() + 4u + 15u) / 16u; }

// Input points: reference representation -> device field (one pass per MSM; (0,0) stays (0,0))
template <class F, class FD>
CTT_HD void convert_point_body(const Affine<F>* in, void* out, uint32_t n, uint32_t j) {
  if (j >= n) return;
  Affine<F> p = in[j];
  Affine<FD> q;
  q.x = FD::from_sat(p.x);
  q.y = FD::from_sat(p.y);
  char* rec = (char*)out + (uint64_t)j * gather_stride<FD>();
  *(Affine<FD>*)rec = q;
  *(uint32_t*)(rec + gather_flag_offset<FD>()) = p.is_inf() ? 1u : 0u;
}

// ---------------------------------------------------------------------------------------------
// Bucket accumulation over the sorted entry list
// ---------------------------------------------------------------------------------------------
template <class F>
struct AccumArgs {
  const uint32_t* entries;       // [W][N]   idx | sign<<31, sorted by bucket within a window
  const uint32_t* bucket_start;  // [W][B+1] exclusive prefix of bucket sizes; [B] = entries in window
  const void* points;            // affine points, `point_stride` bytes apart
  uint32_t point_stride;
  XYZZ<F>* buckets;              // [W][B]   (pre-zeroed = neutral; INTO: the sums so far)
  XYZZ<F>* heads;                // [W][G]
  XYZZ<F>* tails;                // [W][G]
  uint32_t* hkey;                // [W][G]
  uint32_t* tkey;                // [W][G]
  uint32_t N, B, K, G;
};

// How a lane gets the record of its next entry.  The accumulation is a loop of {entry word -> record -> ~4500 (G1) to ~14000
// (G2) instructions of addition}; read on demand, the two dependent reads (1-2 us) stall the wave once per addition, and a
// G2 kernel runs ONE wave per SIMD.  The loop therefore requests the record of entry pos+1 (and the entry word of pos+2)
// before the addition of entry pos and collects it afterwards.  GatherDirect is the plain form (CPU emulator, saturated
// fields): request() only remembers the address.  The device form (hip_backend.h GatherLds) has the record delivered into
// LDS by global_load_lds -- no registers are held while the addition runs.
template <class F>
struct GatherDirect {
  const char* rec = nullptr;
  CTT_HD void request(const char* r) { rec = r; }
  CTT_HD void collect(Affine<F>& pt, bool& qinf) {
    pt = *(const Affine<F>*)rec;
    // carry-free fields: the record's flag word (convert_point_body); reference layout: test x and y
    if constexpr (F::UNSAT) qinf = *(const uint32_t*)(rec + gather_flag_offset<F>()) != 0u; else qinf = pt.is_inf();
  }
};

// State of a lane's walk over its K sorted entries: which bucket it is in and where that bucket ends.
struct BucketWalk {
  const uint32_t* bs;
  uint32_t B, b, bend, bend2;
  bool head0;   // the bucket the lane starts in began before the lane's range (its partial sum is a head)
  // bucket containing position p0: bs[b] <= p0 < bs[b+1]
  CTT_HD void start(const uint32_t* bs_, uint32_t B_, uint32_t p0) {
    bs = bs_;
    B = B_;
    uint32_t lo = 0, hi = B;
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (bs[mid] <= p0) lo = mid; else hi = mid;
    }
    b = lo;
    head0 = bs[b] < p0;
    bend = bs[b + 1];
    bend2 = bs[b + 2 <= B ? b + 2 : B];   // the boundary after the next one, requested ahead: no dependent read when a bucket ends
  }
  // pos == bend: move to the bucket that holds position pos (empty buckets skipped; terminates because pos < bs[B])
  CTT_HD void next(uint32_t pos) {
    b++;
    bend = bend2;
    while (bend == pos) {
      b++;
      bend = bs[b + 1];
    }
    bend2 = bs[b + 2 <= B ? b + 2 : B];
  }
};

// INTO (a later slice of a host-pointer MSM, MsmEngine::submit_host): the bucket set already holds the sums of the earlier slices.
// A lane that STARTS a bucket inside its range -- exactly one lane per non-empty bucket does -- begins with the stored sum
// instead of an empty accumulator: a 2-line load per bucket where a separate bucket set per slice costs a full addition per
// bucket and slice afterwards (rounds 2-3: k_bucket_sum).  Heads (runs that began in an earlier lane) start empty as always, and the head /
// tail merge is unchanged: the tail piece it starts from carries the stored sum.
template <class F, class G, bool INTO = false>
CTT_HD void accum_body_xyzz(const AccumArgs<F>& a, uint32_t w, uint32_t g, G& gq) {
  if (g >= a.G) return;
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  const uint64_t slot = (uint64_t)w * a.G + g;
  const uint32_t nw = bs[a.B];
  const uint64_t p0l = (uint64_t)g * a.K;
  uint32_t hk = KEY_NONE, tk = KEY_NONE;
  if (p0l >= nw) {
    a.hkey[slot] = hk;
    a.tkey[slot] = tk;
    return;
  }
  const uint32_t p0 = (uint32_t)p0l;
  const uint32_t p1 = (p0l + a.K < nw) ? p0 + a.K : nw;
  BucketWalk bw;
  bw.start(bs, a.B, p0);
  bool first_run = true;
  // the accumulator's "neutral" state lives in a flag (xyzz_madd_flag): nothing to zero when a run is flushed
  XYZZ<F> acc;
  bool empty = true;
  if constexpr (INTO) {
    if (!bw.head0) {
      acc = a.buckets[(uint64_t)w * a.B + bw.b];
      empty = acc.is_inf();
    }
  }
  const uint32_t* ent = a.entries + (uint64_t)w * a.N;
  auto record = [&](uint32_t e) {
    return (const char*)__builtin_assume_aligned((const char*)a.points + (uint64_t)(e & 0x7fffffffu) * a.point_stride, 16);
  };
  const uint32_t plast = p1 - 1;
  uint32_t e = ent[p0];
  uint32_t e1 = ent[p0 < plast ? p0 + 1 : plast];
  gq.request(record(e));
  for (uint32_t pos = p0; pos < p1; pos++) {
    Affine<F> pt;
    bool qinf;
    gq.collect(pt, qinf);   // everything this lane has in flight is one addition old here
    if (pos == bw.bend) {
      // bucket b is finished inside this lane's range
      if (empty) acc = XYZZ<F>::inf();
      if (first_run && bw.head0) {
        a.heads[slot] = acc;
        hk = bw.b;
      } else {
        a.buckets[(uint64_t)w * a.B + bw.b] = acc;
      }
      first_run = false;
      empty = true;
      bw.next(pos);
      if constexpr (INTO) {
        acc = a.buckets[(uint64_t)w * a.B + bw.b];
        empty = acc.is_inf();
      }
    }
    gq.request(record(e1));                                        // entry pos+1 (the last one again at the end)
    const uint32_t e2 = ent[pos + 2 < p1 ? pos + 2 : plast];
    if (!qinf) xyzz_madd_flag<F, SignMask>(acc, empty, pt.x, pt.y, SignMask(e));
    e = e1;
    e1 = e2;
  }
  if (empty) acc = XYZZ<F>::inf();
  const uint32_t b = bw.b;
  const bool started_before = first_run && bw.head0;
  const bool ends_after = bw.bend > p1;
  if (started_before) {
    a.heads[slot] = acc;
    hk = b;
  } else if (ends_after) {
    a.tails[slot] = acc;
    tk = b;
  } else {
    a.buckets[(uint64_t)w * a.B + b] = acc;
  }
  a.hkey[slot] = hk;
  a.tkey[slot] = tk;
}

// the accumulate body in the X, Y + ZZ/ZZZ-holder form (ec.h xyzz_madd_core): used for the quadratic-extension fields.
// Everything is read ON DEMAND here -- the boundary of the current bucket, the walk over empty buckets, the entry word, then the record:
// 12.4 % of the wave cycles parked on s_waitcnt, 80.6 % VALU-busy with ONE wave per SIMD (483 of 512 registers).
// Round 3 (pipelined loop) and round 6 (record into LDS with the loop order unchanged; entry words and bucket boundaries one step ahead in
// registers; four waves of a CU in lockstep; a staggered start) each measured ~10 % SLOWER -- and round 6 found out why, and it is not the
// loop: THE KERNEL'S SPEED DEPENDS ON THE 8-BYTE PHASE OF ITS INSTRUCTION STREAM.  With one wave per SIMD an 8-byte instruction whose address
// is 4 mod 8 costs extra issue time; in the shipped code object 73 % of the 18 097 v_mad_u64_u32 of an iteration sit at 0 mod 8, and every
// one of those variants happened to shift the stream by an odd number of dwords (27 % at 0 mod 8).  The proof is a build that differs from
// the shipped one by s_nop instructions at the kernel's entry only: same phase 7.76 ms per launch at 2^20, flipped phase 8.65 ms (+11.5 %)
// (tools/phase_stats.py, profiles/g2_gather_r06.txt).  Re-measured at the good phase, the record-into-LDS form is LEVEL with this loop
// (9.42 against 9.39 ms per MSM at 2^20, 2.96 against 3.03 at 2^18) and the register look-ahead form 1.5 % faster (9.26 against 9.40):
// the parked cycles are not the dependent reads.  This loop stays; hip_backend.h k_accum has the knob that flips the phase
// (CTT_G2_PHASE_NOPS) and __graft_entry__.build() reports the aligned share of the built object, so that a change that flips it is seen.
// The kernels with two or more waves per SIMD (every other curve) do not care: their multiply-adds are split 50 / 50 and a shifted stream
// measures the same.
template <class F, class Z, bool INTO = false>
CTT_HD void accum_body_z(const AccumArgs<F>& a, uint32_t w, uint32_t g, Z& z) {
  if (g >= a.G) return;
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  const uint64_t slot = (uint64_t)w * a.G + g;
  const uint32_t nw = bs[a.B];
  const uint64_t p0l = (uint64_t)g * a.K;
  uint32_t hk = KEY_NONE, tk = KEY_NONE;
  if (p0l >= nw) {
    a.hkey[slot] = hk;
    a.tkey[slot] = tk;
    return;
  }
  const uint32_t p0 = (uint32_t)p0l;
  const uint32_t p1 = (p0l + a.K < nw) ? p0 + a.K : nw;
  // bucket containing position p0: bs[lo] <= p0 < bs[hi]
  uint32_t lo = 0, hi = a.B;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (bs[mid] <= p0) lo = mid; else hi = mid;
  }
  uint32_t b = lo;
  uint32_t bend = bs[b + 1];
  bool first_run = true;
  // the accumulator's "neutral" state lives in a flag (xyzz_madd_core): nothing to zero when a run is flushed
  F X, Y;
  bool empty = true;
  auto current = [&]() {
    XYZZ<F> r;
    if (empty) {
      r = XYZZ<F>::inf();
    } else {
      r.x = X;
      r.y = Y;
      z.get(r.zz, r.zzz);
    }
    return r;
  };
  auto resume = [&](uint32_t bb) {   // INTO: start from the stored sum of the earlier slices (see accum_body_xyzz)
    const XYZZ<F> r = a.buckets[(uint64_t)w * a.B + bb];
    empty = r.is_inf();
    X = r.x;
    Y = r.y;
    z.put(r.zz, r.zzz);
  };
  if constexpr (INTO) {
    if (!(bs[b] < p0)) resume(b);
  }
  const uint32_t* ent = a.entries + (uint64_t)w * a.N;
  for (uint32_t pos = p0; pos < p1; pos++) {
    if (pos == bend) {
      // bucket b is finished inside this lane's range
      if (first_run && bs[b] < p0) {
        a.heads[slot] = current();
        hk = b;
      } else {
        a.buckets[(uint64_t)w * a.B + b] = current();
      }
      first_run = false;
      empty = true;
      b++;
      while (bs[b + 1] == pos) b++;  // skip empty buckets; terminates because pos < nw
      bend = bs[b + 1];
      if constexpr (INTO) resume(b);
    }
    uint32_t e = ent[pos];
    const char* rec = (const char*)__builtin_assume_aligned((const char*)a.points + (uint64_t)(e & 0x7fffffffu) * a.point_stride, 16);
    Affine<F> pt = *(const Affine<F>*)rec;
    bool qinf;   // carry-free fields: the record's flag word (convert_point_body); reference layout: test x and y
    if constexpr (F::UNSAT) qinf = *(const uint32_t*)(rec + gather_flag_offset<F>()) != 0u; else qinf = pt.is_inf();
    if (!qinf) xyzz_madd_core<F, Z>(X, Y, z, empty, pt.x, pt.y, (e >> 31) != 0);
  }
  const bool started_before = first_run && bs[b] < p0;
  const bool ends_after = bend > p1;
  if (started_before) {
    a.heads[slot] = current();
    hk = b;
  } else if (ends_after) {
    a.tails[slot] = current();
    tk = b;
  } else {
    a.buckets[(uint64_t)w * a.B + b] = current();
  }
  a.hkey[slot] = hk;
  a.tkey[slot] = tk;
}
template <class F, class G, bool INTO = false>
CTT_HD void accum_body(const AccumArgs<F>& a, uint32_t w, uint32_t g, G& gq) {
  if constexpr (IsFp2<F>::value && F::UNSAT) {
    ZInRegs<F> z;
    accum_body_z<F, ZInRegs<F>, INTO>(a, w, g, z);
  } else {
    accum_body_xyzz<F, G, INTO>(a, w, g, gq);
  }
}
template <class F, bool INTO = false>
CTT_HD void accum_body(const AccumArgs<F>& a, uint32_t w, uint32_t g) {
  GatherDirect<F> gq;
  accum_body<F, GatherDirect<F>, INTO>(a, w, g, gq);
}

// ---------------------------------------------------------------------------------------------
// Merging the partial sums of buckets that straddle lane ranges
// ---------------------------------------------------------------------------------------------
template <class F>
struct MergeArgs {
  const uint32_t* bucket_start;
  XYZZ<F>* buckets;
  XYZZ<F>* heads;
  const XYZZ<F>* tails;
  const uint32_t* hkey;
  const uint32_t* tkey;
  const uint32_t* maxcount;  // device word: largest bucket size over all windows
  uint32_t B, K, G;
  uint32_t* queue = nullptr;   // queue form (merge_tail_queue_body): first-head slots of the chains with more than one head, [merge_queue_capacity]
  uint32_t* qcount = nullptr;  // entries in the queue (zeroed before the tail merge)
};

// longest possible chain of heads: a bucket of m entries spans at most floor((m-1)/K)+1 lanes (maxcount = the largest
// bucket, written by the sort before the accumulation starts: the merge kernels read it on the device, the host never waits)
template <class F>
CTT_HD uint32_t merge_chain_bound(const MergeArgs<F>& a) {
  const uint32_t mc = *a.maxcount;
  return mc ? (mc - 1) / a.K + 1 : 0;
}

// heads[g+1] += tails[g]: a tail is the first piece of a straddling bucket, the next lane's head continues it.
// When no bucket spans more than two lanes (every chain of heads has length one) the sum is the bucket.
template <class F>
CTT_HD void merge_tail_body(const MergeArgs<F>& a, uint32_t w, uint32_t g) {
  if (g + 1 >= a.G) return;
  const uint64_t slot = (uint64_t)w * a.G + g;
  const uint32_t b = a.tkey[slot];
  if (b == KEY_NONE) return;
  const bool final_ = merge_chain_bound<F>(a) <= 1;
  XYZZ<F> h = a.heads[slot + 1];
  XYZZ<F> t = a.tails[slot];
  XYZZ<F> r = xyzz_add_inl<F>(h, t);
  if (final_) a.buckets[(uint64_t)w * a.B + b] = r; else a.heads[slot + 1] = r;
}

// tree step over the chain of heads of one bucket: heads[g] += heads[g+d] for g-chain_start = 0 mod 2d
// (true when lane g has an addition to do in this step)
template <class F>
CTT_HD bool merge_step_active(const MergeArgs<F>& a, uint32_t w, uint32_t g, uint32_t d) {
  if (g >= a.G) return false;
  // longest possible chain is floor((maxcount-1)/K)+1 heads
  const uint32_t mc = *a.maxcount;
  if (mc == 0 || d >= (mc - 1) / a.K + 1) return false;
  const uint64_t slot = (uint64_t)w * a.G + g;
  const uint32_t b = a.hkey[slot];
  if (b == KEY_NONE) return false;
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  const uint32_t s = bs[b] / a.K + 1;
  const uint32_t e = (bs[b + 1] - 1) / a.K;
  const uint32_t rel = g - s;
  return (rel % (2 * d)) == 0 && g + d <= e;
}
template <class F>
CTT_HD void merge_step_body(const MergeArgs<F>& a, uint32_t w, uint32_t g, uint32_t d) {
  if (!merge_step_active<F>(a, w, g, d)) return;
  const uint64_t slot = (uint64_t)w * a.G + g;
  XYZZ<F> x = a.heads[slot];
  XYZZ<F> y = a.heads[slot + d];
  a.heads[slot] = xyzz_add_inl<F>(x, y);
}

// the first head of each chain now holds the bucket sum
template <class F>
CTT_HD void merge_final_body(const MergeArgs<F>& a, uint32_t w, uint32_t g) {
  if (g >= a.G) return;
  const uint64_t slot = (uint64_t)w * a.G + g;
  const uint32_t b = a.hkey[slot];
  if (b == KEY_NONE) return;
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  if (g != bs[b] / a.K + 1) return;
  a.buckets[(uint64_t)w * a.B + b] = a.heads[slot];
}

// The tree steps d = first_d, 2 first_d, ... that the wide step kernels launched before it did not cover, one workgroup per
// window (the GPU kernel strides its lanes over g and puts a workgroup barrier where `sync` is called).  The host enqueues as
// many wide steps as an ordinary digit distribution needs WITHOUT knowing the largest bucket (MsmPlan::merge_steps); for
// such inputs this returns at once.  An adversarial input (all scalars equal: chains of G heads) finishes its tree here.
template <class F, class Sync>
CTT_HD void merge_finish_body(const MergeArgs<F>& a, uint32_t w, uint32_t first_d, uint32_t lane, uint32_t nlanes, Sync&& sync) {
  const uint32_t chain = merge_chain_bound<F>(a);
  for (uint32_t d = first_d; d < chain; d <<= 1) {
    for (uint32_t g = lane; g < a.G; g += nlanes) merge_step_body<F>(a, w, g, d);
    sync();
  }
}

// ---------------------------------------------------------------------------------------------
// Queue form of the head merge (round 5).  The tree above costs a launch per level -- tail merge, log2(chain) steps, the
// finishing launch, the final copy: six launches and 120-130 us for a 2^16-pair MSM whose chains hold one to three heads (K = 11
// entries per lane, ~16 per bucket) -- and every level behind the first is a grid of G lanes (4 G in the quad form) of which a few
// per wave have an addition to do: the first tree step alone took 46 us for 36 k additions.  When the plan expects SHORT chains:
//   launch 1  merge_tail_queue_body: the tail merge, heads[s] += tails[s-1] (dense: nearly every lane has a tail).  A chain of ONE
//             head is finished -- the sum goes to the bucket (the tree form only knew that when NO chain of the MSM was longer);
//             the first head of a longer chain is appended to a queue (one atomic per wave);
//   launch 2  merge_queue_body: one lane per queued chain adds the heads s .. e one after the other -- dense again, since the
//             queue holds nothing but chains with work left, and two or three additions deep;
//   launch 3  merge_long_body (one workgroup per window) takes the chains longer than `lmax` heads -- skewed digit
//             distributions, "all scalars equal" -- as a tree; it returns at once when the largest bucket says there is none.
// (Measured first, same box: ONE launch in which the lane of a chain's first head walks tail + all heads -- every wave then runs
// as long as its longest chain with a third of its lanes idle from the start: 130-140 us at 2^16 against the tree's 118-123.)
// ---------------------------------------------------------------------------------------------
// first / last lane that holds a head of bucket b (b straddles lane ranges): the tail sits in lane s - 1
CTT_HD uint32_t chain_first(const uint32_t* bs, uint32_t K, uint32_t b) { return bs[b] / K + 1; }
CTT_HD uint32_t chain_last(const uint32_t* bs, uint32_t K, uint32_t b) { return (bs[b + 1] - 1) / K; }

// capacity of the queue: a chain of two or more heads takes two lanes, so at most every second lane of a window starts one
CTT_HD uint32_t merge_queue_capacity(uint32_t W, uint32_t G) { return W * (G / 2u + 1u); }

// lane g: the tail merge.  Returns true when the chain that starts in lane g + 1 has more than one head: *item = its first
// head's slot, for the queue (the caller appends it: wave-aggregated on the GPU).
template <class F>
CTT_HD bool merge_tail_queue_body(const MergeArgs<F>& a, uint32_t w, uint32_t g, uint32_t* item) {
  if (g + 1 >= a.G) return false;
  const uint64_t slot = (uint64_t)w * a.G + g;
  const uint32_t b = a.tkey[slot];
  if (b == KEY_NONE) return false;
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  const XYZZ<F> h = a.heads[slot + 1], t = a.tails[slot];
  const XYZZ<F> r = xyzz_add_inl<F>(h, t);
  const bool more = chain_last(bs, a.K, b) > g + 1;
  if (more) {
    a.heads[slot + 1] = r;
    *item = (uint32_t)(slot + 1);
  } else {
    a.buckets[(uint64_t)w * a.B + b] = r;
  }
  return more;
}

// queue entry qi: the rest of a chain of 2 .. lmax heads (the tail is in its first head already)
template <class F>
CTT_HD void merge_queue_body(const MergeArgs<F>& a, uint32_t qi, uint32_t lmax) {
  if (qi >= *a.qcount) return;
  const uint32_t slot = a.queue[qi];
  const uint32_t w = slot / a.G, s = slot - w * a.G;
  const uint32_t b = a.hkey[slot];
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  const uint32_t len = chain_last(bs, a.K, b) - s + 1;
  if (len > lmax) return;                       // merge_long_body's
  XYZZ<F> r = a.heads[slot];
  for (uint32_t i = 1; i < len; i++) {
    const XYZZ<F> h = a.heads[slot + i];
    r = xyzz_add_inl<F>(r, h);
  }
  a.buckets[(uint64_t)w * a.B + b] = r;
}

// the chains merge_queue_body left (more than lmax heads; their tails are merged), one workgroup per window: the tree over
// the heads, then the chain heads into the buckets, with a workgroup barrier (`sync`) between the levels
template <class F, class Sync>
CTT_HD void merge_long_body(const MergeArgs<F>& a, uint32_t w, uint32_t lmax, uint32_t lane, uint32_t nlanes, Sync&& sync) {
  const uint32_t chain = merge_chain_bound<F>(a);
  if (chain <= lmax) return;                    // (uniform: every lane reads the same word)
  const uint32_t* bs = a.bucket_start + (uint64_t)w * (a.B + 1);
  auto is_long = [&](uint32_t b) { return chain_last(bs, a.K, b) - chain_first(bs, a.K, b) + 1 > lmax; };
  for (uint32_t d = 1; d < chain; d <<= 1) {
    for (uint32_t g = lane; g < a.G; g += nlanes) {
      if (!merge_step_active<F>(a, w, g, d)) continue;
      const uint64_t slot = (uint64_t)w * a.G + g;
      if (!is_long(a.hkey[slot])) continue;
      const XYZZ<F> x = a.heads[slot], y = a.heads[slot + d];
      a.heads[slot] = xyzz_add_inl<F>(x, y);
    }
    sync();
  }
  for (uint32_t g = lane; g < a.G; g += nlanes) {
    const uint64_t slot = (uint64_t)w * a.G + g;
    const uint32_t b = a.hkey[slot];
    if (b == KEY_NONE || g != chain_first(bs, a.K, b) || !is_long(b)) continue;
    a.buckets[(uint64_t)w * a.B + b] = a.heads[slot];
  }
}

// ---------------------------------------------------------------------------------------------
// Bucket reduction  sum_b (b+1) * B_b  per window, in log depth.
//
// The reference's bucketReduce (ec_multi_scalar_mul.nim:186-197) is a serial running sum: 2*2^(c-1)
// dependent additions.  On the GPU a dependent chain of EC additions costs ~23 us per link (one lane
// cannot go faster than one SIMD), so the sum is re-associated by the bits of the bucket index:
//
//     sum_b b*B_b = sum_l 2^l * O_l ,   O_l = sum of the buckets whose index has bit l set.
//
// All O_l come out of ONE shared pairwise-sum pyramid: level l+1 = pairwise sums of level l (aligned
// blocks of 2^l buckets), and O_l = sum of the odd-indexed elements of level l.  Total work is 2*2^(c-1)
// additions (same as the running sum) but the depth is c-1 additions.  Pass p builds pyramid level p+1,
// starts the odd-element tree of level p and halves the trees of the earlier levels.  The host finishes
// with a Horner over the bits of each window (on the device, k_window_sums) and over the windows (msm_pipeline.h).
// ---------------------------------------------------------------------------------------------
template <class F>
struct PyrArgs {
  const XYZZ<F>* buckets;  // [W][B]  pyramid level 0
  XYZZ<F>* pyr;            // [W][B]  levels >= 1, level l at offset B - (B >> (l-1))
  XYZZ<F>* q;              // [W][B/2] odd-element trees, tree l at offset B/2 - (B >> (l+1))
  XYZZ<F>* out;            // [W][c]  O_0 .. O_{c-2}, then TOP = sum of all buckets
  uint32_t B;
  int c;
  int p;                   // pass index, 0 .. c-2
  uint32_t out_stride;     // elements between out[l] and out[l+1] (1 = contiguous; the block-local form writes columns)
};

template <class F>
CTT_HD const XYZZ<F>* pyr_level(const PyrArgs<F>& a, uint32_t w, int l) {
  if (l == 0) return a.buckets + (uint64_t)w * a.B;
  return a.pyr + (uint64_t)w * a.B + (a.B - (a.B >> (l - 1)));
}

// number of tasks (lanes) of pass p per window
CTT_HD uint32_t pyr_pass_tasks(uint32_t B, int c, int p) {
  uint32_t n = B >> (p + 1);          // (a) pyramid level p+1
  uint32_t L = B >> (p + 1);          // (b) odd elements of level p
  n += (L >= 2) ? L / 2 : 1;
  for (int l = 0; l < p; l++) {       // (c) halving of earlier trees
    uint32_t Ll = B >> (l + 1);
    if (Ll < 2) continue;
    uint32_t len = (Ll / 2) >> (p - l - 1);
    if (len >= 2) n += len / 2;
  }
  return n;
}

// Task t of pass a.p in window w: *d1 (and *d2 when set) = *s1 + *s2 (a plain copy when s2 is null).
// Returns false when t is not a task of this pass.
template <class F>
CTT_HD bool pyr_decode(const PyrArgs<F>& a, uint32_t w, uint32_t t, const XYZZ<F>*& s1, const XYZZ<F>*& s2, XYZZ<F>*& d1,
                       XYZZ<F>*& d2) {
  const uint32_t B = a.B;
  const int c = a.c, p = a.p;
  XYZZ<F>* out = a.out + (uint64_t)w * c * a.out_stride;
  const uint64_t os = a.out_stride;
  s1 = nullptr;
  s2 = nullptr;
  d1 = nullptr;
  d2 = nullptr;
  const uint32_t na = B >> (p + 1);                 // (a) pyramid: level p+1 from level p
  const uint32_t L = B >> (p + 1);                  // (b) odd elements of level p
  const uint32_t nb = (L >= 2) ? L / 2 : 1;
  if (t < na) {
    const XYZZ<F>* src = pyr_level<F>(a, w, p);
    s1 = src + 2 * t;
    s2 = src + 2 * t + 1;
    d1 = a.pyr + (uint64_t)w * B + (B - (B >> p)) + t;  // level p+1
    if (p + 1 == c - 1) d2 = out + (uint64_t)(c - 1) * os;  // TOP
  } else if (t - na < nb) {
    const uint32_t u = t - na;
    const XYZZ<F>* src = pyr_level<F>(a, w, p);
    if (L >= 2) {
      s1 = src + 2 * u + 1;
      s2 = src + 2 * (u + L / 2) + 1;
      d1 = a.q + (uint64_t)w * (B / 2) + (B / 2 - (B >> (p + 1))) + u;
      if (L / 2 == 1) d2 = out + (uint64_t)p * os;
    } else {
      s1 = src + 1;   // single odd element: O_p is a copy
      d1 = out + (uint64_t)p * os;
    }
  } else {
    uint32_t u = t - na - nb;                       // (c) halve the trees started in earlier passes
    for (int l = 0; l < p; l++) {
      const uint32_t Ll = B >> (l + 1);
      if (Ll < 2) continue;
      const uint32_t len = (Ll / 2) >> (p - l - 1);
      if (len < 2) continue;
      const uint32_t nc = len / 2;
      if (u < nc) {
        XYZZ<F>* qd = a.q + (uint64_t)w * (B / 2) + (B / 2 - (B >> (l + 1)));
        s1 = qd + u;
        s2 = qd + u + nc;
        d1 = qd + u;
        if (nc == 1) d2 = out + (uint64_t)l * os;
        break;
      }
      u -= nc;
    }
  }
  return s1 != nullptr;
}

template <class F>
CTT_HD void pyr_body(const PyrArgs<F>& a, uint32_t w, uint32_t t) {
  // Decode the task into (src1, src2, dst, dst2) first and run ONE addition afterwards, so that the
  // lanes of a wave that hold different task kinds still execute the (expensive) addition together.
  const XYZZ<F>* s1;
  const XYZZ<F>* s2;
  XYZZ<F>* d1;
  XYZZ<F>* d2;
  if (!pyr_decode<F>(a, w, t, s1, s2, d1, d2)) return;
  XYZZ<F> x = *s1;
  if (s2) {
    XYZZ<F> y = *s2;
    x = xyzz_add_inl<F>(x, y);
  }
  *d1 = x;
  if (d2) *d2 = x;
}

// groups of the bit Horner: group g covers bits [g*h, min((g+1)*h, c-1)) of the bucket index
CTT_HD int horner_groups(int c, int h) { return c > 1 ? (c - 1 + h - 1) / h : 1; }

// ---------------------------------------------------------------------------------------------
// sum_reduce (ec_shortweierstrass_batch_ops.nim:649-663): every point belongs to ONE bucket, so the bucket
// accumulation and head-merging kernels above do the whole job; this body only writes the trivial "sort" result
// (identity entry list, bucket_start = {0, n}, largest bucket = n).
// ---------------------------------------------------------------------------------------------
CTT_HD void iota_body(uint32_t* entries, uint32_t n, uint32_t* bucket_start, uint32_t* maxcount, uint32_t j) {
  if (j == 0) {
    bucket_start[0] = 0;
    bucket_start[1] = n;
    *maxcount = n;
  }
  if (j < n) entries[j] = j;
}

// ---------------------------------------------------------------------------------------------
// batchAffine (ec_shortweierstrass_batch_ops.nim:44-106 Prj, :108-178 Jac, vartime twins :187-345):
// Montgomery's simultaneous inversion.  One lane owns K consecutive points and one inversion; like the reference
// it parks the running products in dst[i].x, so no scratch memory is needed.  Neutral inputs (Z = 0) are skipped
// in the product chain and come out as the affine neutral (0,0).
// ---------------------------------------------------------------------------------------------
enum { SRC_JAC = 1, SRC_PRJ = 2 };

template <class F>
struct BatchAffineArgs {
  const F* src;     // [n][3]  X, Y, Z
  Affine<F>* dst;   // [n]
  uint32_t n;
  int kind;         // SRC_JAC: x = X/Z^2, y = Y/Z^3 ; SRC_PRJ: x = X/Z, y = Y/Z
  uint32_t K;       // points per lane
};

template <class F>
CTT_HD void batch_affine_body(const BatchAffineArgs<F>& a, uint32_t lane) {
  const uint64_t i0 = (uint64_t)lane * a.K;
  if (i0 >= a.n) return;
  const uint64_t i1 = (i0 + a.K < a.n) ? i0 + a.K : a.n;
  F run = F::one();
  for (uint64_t i = i0; i < i1; i++) {
    F z = a.src[3 * i + 2];
    if (!z.is_zero()) run = F::mul(run, z);
    a.dst[i].x = run;
  }
  F inv = F::inv(run);
  for (uint64_t i = i1; i-- > i0;) {
    F z = a.src[3 * i + 2];
    if (z.is_zero()) {
      a.dst[i] = Affine<F>::inf();
      continue;
    }
    F prev = (i > i0) ? a.dst[i - 1].x : F::one();
    F zi = F::mul(inv, prev);  // 1 / z_i
    inv = F::mul(inv, z);
    F X = a.src[3 * i], Y = a.src[3 * i + 1];
    Affine<F> o;
    if (a.kind == SRC_JAC) {
      F zi2 = F::sqr(zi);
      o.x = F::mul(X, zi2);
      o.y = F::mul(Y, F::mul(zi2, zi));
    } else {
      o.x = F::mul(X, zi);
      o.y = F::mul(Y, zi);
    }
    a.dst[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// Quotient polynomial of a KZG opening in evaluation form (kzg_prove, commitments/kzg.nim:204-223; getQuotientPoly,
// math/polynomials/polynomials.nim): for p given by its n evaluations over the (bit-reversed) roots of unity w_i and an
// opening point z that is NOT one of them,
//     y = p(z) = (z^n - 1)/n * sum_i p_i w_i / (z - w_i)          (barycentric)       q_i = (p_i - y) / (w_i - z) .
// The n inversions are one Montgomery-trick run and ONE inversion (modinv.h) per lane over K consecutive elements.  Rounds 1-2
// did this in host Python integers: 4 ms next to a 0.36 ms MSM.  poly: canonical scalars as the MSM takes them; dom, z, inv:
// Montgomery residues; partial / q / y: canonical (a Montgomery product of a canonical and a Montgomery factor is canonical).
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t FR_QUOTIENT_SUM_LANES = 256;   // lanes of the one workgroup that adds up the lane sums
template <class Fr>
struct FrQuotientArgs {
  const uint32_t* poly;   // [n][Fr::N]  p_i, canonical
  const uint32_t* dom;    // [n][Fr::N]  w_i, Montgomery
  Fr z;                   // Montgomery
  Fr scale;               // (z^n - 1) / n, Montgomery
  uint32_t n, K;          // elements; elements per lane of the first pass
  uint32_t* inv;          // [n][Fr::N]  1 / (z - w_i), Montgomery                      (pass 1 -> pass 2)
  uint32_t* partial;      // [ceil(n/K)][Fr::N]  per-lane sums of p_i w_i / (z - w_i), canonical
  uint32_t* tsum;         // [FR_QUOTIENT_SUM_LANES][Fr::N]  strided sums of `partial` (fr_quotient_sum_body)
  uint32_t* q;            // [n][Fr::N]  out: quotient evaluations, canonical
  uint32_t* y;            // [Fr::N]     out: p(z), canonical
};
template <class Fr>
CTT_HD Fr fr_load(const uint32_t* a, uint64_t i) {
  Fr r;
#pragma unroll
  for (int k = 0; k < Fr::N; k++) r.l[k] = a[i * Fr::N + k];
  return r;
}
template <class Fr>
CTT_HD void fr_store(uint32_t* a, uint64_t i, const Fr& v) {
#pragma unroll
  for (int k = 0; k < Fr::N; k++) a[i * Fr::N + k] = v.l[k];
}
// pass 1, lane `lane`: inv_i for its K elements and the lane's share of the barycentric sum
template <class Fr>
CTT_HD void fr_quotient_inv_body(const FrQuotientArgs<Fr>& a, uint32_t lane) {
  const uint64_t i0 = (uint64_t)lane * a.K;
  if (i0 >= a.n) return;
  const uint64_t i1 = i0 + a.K < a.n ? i0 + a.K : a.n;
  Fr run = Fr::one();
  for (uint64_t i = i0; i < i1; i++) {               // prefix products parked in inv[]
    fr_store<Fr>(a.inv, i, run);
    run = Fr::mul(run, Fr::sub(a.z, fr_load<Fr>(a.dom, i)));
  }
  Fr iv = Fr::inv(run);
  Fr sum = Fr::zero();
  for (uint64_t i = i1; i-- > i0;) {
    const Fr w = fr_load<Fr>(a.dom, i);
    const Fr di = Fr::mul(iv, fr_load<Fr>(a.inv, i));  // 1 / (z - w_i)
    iv = Fr::mul(iv, Fr::sub(a.z, w));
    fr_store<Fr>(a.inv, i, di);
    sum = Fr::add(sum, Fr::mul(fr_load<Fr>(a.poly, i), Fr::mul(w, di)));   // canonical p_i times Montgomery w_i / (z - w_i)
  }
  fr_store<Fr>(a.partial, lane, sum);
}
// between the passes, ONE workgroup of T lanes: lane t adds up the lane sums t, t + T, ... into tsum[t] ...
template <class Fr>
CTT_HD void fr_quotient_sum_body(const FrQuotientArgs<Fr>& a, uint32_t t, uint32_t T) {
  const uint32_t lanes = (a.n + a.K - 1) / a.K;
  Fr s = Fr::zero();
  for (uint32_t l = t; l < lanes; l += T) s = Fr::add(s, fr_load<Fr>(a.partial, l));
  fr_store<Fr>(a.tsum, t, s);
}
// ... and (after a workgroup barrier) one lane adds the T sums: y = scale * sum.  (Round 3 had every one of the n lanes of
// pass 2 re-sum all n/K lane sums: O(n^2/K) additions and loads -- nothing at the KZG size 4096, 1.4e11 additions at 2^20.)
template <class Fr>
CTT_HD void fr_quotient_y_body(const FrQuotientArgs<Fr>& a, uint32_t T) {
  Fr s = Fr::zero();
  for (uint32_t t = 0; t < T; t++) s = Fr::add(s, fr_load<Fr>(a.tsum, t));
  fr_store<Fr>(a.y, 0, Fr::mul(s, a.scale));           // canonical
}
// pass 2, element i: q_i = (y - p_i) / (z - w_i)
template <class Fr>
CTT_HD void fr_quotient_out_body(const FrQuotientArgs<Fr>& a, uint32_t i) {
  if (i >= a.n) return;
  const Fr y = fr_load<Fr>(a.y, 0);
  fr_store<Fr>(a.q, i, Fr::mul(Fr::sub(y, fr_load<Fr>(a.poly, i)), fr_load<Fr>(a.inv, i)));
}

// ---------------------------------------------------------------------------------------------
// Probe of the device field FD for the unit tests (GPU: k_field_op_dev, CPU: tests/emu), operands x, y < 2p:
//   0 mul   1 sqr   2 add   3 sub<2>   4 conversion only
//   5, 6, 7  products with operands at the largest bounds -- and in the lazy forms -- that xyzz_madd feeds them
//            (ec.h): 5 = x*y as (x + 10p)(y + 10p) resp. (x + 4p)(y + 10p), 6 = x^2 as (x + 10p)^2 resp. (x + 9p)^2,
//            7 = x*y - x*x as the sum of products R*T + (5p - x)*x of Y3
// ---------------------------------------------------------------------------------------------
template <class FD>
CTT_HD FD dev_field_probe(int op, const FD& x, const FD& y) {
  constexpr bool L1 = LazyOps<FD>::ONE, L2 = LazyOps<FD>::BOTH;
  const FD z = FD::zero();
  switch (op) {
    case 0: return FD::mul(x, y);
    case 1: return FD::sqr(x);
    case 2: return FD::add(x, y);
    case 3: return FD::template sub<2>(x, y);
    case 5: return FD::mul(fsub_lz<FD, (L2 ? 9 : 4), L2>(x, z), fsub_lz<FD, 9, L1>(y, z));
    case 6: return FD::sqr(fsub_lz<FD, 9, L2>(x, z));
    case 7: return fmul_sub_lz<FD, 4, L1>(fsub_lz<FD, 4, L2>(x, z), fsub_lz<FD, 9, L1>(y, z), x, x);
    default: return x;
  }
}

// ---------------------------------------------------------------------------------------------
// Subgroup check of many points at once: ok[j] = (P_j has order r), r = the curve order (the modulus of C::Fr) -- [r]P_j = neutral.
// What the reference's deserialisers do per point (e.g. ethereum_evm_precompiles.nim fromRawCoords -> isInSubgroup); here the
// MSM's callers (EIP-2537 G1MSM/G2MSM, KZG commitments) validate all their points with one launch: the bits of r are the same
// for every lane, so the double-and-add is convergent.
// ---------------------------------------------------------------------------------------------
// BLS12-381 has the reference's endomorphism tests instead (isInSubgroup, named/constants/bls12_381_subgroups.nim:170-207; Scott,
// eprint 2021/1130):  G1: phi(P) = [-x^2]P with phi(X, Y) = (beta X, Y), beta a primitive cube root of unity mod p
// (BLS12_381_cubicRootOfUnity_mod_p, bls12_381_endomorphisms.nim:18-19) -- 126 doublings and 10 additions;  G2: psi(P) = [x]P with the
// untwist-Frobenius-twist psi(X, Y) = (conj(X) c2, conj(Y) c3), c2 = (1/(1+i))^((p-1)/3), c3 = (1/(1+i))^((p-1)/2)
// (= BLS12_381_FrobeniusPsi_psi1_coef2 / _coef3, bls12_381_frobenius.nim:136-145; derived in Python) -- 63 doublings and 5 additions;
// against 255 doublings and ~128 additions.  x = -0xd201000000010000.  Host (Fp64) and device (Fp) run the same templates;
// tests/test_abi_symbols.py and tests/test_batch_ops.py hold them against [r]P = neutral on points inside and outside the subgroups.
struct Bls12381Endo {   // Montgomery residues (R = 2^384), 32-bit words, least significant first
  static constexpr uint32_t BETA[12] = {0x798a64e8u, 0x30f1361bu, 0x7ece5a2au, 0xf3b8ddabu, 0xc61577f7u, 0x16a8ca3au, 0x74fd029bu, 0xc26a2ff8u, 0x60701c6eu, 0x3636b766u, 0x241b6160u, 0x051ba4abu};
  static constexpr uint32_t PSI_C2_1[12] = {0x867545c3u, 0x890dc9e4u, 0x3285a5d5u, 0x2af32253u, 0x309b7e2cu, 0x50880866u, 0x7e881024u, 0xa20d1b8cu, 0xe2db9068u, 0x14e4f04fu, 0x1564853au, 0x14e56d3fu};
  static constexpr uint32_t PSI_C3_0[12] = {0xa55c9ad1u, 0x3e2f585du, 0x86c18183u, 0x4294213du, 0x8b623732u, 0x382844c8u, 0x19103e18u, 0x92ad2afdu, 0xac7cf0b9u, 0x1d794e4fu, 0x7d825ec8u, 0x0bd592fcu};
  static constexpr uint32_t PSI_C3_1[12] = {0x5aa30fdau, 0x7bcfa7a2u, 0x2a927e7cu, 0xdc17dec1u, 0x6b4ebef1u, 0x2f088dd8u, 0xda74d4a7u, 0xd1ca2087u, 0x96cebc1du, 0x2da25966u, 0xbbfd87d2u, 0x0e2b7eedu};
};
// a base-field element from 12 words: Fp (32-bit limbs) or the host's Fp64 (64-bit limbs)
template <class B>
CTT_HD B bls12_381_fp_from_words(const uint32_t* w) {
  B r;
  if constexpr (sizeof(B) / B::N == 8) {
    for (int i = 0; i < B::N; i++) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  } else {
    for (int i = 0; i < B::N; i++) r.l[i] = w[i];
  }
  return r;
}
// [|x|]P, |x| = 0xd201000000010000 (pow_bls12_381_abs_x, bls12_381_subgroups.nim:20-45): 63 doublings and 5 additions
template <class FF>
CTT_HD XYZZ<FF> bls12_381_mul_abs_x(const Affine<FF>& P) {
  XYZZ<FF> r = xyzz_mdbl<FF>(P.x, P.y);
  xyzz_madd<FF>(r, P, false);                                   // 0b11
  const int runs[5] = {2, 3, 9, 32, 16};
  for (int k = 0; k < 5; k++) {
    for (int i = 0; i < runs[k]; i++) r = xyzz_dbl<FF>(r);
    if (k < 4) xyzz_madd<FF>(r, P, false);                      // 0b1101, 0b1101001, ...0000001, ...00000001, then 16 doublings
  }
  return r;
}
// is t (extended Jacobian) the negative of the affine point (qx, qy)?
template <class FF>
CTT_HD bool xyzz_is_neg_of(const XYZZ<FF>& t, const FF& qx, const FF& qy) {
  if (t.is_inf()) return false;
  return FF::eq(t.x, FF::mul(qx, t.zz)) & FF::add(t.y, FF::mul(qy, t.zzz)).is_zero();
}
template <class FF>
CTT_HD bool bls12_381_g1_in_subgroup(const Affine<FF>& P) {
  if (P.is_inf()) return true;
  const Affine<FF> t0 = xyzz_to_affine<FF>(bls12_381_mul_abs_x<FF>(P));   // [|x|]P
  if (t0.is_inf()) return false;
  const XYZZ<FF> t1 = bls12_381_mul_abs_x<FF>(t0);                        // [x^2]P; the test is phi(P) == -t1
  return xyzz_is_neg_of<FF>(t1, FF::mul(P.x, bls12_381_fp_from_words<FF>(Bls12381Endo::BETA)), P.y);
}
template <class FF2>
CTT_HD bool bls12_381_g2_in_subgroup(const Affine<FF2>& P) {
  using B = typename FF2::Base;
  if (P.is_inf()) return true;
  const XYZZ<FF2> t = bls12_381_mul_abs_x<FF2>(P);                        // [|x|]P = -[x]P; the test is psi(P) == -t
  const FF2 c2{B::zero(), bls12_381_fp_from_words<B>(Bls12381Endo::PSI_C2_1)};
  const FF2 c3{bls12_381_fp_from_words<B>(Bls12381Endo::PSI_C3_0), bls12_381_fp_from_words<B>(Bls12381Endo::PSI_C3_1)};
  return xyzz_is_neg_of<FF2>(t, FF2::mul(FF2{P.x.c0, B::neg(P.x.c1)}, c2), FF2::mul(FF2{P.y.c0, B::neg(P.y.c1)}, c3));
}
// one point, any curve: the curve's fast test where there is one, else [r]P = neutral
template <class C, class FF>
CTT_HD bool point_in_subgroup(const Affine<FF>& P) {
  using Fr = typename C::Fr;
  if constexpr (C::ID == 0) {
    return bls12_381_g1_in_subgroup<FF>(P);
  } else if constexpr (C::ID == 1) {
    return bls12_381_g2_in_subgroup<FF>(P);
  } else {
    XYZZ<FF> r = XYZZ<FF>::inf();
    for (int i = 32 * Fr::N - 1; i >= 0; i--) {
      r = xyzz_dbl<FF>(r);
      if ((Fr::Params::P[i >> 5] >> (i & 31)) & 1u) xyzz_madd<FF>(r, P, false);
    }
    return r.is_inf();
  }
}
template <class C>
CTT_HD void subgroup_check_body(const Affine<typename C::F>* pts, uint32_t n, uint8_t* ok, uint32_t j) {
  if (j >= n) return;
  ok[j] = point_in_subgroup<C, typename C::F>(pts[j]) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// Window table for cached bases: next[j] = 2^c * prev[j] (affine in, affine out; one inversion per point -- the table is
// built once per set of bases, MsmEngine::prepare_table).  The neutral (0,0) and points of order two map to (0,0).
// ---------------------------------------------------------------------------------------------
template <class F>
CTT_HD void table_next_body(const Affine<F>* prev, Affine<F>* next, uint32_t n, int c, uint32_t j) {
  if (j >= n) return;
  const Affine<F> p = prev[j];
  if (p.is_inf()) {
    next[j] = Affine<F>::inf();
    return;
  }
  XYZZ<F> r = xyzz_mdbl<F>(p.x, p.y);
  for (int i = 1; i < c; i++) r = xyzz_dbl<F>(r);
  next[j] = xyzz_to_affine<F>(r);
}

// ---------------------------------------------------------------------------------------------
// Synthetic subgroup points for benchmarks/tests: P_i = [s_i]G, s_i = 128-bit splitmix word pair | 1
// (same definition as oracle/pyoracle.py synth_point; mirrors the distribution of the reference's
// bench inputs, benchmarks/bench_elliptic_parallel_template.nim:78-102)
// ---------------------------------------------------------------------------------------------
CTT_HD uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <class F>
CTT_HD void gen_point_body(const Affine<F>& G, uint64_t seed, uint64_t first, uint32_t n, Affine<F>* out, uint32_t j) {
  if (j >= n) return;
  const uint64_t sd = seed ^ 0xA5A5A5A5A5A5A5A5ull;
  uint64_t s[2];
  s[0] = splitmix64(sd + 4 * (first + j) + 0) | 1ull;
  s[1] = splitmix64(sd + 4 * (first + j) + 1);
  XYZZ<F> r = XYZZ<F>::inf();
  for (int i = 127; i >= 0; i--) {
    r = xyzz_dbl<F>(r);
    if ((s[i >> 6] >> (i & 63)) & 1ull) xyzz_madd<F>(r, G, false);
  }
  out[j] = xyzz_to_affine<F>(r);
}

}  // namespace ctt
