// msm_pipeline.h -- host orchestration of one MSM over a Backend (HIP on the GPU; a CPU emulator in
// tests/emu that executes the same per-thread bodies).  See msm_bodies.h for the stage list.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <functional>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

#include "host_fp64.h"
#include "msm_bodies.h"

namespace ctt {

// thrown by a backend's alloc() when the device is out of memory: the engine releases what it holds for the call and the
// C ABI turns it into the call's error value (NULL / -1) where there is one (round 2 aborted the process)
struct OutOfDeviceMemory { size_t bytes; };

struct MsmPlan {
  uint32_t n;
  int c, W;
  uint32_t B;      // buckets per window = 2^(c-1)
  uint32_t S;      // partition blocks (sort pass A): each handles `slice` consecutive scalars
  uint32_t slice;  // scalars per partition block
  uint32_t NG;     // bucket groups per window (sort pass A partitions by group, pass B sorts inside a group)
  uint32_t gshift; // group = bucket >> gshift
  uint32_t gshift_narrow;  // windows one bit narrower than c only reach B/2 buckets: their groups are half as wide
  WinLayout lay;       // widths / offsets of the W (Wd) digit windows; c = lay.cmax()
  uint32_t jbits;  // bits of a point index (sort records pack low bucket bits | sign | index into 32 bits)
  uint32_t cap, big;  // sort pass B: LDS tile entries; bucket size above which the LDS image is bypassed
  uint32_t K;      // sorted entries per accumulate lane
  uint32_t G;      // accumulate lanes per window
  // window-table form (make_table_plan): Wd digit windows share ONE bucket set (W = 1) of nent = Wd*n entries
  int Wd;              // digit windows per scalar (== W otherwise)
  uint32_t merged;     // 1 = window-table form
  uint32_t nent;       // entries per bucket set: n, or Wd*n
  uint32_t id_stride;  // table rows per window (the cached bases' length; a call may use a prefix)
  int h, ngrp;         // bit Horner: bits per group, groups per window (the device returns W*ngrp partial sums)
  int merge_steps;     // wide head-merge tree steps enqueued without knowing the largest bucket (plan_merge_steps)
  uint32_t merge_lmax; // > 0: the queue form of the head merge (msm_bodies.h merge_tail_queue_body) for chains of at most this many heads; 0: the tree
};

struct MsmOptions {
  int c = 0;          // window bits (0 = choose)
  int K = 0;          // entries per lane (0 = choose from resident lanes)
  int S = 0;          // sort: scalars per partition block (0 = choose)
  uint32_t lanes = 196608;  // resident lanes of the accumulate kernel (set by the backend)
  int host_window_sums = 0;  // legacy spelling of horner_bits: 1 = 1 bit per group (the whole bit Horner on the host),
                             // 2 = one group per window (the whole bit Horner on the device), 0 = horner_bits decides
  int horner_bits = 0;       // bits per group of the bit Horner the device runs (0 = choose: 4); see plan_horner
  // cost model of the window choice: ns per mixed addition (accumulate) / per full addition (reduction) with the chip busy;
  // the engine fills in its curve's figures (msm_bodies.h curve descriptors), the defaults are BLS12-381 G1's
  double acc_ns = 0.142, red_ns = 0.26;
  // sort pass A (msm_engine.hip).  sort_xcd: neighbouring slices on one XCD (1, default) or slice b to block b (0).  sort_staged:
  // records staged through an LDS image of the block's output -- 1 = where it pays (default: from 256 bucket groups per window,
  // i.e. ~2^22 pairs, on; measured, profiles/sort_staged_xcd_r04.txt: the sort of 2^22 / 2^24 BLS12-381 pairs 0.545 -> 0.486 /
  // 2.26 -> 2.06 ms, BN254 2^22 0.526 -> 0.464, but 0.148 -> 0.173 ms at 2^20 and 0.056 -> 0.081 at 2^16: there a block has one or
  // two window steps and the five barriers per step are what it sees), 2 = always, 0 = never (one store per record: the form
  // that also serves more than 1024 groups).  Options "sort_xcd" / "sort_staged", $CTT_SORT_XCD / $CTT_SORT_STAGED.
  int sort_xcd = 1, sort_staged = 1;
  int early_tail = 1;   // MsmEngine::submit: merge + every reduction pass on the tail stream for large pipelined MSMs (0 off, 1 automatic, 2 always)
  // head merge: 0 = the queue form (tail merge + one lane per chain with work left, msm_bodies.h merge_tail_queue_body) when the plan
  // expects chains of at most merge_lmax heads, else the tree; 1 = the queue form always; 2 = the tree always.  merge_lmax 0 = 8.
  int merge_chain = 0, merge_lmax = 0;
  int merge_queue_quad = 0;
  int front_side = 0;         // small pipelined MSMs: conversion + sort on the front stream (0 automatic, 1 always when pipelining, 2 never)   // the queue kernel with four lanes per chain (hip_backend.h k_merge_queue_quad): 0 / 1 = on, 2 = one lane per chain
  // experiment knob (round 5, measured and NOT adopted): 1 = small pipelined MSMs (up to 2^17 pairs) put the FIRST reduction pass on the tail
  // stream too, so that the next MSM's sort starts right behind the head merge.  Same box, ms per MSM with two in flight, off / on:
  // BLS12-381 G1 2^16 0.466-0.473 / 0.480, 2^17 0.654-0.657 / 0.676-0.680, BN254 2^16 0.345 / 0.350 -- the fork's event pair costs what the
  // 25 us of overlap give (gpurun_out/r5i)
  int pyr0_tail = 0;
};

// Window size for the GPU pipeline.  The reference's bestBucketBitSize
// (ec_multi_scalar_mul_scheduler.nim:172-223) models a CPU; any c yields the same group element, so the device
// uses its own cost model with constants measured on MI355X for BLS12-381 (profiles/): they only rank the
// candidates, so the same model serves the other curves.
//   accumulate  W*N mixed adds at 0.142 ns each (2.38 ms / 2^24 at full occupancy)
//   reduce      c-1 passes of 12 us latency each, plus 2*2^(c-1)*W adds of work at 0.24 ns (fitted to 0.43 ms at c = 16,
//               0.185 ms at c = 13)
//   merge       45 us + one 28 us tree step per doubling of the longest head chain.  The top window only has
//               bits - (W-1)*c significant bits: when that is small its few buckets each receive N/2^(top-1)
//               entries and the chain is long -- the model steers away from such c (e.g. c = 14 at N = 2^18)
//   sort        0.02 ns per (window, pair) + 60 us
// Round-2 check against measurements (BLS12-381 G1, ms per pipelined step): 2^16 c = 13 0.709 / c = 16 0.756; 2^18 c = 16
// 1.22 / c = 13 1.59; the round-1 constants still chose c = 13 at 2^17, where c = 16 is the faster plan.
// Scalars per partition block (sort pass A).  Large n: 512 blocks, two per CU, ALL of the same size -- a power-of-two slice
// made 257 blocks of 16384 scalars out of n = 2^22 + 77777, and the one CU that got two of them doubled the time of both
// partition kernels (measured: sort 1.00 ms instead of 0.65 ms).  Small n: at least ~64 blocks.
// From 2^23 pairs on: 2048 blocks.  A block walks its slice in steps of 4096 scalars and, per step, all windows; with 1024 group
// regions per window a step leaves 16 bytes in each run and the line is written back partially before the block returns to it
// (k_part_scatter writes 5.0x its 4 bytes per record at 2^24, 1.55x at 2^22: profiles/pmc_r03_hbm_bytes_*).  Shorter slices
// put the neighbouring pieces of a line into blocks that run at the same time: measured at 2^24, sort 3.40 ms with 32768
// scalars per block, 3.16 with 16384, 2.85 with 8192, 3.06 with 4096 (whose count table is 250 MB); no gain at 2^22.
static inline uint32_t plan_partition_slice(uint32_t n, uint32_t min_slice) {
  const uint32_t nblk = n >= (1u << 23) ? 2048u : 512u;
  uint32_t slice = (uint32_t)(((uint64_t)n + nblk - 1u) / nblk);
  slice = (slice + 255u) & ~255u;
  if (slice < min_slice) slice = min_slice;
  while (slice > 64u && (uint64_t)slice * 64u > n) slice >>= 1;
  return slice;
}

static inline uint32_t plan_entries_per_lane(uint32_t n, int W, uint32_t lanes) {
  uint64_t total = (uint64_t)W * n;
  uint32_t K = (uint32_t)((total + lanes - 1) / lanes);
  if (K < 4) K = 4;
  // The accumulate kernel is launched as W rows of ceil(ceil(n/K)/64) one-wave workgroups, and all of them must be resident
  // at once: with even one workgroup more than wave slots, a second round runs that single wave for a whole K entries
  // (measured, BN254 2^22: c = 15 -> 17 x 241 = 4097 workgroups on 4096 slots, accumulate 6.1 ms instead of ~4.9 ms).
  // The rounding of the rows can exceed the slots for any n that is not a power of two: grow K until the grid fits.
  const uint64_t slots = lanes / 64u;
  while ((uint64_t)W * ((((uint64_t)n + K - 1) / K + 63u) / 64u) > slots && K < 0x7ffffff0u) K += 1u;
  return K;
}

// (c is the width asked for; the plan's windows are balanced: window_layout(), msm_bodies.h)
// Round-3 re-fit with balanced windows and per-curve constants, against one box's sweeps (gpurun_out/r3d -> profiles/
// sweep_window_bits_r03.jsonl; ms per MSM with two in flight): the model's choice is the measured best or within 3 % of it
// for BLS12-381 G1 2^12 .. 2^22 (13, 13, 13, 13/14, 14, 16, 16, 16), G2 2^16 .. 2^20, BN254 2^16 .. 2^22, Pallas 2^16 / 2^20.
// A latency term is weighted by the curve's addition time (a pass of the reduction is one addition deep).
static inline int choose_window_bits(uint32_t n, int bits, uint32_t lanes, double acc_ns = 0.142, double red_ns = 0.26) {
  double best = 1e300;
  int bc = 8;
  const double ratio = acc_ns / 0.142;
  // (c = 17, 18 only pay from ~2^23 pairs on -- measured BLS12-381 G1 2^24: 15 windows of 17-18 bits 41.3 ms per MSM against
  // 43.6 ms for 16 windows of 16, 2^22: 11.5 against 10.9 -- and the sort's 32-bit records hold c <= 44 - bits(n), see make_plan)
  for (int c = 6; c <= 18; c++) {
    int W;
    const WinLayout L = window_layout(bits, c, &W);
    const int cm = L.cmax();
    const double B = (double)(1u << (cm - 1));
    const double K = (double)plan_entries_per_lane(n, W, lanes);
    const double acc = (double)W * n * acc_ns * 1e-3;
    const double red = (cm - 1) * 8.0 * ratio + 2.0 * B * W * red_ns * 1e-3;
    double maxcnt = 2.0 * n / (double)(1u << (L.cb - 1));   // the narrower windows fill 2^(cb-1) buckets; twice the mean
    double chain = maxcnt / K;
    int steps = 0;
    while (chain > 1.0) { chain *= 0.5; steps++; }
    const double mer = (45.0 + 28.0 * steps) * ratio;
    const double srt = (double)W * n * 0.02e-3 + 60.0;
    const double cost = acc + red + mer + srt;
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

// Groups of the bit Horner (hip_backend.h k_window_groups, window_group_sum_body): h bits per group, the device returns
// ngrp = ceil((c-1)/h) partial sums per window.
static inline void plan_horner(MsmPlan& p, const MsmOptions& o) {
  int h = o.host_window_sums == 1 ? 1 : o.host_window_sums == 2 ? p.c : o.horner_bits > 0 ? o.horner_bits : 4;
  if (h > p.c) h = p.c;
  p.h = h;
  p.ngrp = horner_groups(p.c, h);
}
// Wide head-merge tree steps to enqueue: enough for the largest bucket an ordinary (uniform) digit distribution produces --
// the mean of the fullest buckets (those of the narrower windows) with a margin of 6 sigma + 8; whatever an unusual input
// needs on top of that is done by the merge-finish launch (one workgroup per window, msm_bodies.h merge_finish_body).
static inline int plan_merge_steps(const MsmPlan& p, int bits) {
  (void)bits;
  // entries per bucket: a window of width cw spreads its n digits over 2^(cw-1) buckets; the window table adds up all windows
  const double narrow = (double)p.n / (double)(1u << (p.lay.cb - 1));
  double m = narrow;
  if (p.merged) {
    const int wide = p.lay.r, nar = p.Wd - p.lay.r;   // (r = 0: every window has cb bits and counts as narrow here)
    m = (double)p.n * ((double)wide + 2.0 * (double)nar) / (double)p.B;
    if (p.lay.r == 0) m = (double)p.n * (double)p.Wd / (double)p.B;
  }
  double sd = 1.0;
  while (sd * sd < m) sd += 1.0;
  m += 6.0 * sd + 8.0;
  double chain = (m - 1.0) / (double)p.K + 1.0;
  int steps = 0;
  while (chain > 1.0 && steps < 31) { chain *= 0.5; steps++; }
  return steps;
}
// Queue form or tree (MsmPlan::merge_lmax)?  What is left of a chain behind its tail merge is walked by ONE lane, so the form pays while the chains an ordinary digit
// distribution produces are short: the same bound as plan_merge_steps (2^steps >= heads of the fullest ordinary bucket) against
// lmax.  Plans with tiny K against full buckets (a 4096-point commitment over a window table: K = 4, ~90 entries per bucket, chains
// of 20 heads) keep the tree; so does sum_reduce, whose single bucket spans every lane.
static inline uint32_t plan_merge_lmax(const MsmPlan& p, const MsmOptions& o) {
  const uint32_t lmax = o.merge_lmax > 0 ? (uint32_t)o.merge_lmax : 8u;
  if (o.merge_chain == 1) return lmax;
  if (o.merge_chain == 2) return 0u;
  // not over the quadratic extensions (acc_ns is the engine's curve constant: 0.47-0.5 for the G2 curves): a full addition there is 3.3 x a G1
  // one, and the queue kernel's one-lane additions cost more than the tree's four-lane steps -- same box, BLS12-381 G2 2^18, ms per MSM with
  // two in flight, tree / queue: 2.96 / 3.02, and 3.22-3.29 / 3.32-3.36 in the A/B against the round-4 library (profiles/ab_prev_vs_r05_first.txt)
  if (o.acc_ns >= 0.3) return 0u;
  return (p.merge_steps <= 30 && (1u << p.merge_steps) <= lmax) ? lmax : 0u;
}

static inline MsmPlan make_plan(uint32_t n, int bits, const MsmOptions& o) {
  MsmPlan p;
  p.n = n;
  p.c = o.c > 0 ? o.c : choose_window_bits(n, bits, o.lanes, o.acc_ns, o.red_ns);
  if (p.c < 2) p.c = 2;
  if (p.c > 20) p.c = 20;   // (the automatic choice stays <= 18; wider windows on request: 2^19 buckets per window at most)
  {
    // the sort packs (low bucket bits | sign | point index) into 32 bits with at most 4096 bucket groups per window:
    // beyond 2^28 pairs that caps the window width (c <= 44 - bits(n): 15 at 2^29, 13 at 2^31)
    uint32_t jb = 1;
    while (jb < 31 && (1ull << jb) < n) jb++;
    while (p.c > 2 && (int)jb + 1 + (p.c - 1 - 12) > 32) p.c--;
  }
  // balanced windows over bits + 1 bits (msm_bodies.h WinLayout); the reference's count, bits/c + 1 windows when c | bits
  // (ec_multi_scalar_mul_parallel.nim:157-158), comes out of the same formula
  p.lay = window_layout(bits, p.c, &p.W);
  p.c = p.lay.cmax();
  p.B = 1u << (p.c - 1);
  // sort pass A: ~512 partition blocks of at least 2048 scalars; pass B: groups of ~16384 entries (one workgroup
  // sorts a group inside LDS), at most 4096 buckets per group (LDS counters)
  static const uint32_t slenv = getenv("CTT_SORT_SLICE") ? (uint32_t)atoi(getenv("CTT_SORT_SLICE")) : 2048u;
  uint32_t slice = o.S > 0 ? (uint32_t)o.S : plan_partition_slice(n, slenv);
  p.slice = slice;
  p.S = (n + slice - 1) / slice;
  p.jbits = 1;
  while (p.jbits < 31 && (1ull << p.jbits) < n) p.jbits++;
  // groups of ~16384 entries (k_group_sort holds one in LDS), at most 1024 buckets per group, and the packed
  // record (low bucket bits | sign | index) must fit 32 bits
  // (18432, not 16384: a size just above a power of two keeps the group count of that power of two -- 2^22 + 77777 pairs with
  // 512 groups of 8192 instead of 256 of 16384 sorted in 0.82 ms instead of 0.65 ms; a group may hold 20480 in one sweep)
  static const uint32_t gsz = getenv("CTT_SORT_GROUP") ? (uint32_t)atoi(getenv("CTT_SORT_GROUP")) : 18432u;
  static const uint32_t capenv = getenv("CTT_SORT_CAP") ? (uint32_t)atoi(getenv("CTT_SORT_CAP")) : 20480u;
  static const uint32_t bigenv = getenv("CTT_SORT_BIG") ? (uint32_t)atoi(getenv("CTT_SORT_BIG")) : 1024u;
  p.cap = capenv;
  p.big = bigenv;
  uint32_t NG = 1;
  while ((uint64_t)NG * gsz < n && NG < 4096u) NG <<= 1;  // beyond 2^26 pairs the groups grow instead (tiled in pass B)
  while (NG < p.B && p.B / NG > 1024u) NG <<= 1;
  if (NG > p.B) NG = p.B;
  p.gshift = 0;
  while ((p.B >> p.gshift) > NG) p.gshift++;
  while (p.gshift > 0 && p.jbits + 1 + p.gshift > 32) { p.gshift--; NG <<= 1; }
  p.NG = NG;
  p.gshift_narrow = (p.lay.r > 0 && p.gshift > 0) ? p.gshift - 1 : p.gshift;
  // entries per lane: fill the resident lanes once
  uint32_t K = o.K > 0 ? (uint32_t)o.K : plan_entries_per_lane(n, p.W, o.lanes);
  if (K < 4) K = 4;
  p.K = K;
  p.G = (n + K - 1) / K;
  p.Wd = p.W;
  p.merged = 0;
  p.nent = n;
  p.id_stride = 0;
  plan_horner(p, o);
  p.merge_steps = plan_merge_steps(p, bits);
  p.merge_lmax = plan_merge_lmax(p, o);
  return p;
}

// Window bits of a window table over `ntab` bases (MsmEngine::prepare_table), chosen when the table is built: the table
// fixes c for every later call.  One bucket set serves all windows, so the reduction costs 2*2^(c-1) additions once
// instead of once per window and c can grow until those balance the (bits/c + 1)*N accumulations (an entry is a table row |
// sign << 31: at most 2^31 - 1 rows).
static inline int table_index_bits(uint64_t rows) {
  int jb = 1;
  while (jb < 31 && (1ull << jb) < rows) jb++;
  return jb;
}
static inline bool table_plan_fits(uint32_t ntab, int bits, int c) {
  int Wd;
  (void)window_layout(bits, c, &Wd);
  const uint64_t rows = (uint64_t)Wd * ntab;   // an entry is a table row | sign << 31
  return rows <= 0x7fffffffull && Wd <= 128;
}
static inline int choose_table_window_bits(uint32_t ntab, int bits) {
  // Same constants as choose_window_bits, one bucket set: 2*2^(c-1) additions of reduction in total, not per window, so c
  // grows until those balance the Wd*N accumulations.  With balanced windows (round 3) the windows one bit narrower than
  // c fill only the lower half of the shared buckets -- twice the mean there, nothing worse: round 2's layout put all N
  // digits of a narrow top window into 2^top buckets and restricted the table to the few c with a wide remainder
  // (measured then, BLS12-381 G1 2^20: c = 20 2.68 ms per MSM, c = 19 3.17 ms, c = 21 4.42 ms).
  double best = 1e300;
  int bc = 0;
  for (int c = 4; c <= 22; c++) {
    if (!table_plan_fits(ntab, bits, c)) continue;
    int Wd;
    const WinLayout L = window_layout(bits, c, &Wd);
    const int cm = L.cmax();
    const double B = (double)(1u << (cm - 1));
    const double total = (double)Wd * ntab;
    const double K = (double)plan_entries_per_lane((uint32_t)(total > 4e9 ? 4e9 : total), 1, 131072);
    const double acc = total * 0.142e-3;
    const double red = (cm - 1) * 12.0 + 2.0 * B * 0.24e-3;
    const double maxcnt = 2.0 * (L.r ? (double)ntab * (L.r + 2.0 * (Wd - L.r)) / B : total / B);
    double chain = maxcnt / K;
    int steps = 0;
    while (chain > 1.0) { chain *= 0.5; steps++; }
    const double mer = 45.0 + 28.0 * steps;
    const double srt = total * 0.02e-3 + 60.0;
    const double cost = acc + red + mer + srt;
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

// Plan of one MSM over the first n bases of a window table built with c window bits over ntab bases.
static inline MsmPlan make_table_plan(uint32_t n, int bits, int c, uint32_t ntab, const MsmOptions& o) {
  MsmPlan p;
  p.n = n;
  p.lay = window_layout(bits, c, &p.Wd);
  p.c = p.lay.cmax();
  p.W = 1;
  p.B = 1u << (p.c - 1);
  p.merged = 1;
  p.nent = (uint32_t)((uint64_t)p.Wd * n);
  p.id_stride = ntab;
  uint32_t slice = o.S > 0 ? (uint32_t)o.S : plan_partition_slice(n, 2048u);
  p.slice = slice;
  p.S = (n + slice - 1) / slice;
  p.jbits = 0;   // (the 64-bit partition records of this form hold the whole table row)
  p.cap = 20480u;
  p.big = 1024u;
  // bucket groups of ~12288 records over all windows (the groups of the top window's buckets receive its records on top, and
  // a group of up to 20480 is sorted in one sweep), at most 1024 buckets per group, and the packed record must fit 32 bits
  // (the narrower windows only reach the lower half of the buckets: the groups there hold `heavy` records between them)
  const uint64_t heavy = p.lay.r ? 2ull * ((uint64_t)p.nent - (uint64_t)p.lay.r * n / 2u) : (uint64_t)p.nent;
  uint32_t NG = 1;
  while ((uint64_t)NG * 12288u < heavy && NG < 4096u) NG <<= 1;
  while (NG < p.B && p.B / NG > 1024u) NG <<= 1;
  if (NG > p.B) NG = p.B;
  p.gshift = 0;
  while ((p.B >> p.gshift) > NG) p.gshift++;
  p.NG = NG;
  p.gshift_narrow = p.gshift;  // all windows share the groups
  uint32_t K = o.K > 0 ? (uint32_t)o.K : plan_entries_per_lane(p.nent, 1, o.lanes);
  if (K < 4) K = 4;
  p.K = K;
  p.G = (p.nent + K - 1) / K;
  plan_horner(p, o);
  p.merge_steps = plan_merge_steps(p, bits);
  p.merge_lmax = plan_merge_lmax(p, o);
  return p;
}

// Window combine (ec_multi_scalar_mul.nim:250-254; _parallel.nim:199-203), in two parts.
// On the device, fused into the second launch of the bucket reduction: per window the reduction leaves O_0..O_{c-2} (sum of
// the buckets whose index has bit l set) and TOP (sum of all buckets); the window sum is S_w = sum_l 2^l O_l + TOP.  The
// bits are cut into groups of h: P_{w,g} = sum_{l in group g} 2^(l - g*h) O_l (+ TOP in group 0), one quad of lanes per group
// (window_group_sum_body is what a quad computes) -- a chain of h-1 doublings instead of c-2.  On the host: the Horner over
// the windows, result = sum_w 2^(c*w) sum_g 2^(g*h) P_{w,g}: its W*c doublings are one dependent chain whatever h is (0.25 us
// per doubling on a CPU core against 5 us for four GPU lanes), the groups only add ngrp - 1 additions per window to it.
template <class F>
CTT_HD XYZZ<F> window_group_sum_body(const XYZZ<F>* ow, int c, int h, int g) {
  const int lo = g * h;
  int hi = lo + h;
  if (hi > c - 1) hi = c - 1;
  XYZZ<F> r = XYZZ<F>::inf();
  for (int l = hi - 1; l >= lo; l--) {
    r = xyzz_dbl<F>(r);
    xyzz_add<F>(r, ow[l]);
  }
  if (g == 0) xyzz_add<F>(r, ow[c - 1]);
  return r;
}
template <class F>
static inline XYZZ<F> combine_groups(const XYZZ<F>* s, int W, const WinLayout& L, int h, int ngrp) {
  XYZZ<F> r = XYZZ<F>::inf();
  for (int w = W - 1; w >= 0; w--) {
    // window w+1 starts width(w) bits above window w: those doublings are spread over the groups of window w
    const int cw = L.width((uint32_t)w);
    for (int g = ngrp - 1; g >= 0; g--) {
      const int nd = g == ngrp - 1 ? cw - g * h : h;
      for (int l = 0; l < nd; l++) r = xyzz_dbl<F>(r);
      xyzz_add<F>(r, s[(size_t)w * ngrp + g]);
    }
  }
  return r;
}

// Stage indices for timings
enum { ST_DIGITS = 0, ST_SORT, ST_ACCUM, ST_MERGE, ST_REDUCE, ST_TOTAL, ST_COUNT };

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

template <class C, class BK>
struct MsmEngine {
  using F = typename C::F;    // reference representation (C API)
  using FD = typename C::FD;  // device representation
  static constexpr bool kConvert = !std::is_same<F, FD>::value;
  BK& bk;
  MsmOptions opt;
  MsmPlan last_plan;

  // grow-only workspace
  struct Buf { void* p = nullptr; size_t cap = 0; };
  // In-flight slots.  Rounds 1-5 had two; round 6 read the timeline of a pipelined 2^16-pair loop (profiles/cu_mask_r06.txt): with two
  // slots, submit(i+2) has to follow finish(i) -- the wait for MSM i's result copy, then its host tail (~0.1-0.2 ms: W c doublings on one
  // core), then the enqueueing of ~24 launches -- and the main stream idles ~90 us of every 450 us step because MSM i+2's first kernel
  // has not been enqueued yet when MSM i+1's main-stream chain ends.  With a third slot a caller submits MSM i+2 BEFORE it finishes MSM i.
  static constexpr int NSLOT = 3;
  // bstartS / maxcountS / bucketsS: one per in-flight slot -- the head merge and the first reduction pass of MSM i read them on the tail
  // stream while the sort of MSM i+1 already writes its own (submit(): early tail)
  Buf part, counts, bstartS[NSLOT], entries, bucketsS[NSLOT], heads, tails, hkey, tkey, rA[2], rP[2], scal, maxcountS[NSLOT], cpoints, totals, gbase, mqueue;

  explicit MsmEngine(BK& b) : bk(b) {}
  ~MsmEngine() {
    Buf* all[] = {&part, &gbase, &counts, &entries, &heads, &tails, &hkey, &tkey, &rA[0], &rA[1], &rP[0], &rP[1], &scal, &cpoints, &totals, &mqueue};
    for (int i = 0; i < NSLOT; i++) {
      bk.free_quiet(bstartS[i].p);
      bk.free_quiet(bucketsS[i].p);
      bk.free_quiet(maxcountS[i].p);
    }
    // (a destructor is noexcept: the quiet forms -- ctt_hip_msm_ctx_destroy of a context lost to a HIP failure comes through here)
    for (Buf* b : all) if (b->p) bk.free_quiet(b->p);
    for (Slot& sl : slots) if (sl.hraw) bk.free_host_quiet(sl.hraw);
  }
  void* need(Buf& b, size_t bytes) {
    if (bytes > b.cap) {
      void* old = b.p;
      b.p = nullptr;      // (before the free: if it throws, the destructor must not free the pointer a second time)
      b.cap = 0;
      if (old) bk.free(old);
      size_t cap = bytes + bytes / 8 + 256;
      b.p = bk.alloc(cap);   // may throw OutOfDeviceMemory: the buffer is then simply empty
      b.cap = cap;
    }
    return b.p;
  }

  // Up to NSLOT MSMs may be in flight: submit() enqueues every kernel of one MSM plus the asynchronous copy of its
  // c points per window into a pinned host buffer; finish() waits for that copy and runs the host tail.
  // Calling submit(i+1) before finish(i) overlaps the host tail of MSM i with the GPU work of MSM i+1
  // (the workspace is shared: stream order keeps the two apart on the device).
  using HF = typename HostField<FD>::type;
  struct Slot {
    MsmPlan plan;
    bool busy = false;
    bool empty = false;   // len == 0
    void* hraw = nullptr; // pinned host buffer for the device output
    size_t hcap = 0;
  };
  Slot slots[NSLOT];
  int next_slot = 0;
  int busy_count() const {
    int k = 0;
    for (const Slot& s : slots) k += s.busy ? 1 : 0;
    return k;
  }
  // another MSM is in flight beside slot sl's: the caller is pipelining
  bool other_busy(int sl) const {
    for (int i = 0; i < NSLOT; i++) if (i != sl && slots[i].busy) return true;
    return false;
  }

  // d_coefs: canonical scalars [n][8] (coef_is_fr = false) or Montgomery Fr elements (true), device memory.
  // d_points: affine Montgomery points (reference representation), device memory.  Returns the slot.
  // Converted point records for base points that are reused across MSMs (the ZAL "base descriptor",
  // constantine-halo2-zal/src/lib.rs:68-71): returns a device buffer owned by the caller (free with bk.free).
  void* prepare_bases(const Affine<F>* d_points_in, uint32_t n) {
    if (n == 0) return nullptr;
    if constexpr (kConvert) {
      void* cp = bk.alloc((size_t)n * gather_stride<FD>());
      bk.template launch_convert<F, FD>(d_points_in, cp, n);
      return cp;
    } else {
      void* cp = bk.alloc((size_t)n * sizeof(Affine<F>));
      bk.d2d_async(cp, d_points_in, (size_t)n * sizeof(Affine<F>));
      return cp;
    }
  }

  // Window table over bases that are reused across MSMs (an SRS, a commitment key): T[w][j] = 2^(c*w) * P_j for the
  // Wd = bits/c + 1 digit windows, as converted point records.  With it every Booth digit of every window is a signed
  // multiple of a TABLE row, all windows share one bucket set, and the doublings of the window combine disappear: an MSM
  // is (bits/c + 1)*N accumulations and ONE bucket reduction over 2^(c-1) buckets -- so c is larger than without the
  // table (20 instead of 16 at N = 2^20: 13 instead of 16 accumulations per pair).  The price is memory, Wd times the
  // point records (1.7 GB for 2^20 BLS12-381 G1 bases: this is what 288 GB of HBM is for), and c*(Wd-1) doublings plus
  // Wd-1 inversions per base when the table is built.  The reference has no such table; its nearest relative is the ZAL
  // base descriptor (constantine-halo2-zal/src/lib.rs:68-95), which upstream passes through unchanged.
  // Returns the records (caller-owned, free with bk.free); c = 0 chooses the window bits.  *c_out = the bits used.
  void* prepare_table(const Affine<F>* d_points_in, uint32_t n, int c, int* c_out) {
    // c above what choose_table_window_bits searches (22: 2^21 buckets, 16384 bucket groups of the sort) is not a plan the
    // sort can run, c < 2 is no window at all: take the automatic choice instead
    if (c < 2 || c > 22 || !table_plan_fits(n, C::BITS, c)) c = choose_table_window_bits(n, C::BITS);
    *c_out = c;
    if (n == 0 || c == 0) return nullptr;
    int Wd;
    const WinLayout lay = window_layout(C::BITS, c, &Wd);
    const size_t stride = kConvert ? (size_t)gather_stride<FD>() : sizeof(Affine<F>);
    char* tab = nullptr;
    Affine<F>* lvl[2] = {nullptr, nullptr};
    try {
      tab = (char*)bk.alloc((size_t)Wd * n * stride);
      lvl[0] = (Affine<F>*)bk.alloc((size_t)n * sizeof(Affine<F>));
      lvl[1] = (Affine<F>*)bk.alloc((size_t)n * sizeof(Affine<F>));
    } catch (const OutOfDeviceMemory&) {
      // (bits/c + 1) x the records did not fit: no table, the caller keeps plain records (c_out = 0)
      if (tab) bk.free(tab);
      if (lvl[0]) bk.free(lvl[0]);
      *c_out = 0;
      return nullptr;
    }
    const Affine<F>* cur = d_points_in;
    for (int w = 0; w < Wd; w++) {
      if (w > 0) {
        Affine<F>* nxt = lvl[w & 1];
        bk.template launch_table_next<F>(cur, nxt, n, lay.width((uint32_t)w - 1));   // level w = 2^width(w-1) x level w-1
        cur = nxt;
      }
      if constexpr (kConvert) bk.template launch_convert<F, FD>(cur, tab + (size_t)w * n * stride, n);
      else bk.d2d_async(tab + (size_t)w * n * stride, cur, (size_t)n * sizeof(Affine<F>));
    }
    bk.sync();
    bk.free(lvl[0]);
    bk.free(lvl[1]);
    return tab;
  }

  // ---- the stages of one MSM ---------------------------------------------------------------------------------------
  // Stage 1 (per MSM, or per chunk of a host-pointer MSM): scalars -> Booth digits -> entries sorted by bucket ->
  // bucket sums of these pairs in `d_buckets` (heads/tails of the runs that straddle lane ranges still to be merged).
  // Ends with the accumulate kernel enqueued.
  struct Staged {
    uint32_t* d_bstart;
    uint32_t* d_maxcount;
    XYZZ<FD>* d_buckets;
    XYZZ<FD>* d_heads;
    XYZZ<FD>* d_tails;
    uint32_t* d_hkey;
    uint32_t* d_tkey;
  };
  // the grow-only workspace of stage 1 sized for plan p up front (need() frees and reallocates -- a device-wide
  // synchronisation -- when a later, larger slice of a host-pointer MSM asks for more)
  void reserve_stage1(int sl, const MsmPlan& p, bool coef_is_fr) {
    const size_t W = p.W, n = p.nent;
    if (coef_is_fr) need(scal, (size_t)p.n * 32);
    need(part, W * n * (p.merged ? 8 : 4));
    need(counts, (size_t)p.S * W * p.NG * 4);
    need(totals, W * p.NG * 4);
    need(gbase, W * (p.NG + 1) * 4);
    need(bstartS[sl], W * (p.B + 1) * 4);
    need(entries, W * n * 4);
    need(maxcountS[sl], 256);
    need(heads, W * p.G * sizeof(XYZZ<FD>));
    need(tails, W * p.G * sizeof(XYZZ<FD>));
    need(hkey, W * p.G * 4);
    need(tkey, W * p.G * 4);
    if (p.merge_lmax > 0) need(mqueue, (size_t)merge_queue_capacity(p.W, p.G) * 4);
  }
  // points_arrive (submit_host, a slice copied by the submitting thread): the points of this slice are not on the device yet -- the
  // digits and the sort need the coefficients only, so they are enqueued first, the hook then copies the points (the thread sits in that
  // copy while the GPU sorts) and the conversion follows the sort instead of preceding it.
  // front_side (submit(), small MSMs kept in flight): conversion, digits and sort on the backend's front stream, beside the previous MSM's
  // head merge and first reduction pass (HipBackend::front_begin has the ordering argument)
  Staged accumulate_pairs(int sl, const MsmPlan& p, const uint32_t* d_coefs, bool coef_is_fr, const Affine<F>* d_points_in,
                          const void* d_prepared, void* d_converted, XYZZ<FD>* d_buckets, bool into = false,
                          const std::function<void()>* points_arrive = nullptr, bool front_side = false) {
    const uint32_t n = p.n, W = p.W, B = p.B;
    if (front_side) bk.front_begin();
    bk.stage_begin(sl, ST_DIGITS);
    const uint32_t* d_scalars = d_coefs;
    if (coef_is_fr) {
      uint32_t* t = (uint32_t*)need(scal, (size_t)n * 32);
      bk.template launch_fr_from_mont<typename C::Fr>(d_coefs, t, n);
      d_scalars = t;
    }
    const void* d_points;
    uint32_t point_stride;
    const bool convert_late = points_arrive != nullptr;
    if constexpr (kConvert) {
      if (d_prepared) {
        d_points = d_prepared;
      } else {
        if (!convert_late) bk.template launch_convert<F, FD>(d_points_in, d_converted, n);
        d_points = d_converted;
      }
      point_stride = gather_stride<FD>();
    } else {
      d_points = d_prepared ? d_prepared : (const void*)d_points_in;
      point_stride = (uint32_t)sizeof(Affine<F>);
    }
    bk.stage_end(sl, ST_DIGITS);

    // Booth digits + sort by bucket (two passes: partition by bucket group, then sort each group inside LDS)
    bk.stage_begin(sl, ST_SORT);
    SortArgs sa;
    sa.scalars = d_scalars;
    sa.n = n; sa.c = p.c; sa.lay = p.lay; sa.W = W; sa.B = B;
    sa.Wd = (uint32_t)p.Wd; sa.merged = p.merged; sa.nent = p.nent; sa.id_stride = p.id_stride;
    sa.NG = p.NG; sa.gshift = p.gshift; sa.gshift_narrow = p.gshift_narrow; sa.slice = p.slice; sa.nblk = p.S;
    sa.jbits = p.jbits;
    sa.cap = p.cap; sa.big = p.big;
    sa.xcd_map = opt.sort_xcd ? 1u : 0u;
    sa.staged = (opt.sort_staged >= 2 || (opt.sort_staged == 1 && p.NG >= 256u)) ? 1u : 0u;
    sa.part = (uint32_t*)need(part, (size_t)W * p.nent * (p.merged ? 8 : 4));
    sa.cntA = (uint32_t*)need(counts, (size_t)p.S * W * p.NG * 4);
    sa.gtot = (uint32_t*)need(totals, (size_t)W * p.NG * 4);
    sa.gbase = (uint32_t*)need(gbase, (size_t)W * (p.NG + 1) * 4);
    Staged st;
    st.d_bstart = (uint32_t*)need(bstartS[sl], (size_t)W * (B + 1) * 4);
    uint32_t* d_entries = (uint32_t*)need(entries, (size_t)W * p.nent * 4);
    st.d_maxcount = (uint32_t*)need(maxcountS[sl], 256);
    // (d_maxcount: [0] the largest bucket, [2] the head merge's queue count -- zeroed by the sort's first kernel: no fill launch)
    sa.bstart = st.d_bstart; sa.entries = d_entries; sa.maxcount = st.d_maxcount;
    // the sort leaves the empty buckets of the set neutral (round 5; a fill launch of the whole set before).  Not for a later
    // slice of a host-pointer MSM (into): its runs continue the stored sums.  The set is this slot's: the previous MSM of the
    // slot read it in its first reduction pass, which the accumulation before this sort waited for.
    sa.zero_base = into ? nullptr : (void*)d_buckets;
    sa.zero_bytes = into ? 0u : (uint32_t)sizeof(XYZZ<FD>);
    bk.launch_digits_sort(sa);   // (leaves the largest bucket in d_maxcount[0]: the merge kernels read it there)
    if (convert_late) {
      (*points_arrive)();
      if constexpr (kConvert) {
        if (!d_prepared) bk.template launch_convert<F, FD>(d_points_in, d_converted, n);   // (timed with the sort in this form)
      }
    }
    bk.stage_end(sl, ST_SORT);
    if (front_side) bk.front_end();     // the accumulation below (main stream) waits for the sort

    // The previous MSM's tail (narrow reduction passes + result copy on the backend's second stream) has had this MSM's
    // conversion and sort to run underneath; the accumulation must not start before it is done: k_accum takes every
    // wave slot of the chip for its whole duration, and a tail kernel enqueued behind it would wait it out (measured on
    // a slower box: reduce span 2.4 ms, the pipeline slower than the serial order).  Worst case this is the serial order.
    // (the sort above has left this slot's bucket set with its empty buckets neutral -- SortArgs::zero_base -- and the previous MSM
    // runs on the other slot's set)
    // (into: a later slice of a host-pointer MSM continues the sums of the earlier slices, accum_body_xyzz)
    // When the accumulate grid leaves wave slots free (submit() sees to that for a caller that keeps MSMs in flight: 1/16 of the
    // slots up to 2^17 pairs, 1/64 of them while that costs less than half the wait), the previous tail's narrow passes run next to it, and the wait moves to the
    // start of this MSM's reduction (reduce_buckets), the first kernel that writes what the tail still reads.
    const uint64_t accum_waves = (uint64_t)W * ((p.G + 63u) / 64u);
    // (partitioned chip, HipBackend::partitioned: the tail stream has compute units of its own -- nothing to wait for, no slots to leave)
    // (Curve::WHOLE_TAIL_LOG2N: large MSMs of the curves whose accumulate kernel owns every register wait for the whole tail whatever the grid leaves free)
    const bool whole_tail = C::WHOLE_TAIL_LOG2N > 0 && p.n >= (1u << C::WHOLE_TAIL_LOG2N);
    if (!bk.partitioned() && (whole_tail || accum_waves + tail_min_free_waves() > (uint64_t)opt.lanes / 64u)) bk.tail_wait();
    bk.wide_wait();   // (nothing to wait for unless the previous reduction put wide passes on the tail stream)
    bk.stage_begin(sl, ST_ACCUM);
    st.d_buckets = d_buckets;
    st.d_heads = (XYZZ<FD>*)need(heads, (size_t)W * p.G * sizeof(XYZZ<FD>));
    st.d_tails = (XYZZ<FD>*)need(tails, (size_t)W * p.G * sizeof(XYZZ<FD>));
    st.d_hkey = (uint32_t*)need(hkey, (size_t)W * p.G * 4);
    st.d_tkey = (uint32_t*)need(tkey, (size_t)W * p.G * 4);
    AccumArgs<FD> aa{d_entries, st.d_bstart, d_points, point_stride, d_buckets, st.d_heads, st.d_tails, st.d_hkey, st.d_tkey, p.nent, B, p.K, p.G};
    bk.template launch_accum<FD>(aa, W, into);
    bk.accum_mark(front_side);          // (what the next small MSM's front stage waits for: the shared entry list and records are free again)
    bk.stage_end(sl, ST_ACCUM);
    return st;
  }

  // Stage 2: partial sums of the buckets that straddle lane ranges.  A bucket of m entries spans at most floor((m-1)/K)+1
  // heads, the tree over such a chain takes log2 steps; with chains of length one -- the common case, no bucket larger than
  // K -- the tail merge writes the buckets itself.  The largest bucket is known on the DEVICE (the sort leaves it in
  // d_maxcount before the accumulation starts) and the kernels decide there: the host enqueues the steps an ordinary digit
  // distribution needs (MsmPlan::merge_steps) plus the finishing launch that covers everything else, and never waits
  // (rounds 1-2 read the word back in the middle of submit: a host round-trip per MSM, 0.1 ms of a 0.7 ms step at 2^16).
  void merge_buckets(int sl, const MsmPlan& p, const Staged& st) {
    bk.stage_begin(sl, ST_MERGE);
    MergeArgs<FD> ma{st.d_bstart, st.d_buckets, st.d_heads, st.d_tails, st.d_hkey, st.d_tkey, st.d_maxcount, p.B, p.K, p.G};
    if (p.merge_lmax > 0) {
      // queue form (round 5; msm_bodies.h merge_tail_queue_body): tail merge + queue of the chains with work left, one lane per
      // queued chain, and the long chains of unusual inputs by one workgroup per window
      ma.queue = (uint32_t*)need(mqueue, (size_t)merge_queue_capacity(p.W, p.G) * 4);
      ma.qcount = st.d_maxcount + 2;     // (zeroed with the largest-bucket word before the sort)
      bk.template launch_merge_tail_queue<FD>(ma, p.W);
      bk.template launch_merge_queue<FD>(ma, p.W, p.merge_lmax, opt.merge_queue_quad != 2);
      bk.template launch_merge_long<FD>(ma, p.W, p.merge_lmax);
    } else {
      bk.template launch_merge_tail<FD>(ma, p.W);
      uint32_t d = 1;
      for (int i = 0; i < p.merge_steps && d < p.G; i++, d <<= 1) bk.template launch_merge_step<FD>(ma, p.W, d);
      bk.template launch_merge_finish<FD>(ma, p.W, d);
    }
    bk.stage_end(sl, ST_MERGE);
  }

  // Stage 3: bucket reduction (c-1 pyramid passes, one launch each: every wave of a launch runs the same straight-line
  // addition at the same time, which is what keeps its 64 KiB of code flowing through the instruction caches), the bit Horner
  // in groups, and the copy of the W*ngrp partial sums to the slot's pinned buffer.
  // Round 3 measured the alternative the round-2 review asked for -- an aligned block of 256..1024 buckets taken through its
  // pyramid levels by ONE workgroup, workgroup barriers instead of launches, and one workgroup per window for the rest: two
  // launches instead of c-1.  Same box, BLS12-381 G1 2^20: 332 + 217 us against 315 + 75 (+ 45 us of Horner in both); 2^16:
  // 128 + 91 against 67 + 88.  A narrow pass is ~10 us of which ~7 us is the latency of one four-lane addition, which a fused
  // kernel pays as well, and a launch spreads every level over the whole chip where a workgroup has 256 lanes; the waves of a
  // fused kernel also drift apart (different passes, different task kinds) and each then streams the addition's code through
  // the instruction cache on its own.  profiles/reduce_fused_vs_passes_r03.txt; the code is in the history (a64cd06).
  void reduce_buckets(int sl, const MsmPlan& p, XYZZ<FD>* d_buckets) {
    Slot& S = slots[sl];
    const uint32_t W = p.W, B = p.B;
    const bool forked_early = bk.tail_forked();   // submit() forked in front of the head merge (early tail)
    if (!forked_early) bk.tail_wait();   // (a no-op unless accumulate_pairs left the previous tail running: small MSMs)
    bk.stage_begin(sl, ST_REDUCE);
    XYZZ<FD>* d_pyr = (XYZZ<FD>*)need(rA[0], (size_t)W * B * sizeof(XYZZ<FD>));
    XYZZ<FD>* d_q = (XYZZ<FD>*)need(rA[1], (size_t)W * (B / 2 + 1) * sizeof(XYZZ<FD>));
    XYZZ<FD>* d_out = (XYZZ<FD>*)need(rP[0], (size_t)W * p.c * sizeof(XYZZ<FD>));
    XYZZ<FD>* d_wsum = (XYZZ<FD>*)need(rP[1], (size_t)W * p.ngrp * sizeof(XYZZ<FD>));
    // The narrow passes at the end, the bit Horner and the result copy move to the backend's tail stream: they are
    // latency-bound and the next MSM's conversion and sort fit underneath them (the tail_wait() before the accumulation also
    // orders the previous tail before this MSM's first write to the pyramid buffers).
    // (only for a caller that keeps MSMs in flight -- the other slot is busy: a lone blocking call would pay the fork's
    // event record and wait, ~15 us, for nothing)
    const bool pipelining = other_busy(sl);
    // wide_early: every pass but the first goes to the tail stream, the wide ones next to the following MSM's conversion and sort
    // (VALU-bound additions beside memory-bound kernels); that MSM's accumulation waits for the end of the WIDE passes only
    // (wide_mark / wide_wait) -- the narrow rest runs beside it in the wave slots its grid leaves free.
    // Same box, ms per MSM with / without (profiles/wide_passes_on_tail_r03.txt): BLS12-381 G1 2^20 2.92 / 2.97, 2^18 0.965 / 0.977, 2^22 10.70 /
    // 10.75; no difference for the other curves -- the kernels do slow each other down (round 2 measured the sort 0.19 -> 0.23 ms
    // under wide passes), a quarter of the overlap is what remains.
    static const bool wide_early = !(getenv("CTT_HIP_MSM_WIDE_EARLY") && atoi(getenv("CTT_HIP_MSM_WIDE_EARLY")) == 0);
    bool forked = forked_early, marked = forked_early && bk.partitioned();   // (partitioned: submit() marked behind the merge)
    const bool first_pass_on_tail = opt.pyr0_tail == 1 && p.n <= (1u << 17) && !p.merged;   // (MsmOptions::pyr0_tail: measured, off)
    bk.narrow_priority(C::NARROW_PRIO_LOG2N > 0 && p.n <= (1u << C::NARROW_PRIO_LOG2N));
    for (int pass = 0; pass <= p.c - 2; pass++) {
      PyrArgs<FD> pa{d_buckets, d_pyr, d_q, d_out, B, p.c, pass, 1u};
      const uint32_t ntasks = pyr_pass_tasks(B, p.c, pass);
      const bool narrow = bk.pyr_goes_to_tail(ntasks, W);
      if (pipelining && !forked && (pass > 0 || first_pass_on_tail) && (narrow || wide_early)) {
        bk.tail_begin();
        forked = true;
      }
      if (forked && !marked && narrow) {
        bk.wide_mark();
        marked = true;
      }
      bk.template launch_pyr<FD>(pa, W, ntasks);
    }
    if (pipelining && !forked) {
      bk.tail_begin();
      forked = true;
    }
    if (forked && !marked) bk.wide_mark();
    bk.template launch_window_groups<FD>(d_out, d_wsum, W, p.c, p.h, p.ngrp);
    bk.stage_end(sl, ST_REDUCE);

    const size_t bytes = (size_t)W * p.ngrp * sizeof(XYZZ<FD>);
    if (bytes > S.hcap) {
      if (S.hraw) bk.free_host(S.hraw);
      S.hraw = bk.alloc_host(bytes);
      S.hcap = bytes;
    }
    bk.d2h_async(sl, S.hraw, d_wsum, bytes);
    bk.stage_end(sl, ST_TOTAL);
    if (forked) bk.tail_end();
  }

  // Wave slots the accumulate grid must leave free for the previous MSM's tail to run beside it (fewer: the accumulation waits
  // for that tail first), and how much of an accumulation submit() may spend on leaving slots free on purpose for an MSM kept
  // in flight ...
  static uint32_t tail_min_free_waves() {
    static const uint32_t v = getenv("CTT_HIP_MSM_TAIL_MIN_FREE") ? (uint32_t)atoi(getenv("CTT_HIP_MSM_TAIL_MIN_FREE")) : 16u;
    return v;
  }
  // ... as a fraction of the wait it removes, which is about one tenth of a millisecond for BLS12-381 G1 and scales with the
  // curve's addition time (the tail is a chain of dependent additions)
  static double tail_free_cost_ratio() {
    static const double v = getenv("CTT_HIP_MSM_TAIL_FREE_RATIO") ? atof(getenv("CTT_HIP_MSM_TAIL_FREE_RATIO")) : 0.5;
    return v;
  }

  int claim_slot(uint32_t n) {
    int sl = next_slot;
    for (int k = 0; k < NSLOT && slots[sl].busy; k++) sl = (sl + 1) % NSLOT;    // tickets may be finished in any order: take whichever slot is free
    Slot& S = slots[sl];
    if (S.busy) return -1;  // NSLOT MSMs in flight already: the caller finishes the oldest first (C ABI: error code)
    next_slot = (sl + 1) % NSLOT;
    S.busy = true;
    S.empty = (n == 0);  // len == 0 is UB upstream; we return the neutral
    return sl;
  }

  // One MSM on device-resident inputs.  d_prepared (optional): records made by prepare_bases for the same points; skips
  // the per-MSM conversion.  Returns the slot, or -1 when both slots are in flight.
  // d_prepared with table_c > 0: a window table built by prepare_table over table_n bases with table_c window bits.
  int submit(const uint32_t* d_coefs, bool coef_is_fr, const Affine<F>* d_points_in, uint32_t n,
             const void* d_prepared = nullptr, int table_c = 0, uint32_t table_n = 0) {
    const int sl = claim_slot(n);
    if (sl < 0 || n == 0) return sl;
    opt.acc_ns = C::ACC_NS;
    opt.red_ns = C::RED_NS;
    MsmOptions po = opt;
    // A caller that keeps MSMs in flight gets the tail of the previous MSM (reduction passes behind the first, bit Horner, result
    // copy: latency-bound, a few dozen waves at a time) run BESIDE this accumulation when the accumulate grid leaves wave slots free
    // (accumulate_pairs); it pays to leave them free on purpose.  Up to 2^17 pairs: 1/16 of the lanes.  (Round 2 took 5/32 -- BLS12-381
    // G1 2^17, ms per MSM with two in flight: 0.69 with 17 % of the slots free, 0.80 with 5 %; since the larger sizes stopped waiting
    // for the tail too, 1/16 measures level or better: 2^17 0.742 against 0.759 ms with 5/32, G2 2^16 1.048 against 1.074.)
    // (the window size is chosen for the whole chip first: fewer lanes must only lengthen K)
    if (bk.partitioned()) {
      // (the tail runs on compute units the accumulate grid never sees: opt.lanes counts the main stream's CUs only)
    } else if (other_busy(sl) && n <= (1u << 17) && opt.K <= 0 && table_c <= 0) {
      if (po.c <= 0) po.c = choose_window_bits(n, C::BITS, opt.lanes, opt.acc_ns, opt.red_ns);
      static const uint32_t free32 = getenv("CTT_HIP_MSM_SMALL_FREE_32NDS") ? (uint32_t)atoi(getenv("CTT_HIP_MSM_SMALL_FREE_32NDS")) : 2u;
      po.lanes = (uint32_t)((uint64_t)opt.lanes * (32u - (free32 < 31u ? free32 : 31u)) / 32u);
    } else if (other_busy(sl) && opt.K <= 0 && table_c <= 0 && opt.lanes >= 64u * 1024u && opt.acc_ns >= 0.1) {
      // Larger ones: the accumulation used to wait for the previous tail -- ten dependent narrow passes, the bit Horner and the result
      // copy, ~0.25 ms after the last wide pass, 0.1 ms longer than this MSM's sort (rocprof timeline, BLS12-381 2^20: the sort ends
      // at 177 us, the accumulation started at 281).  With 1/64 of the wave slots left free (32 of 2048; K 128 -> 131 at 2^20) the
      // tail finishes beside the accumulation instead.  Same box, ms per MSM with / without: BLS12-381 G1 2^19 1.72 / 1.82,
      // 2^20 2.92 / 2.98, 2^21 5.54 / 5.59, 2^22 10.60 / 10.47; G2 2^18 2.92 / 3.13, 2^20 9.50 / 9.46 -- it pays while that share of the
      // accumulation is well below the wait (profiles/sweep_free_wave_slots_r03.txt).  Not for the 254/255-bit G1 fields (acc_ns < 0.1):
      // their additions are twice as fast, their tail ends before their sort does, and the free slots only cost (Pallas 2^20 1.455 / 1.433).
      if (po.c <= 0) po.c = choose_window_bits(n, C::BITS, opt.lanes, opt.acc_ns, opt.red_ns);
      int Wc;
      window_layout(C::BITS, po.c, &Wc);
      // 32 slots at least: a narrow pass is up to 64 waves, and a G2 wave needs a SIMD to itself (16 free slots of its 1024
      // measured no gain, 32 did: 3.13 -> 2.92 ms at 2^18)
      const uint32_t nslots = opt.lanes / 64u, free_slots = nslots / 64u > 32u ? nslots / 64u : 32u;
      const double cost_ms = (double)Wc * n * opt.acc_ns * 1e-6 * free_slots / nslots, wait_ms = 0.1 * opt.acc_ns / 0.142;
      if (cost_ms <= tail_free_cost_ratio() * wait_ms) po.lanes = opt.lanes - 64u * free_slots;
    }
    const MsmPlan p = table_c > 0 ? make_table_plan(n, C::BITS, table_c, table_n, po) : make_plan(n, C::BITS, po);
    slots[sl].plan = p;
    last_plan = p;
    try {
      bk.stage_begin(sl, ST_TOTAL);
      void* d_converted = nullptr;
      if constexpr (kConvert) {
        if (!d_prepared) d_converted = need(cpoints, (size_t)n * gather_stride<FD>());
      }
      XYZZ<FD>* d_buckets = (XYZZ<FD>*)need(bucketsS[sl], (size_t)p.W * p.B * sizeof(XYZZ<FD>));
      // Round 5 experiment (option front_side = 1; off by default: front_side_applies has the measurements): conversion and sort of a small MSM
      // submitted while another is in flight on the front stream, beside that MSM's head merge and first reduction pass.
      const bool front_side = front_side_applies(p);
      const Staged st = accumulate_pairs(sl, p, d_coefs, coef_is_fr, d_points_in, d_prepared, d_converted, d_buckets, /*into=*/false, nullptr, front_side);
      // Early tail (round 4): a caller that keeps large MSMs in flight gets the head merge and EVERY reduction pass of this MSM on the
      // tail stream, so that the next MSM's conversion and sort -- memory-bound -- start right behind this accumulation instead of
      // behind the merge and the widest pass (VALU-bound additions, 0.16 ms at 2^20).  What those stages read is per slot (bstartS,
      // maxcountS, bucketsS); the next accumulation still waits for the end of the wide passes (wide_wait), so nothing of this tail
      // competes with an accumulation for wave slots (which is what sank round 3's version for small MSMs, section 5 of DESIGN.md).
      if (early_tail_applies(p)) {
        // (the previous tail ended long ago: it ran beside this accumulation.  Partitioned chip: it may still be crawling through its
        // wide passes on its few compute units -- the tail stream is in order, and nothing on the main stream touches what it uses)
        if (!bk.partitioned()) bk.tail_wait();
        bk.tail_begin();
      }
      merge_buckets(sl, p, st);
      // Partitioned chip + early tail: the next accumulation shares the head / tail slots with this merge and nothing else with this
      // tail -- it waits for the merge, not for the wide reduction passes (which run on the tail stream's own compute units)
      if (bk.partitioned() && bk.tail_forked()) bk.merge_mark();
      reduce_buckets(sl, p, d_buckets);
    } catch (const OutOfDeviceMemory&) {
      return release_slot(sl);
    }
    return sl;
  }
  // Only for a caller that is pipelining (the other slot is busy), from ~2^18 pairs on, and while the accumulation is shorter than ~6 ms.
  // Measured on one box, ms per MSM with two in flight, off / on, three repetitions (profiles/early_tail_r04.txt): BLS12-381 G1 2^19
  // 1.70 / 1.63, 2^20 2.90 / 2.81 (-3.1 %), BN254 2^20 1.62 / 1.58, 2^22 5.39 / 5.25, Pallas 2^20 1.46 / 1.40 (-3.9 %), 2^18 1.011 / 0.998;
  // but BLS12-381 G1 2^22 10.5 / 10.6 and G2 2^20 9.45 / 9.49: with a 8-10 ms accumulation the 0.16 ms are 1.5 % at best and the sort
  // (0.5 ms at 2^22) running beside the widest pass loses more than the overlap gives.
  bool front_side_applies(const MsmPlan& p) const {
    static const int mode = getenv("CTT_HIP_MSM_FRONT") ? atoi(getenv("CTT_HIP_MSM_FRONT")) : 1;   // 0 off, 1 automatic, 2 whenever pipelining
    if (mode == 0 || opt.front_side == 2) return false;
    if (busy_count() < 2) return false;    // a lone blocking call has nothing to run beside
    if (mode >= 2 || opt.front_side == 1) return true;
    // Measured and NOT adopted (profiles/front_stream_small_msm_r05.txt, same box, ms per MSM with two in flight, main stream only / front stream):
    // BLS12-381 G1 2^12 0.278 / 0.287, 2^14 0.388-0.393 / 0.383-0.386, 2^16 0.470 / 0.494, 2^17 0.637 / 0.664, G2 2^16 1.077 / 1.227.  The overlap
    // is real (the sort leaves the main stream's chain), but with a second hardware queue active the dependent kernels of BOTH chains start later
    // (round 3 saw the same with two engine lanes: 50-70 us between dependent launches).  What the experiment did find: a process has FOUR hardware
    // queues (GPU_MAX_HW_QUEUES); with the front stream as a FIFTH stream of the context (null, main, tail, copy, front) two streams shared a queue
    // and every small pipelined MSM lost 8 % (2^16: 0.52 -> 0.56 ms) without a single kernel on the new stream -- the front stage therefore
    // runs on the copy stream (HipBackend::init), and nothing in the library may add a stream lightly.
    (void)p;
    return false;
  }
  bool early_tail_applies(const MsmPlan& p) const {
    static const int mode = getenv("CTT_HIP_MSM_EARLY_TAIL") ? atoi(getenv("CTT_HIP_MSM_EARLY_TAIL")) : 1;   // 0 off, 1 automatic, 2 whenever pipelining
    if (mode == 0 || opt.early_tail == 0) return false;
    const bool pipelining = busy_count() >= 2;
    if (!pipelining) return false;
    if (mode >= 2 || opt.early_tail >= 2) return true;
    const double additions = (double)p.nent * (double)p.W;
    return additions >= (double)(1u << 22) && additions * opt.acc_ns < 6.0e6;
  }
  // a submit that ran out of device memory: what it enqueued so far runs to its end on buffers that stay valid (a buffer is
  // only ever freed by need(), and hipFree waits for the device); the slot is free again.  Returns the error value -2.
  int release_slot(int sl) {
    bk.front_abort();       // (an exception between front_begin and front_end must not leave the next MSM's sort on the front stream: ADVICE r5)
    slots[sl].busy = false;
    next_slot = sl;
    return -2;
  }

  // One MSM on HOST-resident inputs (what the Constantine C symbols hand over), uploaded in `chunks` slices of pairs so
  // that the upload of slice i+1 runs underneath the accumulation of slice i: the copies go through the backend's copy
  // stream (a pageable hipMemcpyAsync occupies the calling thread but not the GPU's compute queues, and runs at PCIe
  // speed while k_accum holds every wave slot: profiles/h2d_overlap_r02.jsonl).  Every slice is sorted and accumulated
  // into its own bucket set (one owner per bucket per launch, no atomics); the sets are summed before the one bucket
  // reduction.  d_stage_coefs / d_stage_points: device staging for all n pairs (caller-owned).  Blocking on the copies,
  // asynchronous from the last accumulation on; returns the slot, or -1 when both slots are in flight.
  // The slices of a host-pointer call: bound[0] = 0 < bound[1] < ... < bound[nch] = n.
  // Model (round 4; fitted to the timeline in profiles/hostptr_timeline_r04.txt): the link moves a pair in copy_ns (56 GB/s pageable,
  // profiles/h2d_overlap_r02.jsonl), the copies of slice i end at C_i = copy_ns * (pairs up to and including slice i); the GPU takes
  // gpu_ns per pair (windows x the curve's accumulate time) plus a fixed fix_ns per slice (conversion, the sort's launch chain, the
  // head merge: ~0.19 ms whatever the slice holds) and finishes slice i at F_i = max(F_(i-1), C_i) + gpu_ns * s_i + fix_ns.  Sizes
  // s_i ~ r^i; the slice count (1..6, or the caller's) and r (0.5..2.5) are the pair with the smallest F_last, a further slice
  // having to buy 3 %.  GPU-bound curves (BLS12-381 G1: 2.5 against 2.3 ns per pair; G2) come out with a small first slice -- its
  // copy is the only one exposed -- and growing ones after it (2^20 BLS12-381 G1 pairs: 24 / 32 / 44 %, G2: 12 / 27 / 61 %); copy-bound
  // curves (the 254/255-bit ones, 1.2 against 1.7 ns -- the Halo2-ZAL configuration) with shrinking ones, since what is exposed there
  // is the last slice's GPU work (BN254 2^22: 24 / 20 / 17 / 15 / 13 / 11 %).  Small calls stay whole: one slice up to 2^17 pairs, two at
  // 2^18.  Rounds 2-3 took weights g^i with g = gpu_ns / copy_ns clamped to [0.7, 1.4] and 2 / 3 / 4 slices from 2^18 / 3 * 2^18 / 2^21.
  // Measured, same box, old / new (gpurun_out/r4l -> profiles/hostptr_r04.txt, ms per call): BLS12-381 G1 2^20 4.47-4.58 / 4.36-4.58 (level),
  // 2^22 14.0-14.2 / 13.3-13.8, 2^24 52.1 / 49.6-50.1; G2 2^20 12.8-13.0 / 12.1-12.2; BN254 2^22 9.33-9.35 / 8.64-8.75; Pallas 2^20 2.87 / 2.86.
  // Every slice a multiple of 64 pairs except the last.  An explicit slice count is honoured.
  static constexpr double HOST_SLICE_FIX_NS = 1.9e5;
  static std::vector<uint32_t> host_slices(uint32_t n, int want, bool scalars_only = false) {
    // (scalars_only: the bases are cached on the device, 32 bytes per pair cross the link)
    const double gpu_ns = (double)((C::BITS + 16) / 16) * C::ACC_NS * 1.09, copy_ns = (double)(32 + (scalars_only ? 0 : sizeof(Affine<F>))) / 56.0;
    if (want <= 0 && n < (1u << 15)) return std::vector<uint32_t>{0u, n};   // (small calls stay whole; no search for a 4096-point commitment)
    auto sizes = [&](uint32_t nch, double r) {
      std::vector<uint32_t> bound(nch + 1, 0);
      double wsum = 0, w = 1;
      for (uint32_t i = 0; i < nch; i++, w *= r) wsum += w;
      double acc = 0;
      w = 1;
      for (uint32_t i = 0; i + 1 < nch; i++, w *= r) {
        acc += w;
        uint64_t b = (uint64_t)((double)n * acc / wsum);
        b &= ~63ull;
        if (b <= bound[i]) b = bound[i] + 1;     // tiny inputs: at least one pair per slice
        if (b > n - (nch - 1 - i)) b = n - (nch - 1 - i);
        bound[i + 1] = (uint32_t)b;
      }
      bound[nch] = n;
      return bound;
    };
    auto finish_ns = [&](const std::vector<uint32_t>& bound) {
      double f = 0;
      // (two slices are copied by the submitting thread, which enqueues the first slice's launches in between: ~0.1 ms of idle link)
      const double gap_ns = bound.size() == 3 ? 1.0e5 : 0.0;
      for (size_t i = 0; i + 1 < bound.size(); i++) {
        const double c = copy_ns * (double)bound[i + 1] + gap_ns * (double)i;
        f = (f > c ? f : c) + gpu_ns * (double)(bound[i + 1] - bound[i]) + HOST_SLICE_FIX_NS;
      }
      return f;
    };
    uint32_t lo = 1, hi = 6;
    if (want > 0) lo = hi = (uint32_t)(want > 8 ? 8 : want);
    if (hi > n) hi = n;
    if (lo > hi) lo = hi;
    std::vector<uint32_t> best;
    double best_ns = 0;
    for (uint32_t nch = lo; nch <= hi; nch++) {
      std::vector<uint32_t> b = sizes(nch, 1.0);
      double t = finish_ns(b);
      for (int k = 0; k <= 40 && nch > 1; k++) {
        const std::vector<uint32_t> cand = sizes(nch, 0.5 + 0.05 * k);
        const double tc = finish_ns(cand);
        if (tc < t) {
          t = tc;
          b = cand;
        }
      }
      if (best.empty() || t < best_ns * 0.97) {   // (a further slice has to buy 3 %: small slices accumulate less efficiently than the model says)
        best_ns = t;
        best = b;
      }
    }
    return best;
  }
  // d_prepared (with table_c / table_n for a window table): the bases are cached on the device (prepare_bases / prepare_table), only the
  // coefficients are host-resident -- the ZAL msm_with_cached_base shape with host scalars; h_points and d_stage_points are not used.
  int submit_host(const void* h_coefs, bool coef_is_fr, const void* h_points, uint32_t n, void* d_stage_coefs,
                  void* d_stage_points, int want_chunks, const void* d_prepared = nullptr, int table_c = 0, uint32_t table_n = 0) {
    const int sl = claim_slot(n);
    if (sl < 0 || n == 0) return sl;
    const std::vector<uint32_t> bound = host_slices(n, want_chunks, d_prepared != nullptr);
    const size_t prepared_stride = kConvert ? (size_t)gather_stride<FD>() : sizeof(Affine<F>);
    auto plan_of = [&](uint32_t cnt, const MsmOptions& oo) {
      return table_c > 0 ? make_table_plan(cnt, C::BITS, table_c, table_n, oo) : make_plan(cnt, C::BITS, oo);
    };
    const uint32_t nch = (uint32_t)bound.size() - 1;
    uint32_t largest = 0;
    for (uint32_t i = 0; i < nch; i++) largest = bound[i + 1] - bound[i] > largest ? bound[i + 1] - bound[i] : largest;
    opt.acc_ns = C::ACC_NS;
    opt.red_ns = C::RED_NS;
    MsmOptions o = opt;
    if (o.c <= 0) o.c = choose_window_bits(n, C::BITS, o.lanes, o.acc_ns, o.red_ns);  // one window size for the whole MSM
    try {
    bk.stage_begin(sl, ST_TOTAL);
    const MsmPlan p0 = plan_of(largest, o);   // the largest slice sizes the workspace
    // ONE bucket set for all slices (round 4): the accumulation of slice i > 0 continues the stored sums (k_accum<FD, true>).  Rounds
    // 2-3 gave every slice its own set and added the sets afterwards: a full addition per bucket and slice, 0.24 ms of a 4.6 ms call
    // at 2^20 BLS12-381 G1 in three slices (k_bucket_sum, profiles/hostptr_timeline_r04.txt).
    XYZZ<FD>* d_sets = (XYZZ<FD>*)need(bucketsS[sl], (size_t)p0.W * p0.B * sizeof(XYZZ<FD>));
    void* d_conv_all = nullptr;
    if constexpr (kConvert) {
      if (!d_prepared) d_conv_all = need(cpoints, (size_t)n * gather_stride<FD>());
    }
    reserve_stage1(sl, p0, coef_is_fr);
    MsmPlan plast = p0;
    Staged st_prev{};
    MsmPlan p_prev = p0;
    auto upload_coefs = [&](uint32_t i) {
      const uint32_t start = bound[i], cnt = bound[i + 1] - bound[i];
      bk.h2d((uint32_t*)d_stage_coefs + (size_t)start * 8, (const char*)h_coefs + (size_t)start * 32, (size_t)cnt * 32);
    };
    auto upload_points = [&](uint32_t i) {
      const uint32_t start = bound[i], cnt = bound[i + 1] - bound[i];
      if (!d_prepared)
        bk.h2d((Affine<F>*)d_stage_points + start, (const char*)h_points + (size_t)start * sizeof(Affine<F>), (size_t)cnt * sizeof(Affine<F>));
    };
    // Several slices: a thread of its own issues the copies back to back (HipBackend::h2d_slice_done has the reason), this one
    // enqueues slice i's kernels as soon as slice i has been handed to the link.
    bool threaded = BK::THREADED_UPLOAD && nch > 2;   // (two slices: one gap of ~0.1 ms against a thread start of about as much -- measured: 2^18 1.85 ms without, 1.93 with)
    std::atomic<uint32_t> uploaded{0}, coefs_up{0};
    std::atomic<bool> upload_failed{false};   // a HIP call failed on the uploader thread: reported by THIS thread (the one an entry point's error channel belongs to)
    std::thread uploader;
    struct Joiner {
      std::thread& t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{uploader};                          // (also on the out-of-memory exit: the staging buffers outlive this call)
    if (threaded) {
      try {
        uploader = std::thread([&]() {
          // (a failed copy on this thread must not abort a process whose entry point can report: the guard makes HIP_CHECK throw here,
          // the flag hands the failure to the submitting thread -- ADVICE r5; the emulator's guard is empty)
          typename BK::GuardScope guard;
          try {
          bk.uploader_begin();
          for (uint32_t i = 0; i < nch; i++) {
            upload_coefs(i);
            // An event between the two copies only for the first slice.  Measured with a record after every slice's coefficients
            // (rocprofv3 --memory-copy-trace): the copy behind such a record started only when the accumulation running on the main
            // stream had ended -- no overlap left, 2^20 pairs 5.5 ms instead of 4.4, 2^22 19.6 instead of 13.5.  (A record is a
            // barrier packet in a hardware queue, and the copy ordered behind it waits for it; giving the copy stream a priority
            // class of its own changed nothing.)  Behind the first slice's coefficients nothing long is running.
            if (i == 0 && !d_prepared) bk.h2d_coefs_done(i);
            coefs_up.store(i + 1, std::memory_order_release);
            upload_points(i);
            bk.h2d_slice_done(i);
            uploaded.store(i + 1, std::memory_order_release);
          }
          } catch (...) {
            upload_failed.store(true, std::memory_order_release);
            coefs_up.store(nch, std::memory_order_release);     // (release the submitting thread's waits)
            uploaded.store(nch, std::memory_order_release);
          }
        });
      } catch (const std::system_error&) {
        threaded = false;   // (no thread to be had: this one copies, as with two slices)
      }
    }
    for (uint32_t i = 0; i < nch; i++) {
      const uint32_t start = bound[i], cnt = bound[i + 1] - bound[i];
      bk.stage_chunk((int)i);   // stage events of this slice (the stage times of the call are the sums over its slices)
      uint32_t* d_c = (uint32_t*)d_stage_coefs + (size_t)start * 8;
      Affine<F>* d_p = (Affine<F>*)d_stage_points + start;
      auto await = [&](std::atomic<uint32_t>& flag, uint32_t i) {
        for (uint32_t spin = 0; flag.load(std::memory_order_acquire) <= i; spin++) {
          if (spin < 20000u) cpu_relax(); else std::this_thread::yield();
        }
        if (upload_failed.load(std::memory_order_acquire)) bk.uploader_failed();   // (throws to the entry point's boundary, or aborts where there is none)
      };
      // The first slice starts on its coefficients: the digits and the sort need nothing else, and they run while the slice's points
      // cross the link (accumulate_pairs, points_arrive).  Later slices wait for both copies at once (their sort queues behind the
      // previous accumulation anyway, and an event between their copies would hold the second one back: see the uploader).
      static const bool late_env = !(getenv("CTT_HIP_MSM_LATE_POINTS") && atoi(getenv("CTT_HIP_MSM_LATE_POINTS")) == 0);   // (0: both copies first)
      const bool late_points = late_env && i == 0 && !d_prepared;
      if (threaded) {
        if (late_points) {
          await(coefs_up, i);
          bk.h2d_coefs_wait(i);                                                                   // main stream waits for the coefficients
        } else {
          await(uploaded, i);
          bk.h2d_slice_wait(i);                                                                   // ... for both copies
        }
      } else {
        upload_coefs(i);
        if (late_points) bk.h2d_done();
        else {
          upload_points(i);
          bk.h2d_done();
        }
      }
      const std::function<void()> points_arrive = [&, i]() {
        if (threaded) {
          await(uploaded, i);     // (the uploader thread has handed the slice's points to the link)
          bk.h2d_slice_wait(i);
        } else {
          upload_points(i);
          bk.h2d_done();
        }
      };

      if (i > 0) {   // slice i-1's merge goes in front of slice i's kernels (shared workspace)
        bk.stage_chunk((int)i - 1);
        merge_buckets(sl, p_prev, st_prev);
        bk.stage_chunk((int)i);
      }
      const MsmPlan p = plan_of(cnt, o);
      if (p.W != p0.W || p.B != p0.B) {   // (the slices share one bucket set: same windows, same buckets -- o.c is fixed above)
        fprintf(stderr, "[ctt_msm] FATAL: slice %u of a host-pointer MSM planned %u windows of %u buckets, the call %u of %u\n", i, p.W, p.B, p0.W, p0.B);
        abort();
      }
      void* d_conv = (kConvert && !d_prepared) ? (void*)((char*)d_conv_all + (size_t)start * gather_stride<FD>()) : nullptr;
      // (cached bases: the slice's records -- or, in a window table, its column of every row block: row w * table_n + j -- start `start` records in)
      const void* d_prep = d_prepared ? (const void*)((const char*)d_prepared + (size_t)start * prepared_stride) : nullptr;
      st_prev = accumulate_pairs(sl, p, d_c, coef_is_fr, d_prepared ? nullptr : d_p, d_prep, d_conv, d_sets, /*into=*/i > 0,
                                 late_points ? &points_arrive : nullptr);
      p_prev = p;
      plast = p;
    }
    merge_buckets(sl, p_prev, st_prev);
    plast.n = n;
    slots[sl].plan = plast;
    last_plan = plast;
    last_chunks = nch;
    reduce_buckets(sl, plast, d_sets);
    } catch (const OutOfDeviceMemory&) {
      return release_slot(sl);
    }
    return sl;
  }
  uint32_t last_chunks = 1;

  // Host tail of a submitted MSM: wait for its W window sums, Horner over the windows.
  // (Round 6 built this tail on a helper thread of the engine -- handed over at the end of submit(), finish() only collecting the result --
  // on the reading that the caller's thread, 0.1 ms of enqueueing + 0.1-0.2 ms of tail per MSM, is what starves the main stream of a small
  // pipelined MSM.  Measured, same box, same process, finishing thread / helper thread: 2^14 0.310 / 0.309 ms per MSM, 2^16 0.428 / 0.428,
  // 2^17 0.601 / 0.604 with three in flight, 2^16 0.449 / 0.444 with two -- nothing: with a third slot the GPU has the next MSM queued
  // whatever the host does meanwhile.  Removed again; EXPERIMENTS.md has the table.)
  XYZZ<HF> finish(int sl) {
    Slot& S = slots[sl];
    if (!S.busy) {
      fprintf(stderr, "[ctt_msm] FATAL: finish() of a slot that was not submitted\n");
      abort();
    }
    S.busy = false;
    if (S.empty) return XYZZ<HF>::inf();
    bk.d2h_wait(sl);
    const MsmPlan& p = S.plan;
    const XYZZ<FD>* raw = (const XYZZ<FD>*)S.hraw;
    const size_t cnt = (size_t)p.W * p.ngrp;
    std::vector<XYZZ<HF>> sums(cnt);
    for (size_t i = 0; i < cnt; i++) sums[i] = xyzz_to_host<FD>(raw[i]);
    return combine_groups<HF>(sums.data(), p.W, p.lay, p.h, p.ngrp);
  }

  bool in_flight(int sl) const { return sl >= 0 && sl < NSLOT && slots[sl].busy; }

  // r = sum of n affine points (sum_reduce_vartime, ec_shortweierstrass_batch_ops.nim:649-663).  One window, one
  // bucket: the identity entry list goes through the accumulate kernel (K points per lane) and the head-merging
  // tree; blocking.  Shares the MSM workspace (stream order keeps it apart from MSMs in flight).
  uint32_t last_sum_K = 0;
  XYZZ<HF> sum_reduce(const Affine<F>* d_points_in, uint32_t n) {
    if (n == 0) return XYZZ<HF>::inf();
    bk.tail_wait();  // shares the workspace with MSMs in flight
    const uint32_t lanes = opt.lanes ? opt.lanes : 65536;
    uint32_t K = (uint32_t)(((uint64_t)n + lanes - 1) / lanes);
    if (K < 16) K = 16;
    if (opt.K > 0) K = (uint32_t)opt.K;
    const uint32_t G = (n + K - 1) / K;
    last_sum_K = K;
    const void* d_points;
    uint32_t point_stride;
    if constexpr (kConvert) {
      void* cp = need(cpoints, (size_t)n * gather_stride<FD>());
      bk.template launch_convert<F, FD>(d_points_in, cp, n);
      d_points = cp;
      point_stride = gather_stride<FD>();
    } else {
      d_points = (const void*)d_points_in;
      point_stride = (uint32_t)sizeof(Affine<F>);
    }
    uint32_t* d_entries = (uint32_t*)need(entries, (size_t)n * 4);
    uint32_t* d_bstart = (uint32_t*)need(bstartS[0], 8);
    uint32_t* d_maxcount = (uint32_t*)need(maxcountS[0], 256);
    bk.launch_iota(d_entries, n, d_bstart, d_maxcount);
    XYZZ<FD>* d_buckets = (XYZZ<FD>*)need(bucketsS[0], sizeof(XYZZ<FD>));
    bk.memset0(d_buckets, sizeof(XYZZ<FD>));
    XYZZ<FD>* d_heads = (XYZZ<FD>*)need(heads, (size_t)G * sizeof(XYZZ<FD>));
    XYZZ<FD>* d_tails = (XYZZ<FD>*)need(tails, (size_t)G * sizeof(XYZZ<FD>));
    uint32_t* d_hkey = (uint32_t*)need(hkey, (size_t)G * 4);
    uint32_t* d_tkey = (uint32_t*)need(tkey, (size_t)G * 4);
    AccumArgs<FD> aa{d_entries, d_bstart, d_points, point_stride, d_buckets, d_heads, d_tails, d_hkey, d_tkey, n, 1, K, G};
    bk.template launch_accum<FD>(aa, 1);
    MergeArgs<FD> ma{d_bstart, d_buckets, d_heads, d_tails, d_hkey, d_tkey, d_maxcount, 1, K, G};
    bk.template launch_merge_tail<FD>(ma, 1);
    uint32_t d = 1;
    for (; d < G; d <<= 1) bk.template launch_merge_step<FD>(ma, 1, d);
    bk.template launch_merge_finish<FD>(ma, 1, d);
    XYZZ<FD> raw;
    bk.d2h_sync(&raw, d_buckets, sizeof(raw));
    return xyzz_to_host<FD>(raw);
  }
};

// ---------------------------------------------------------------------------------------------
// Output conversion to the caller's coordinate system (C API structs, include/constantine/curves/*.h)
// ---------------------------------------------------------------------------------------------
enum { OUT_AFF = 0, OUT_JAC = 1, OUT_PRJ = 2 };

// EC_ShortW_Jac neutral (1,1,0) jacobian.nim:58-64; EC_ShortW_Prj neutral (0,1,0) projective.nim:56-62.
// Any representative of the same group element is a valid result (the reference's own raw coordinates
// differ between its serial and parallel paths); we emit the canonical one with Z = 1.
template <class F>
static inline void write_result(void* r, const XYZZ<F>& res, int out_kind) {
  Affine<F> a = xyzz_to_affine<F>(res);
  F* o = (F*)r;
  if (out_kind == OUT_AFF) {
    o[0] = a.x;
    o[1] = a.y;
    return;
  }
  if (a.is_inf()) {
    o[0] = out_kind == OUT_JAC ? F::one() : F::zero();
    o[1] = F::one();
    o[2] = F::zero();
  } else {
    o[0] = a.x;
    o[1] = a.y;
    o[2] = F::one();
  }
}

}  // namespace ctt
