/* ctt_msm_hip.h -- C ABI of libctt_msm_hip.so, the MI355X (gfx950) multi-scalar-multiplication engine.
 *
 * Part 1 re-declares, with identical names, argument order and struct layouts, the MSM symbols of the
 * reference's generated C API, so the library can be linked in place of libconstantine for this path:
 *
 *   serial   : bindings/c_curve_decls.nim:418-431           -> include/constantine/curves/bls12_381.h:163-164,184-185,212-213,233-234
 *                                                               include/constantine/curves/bn254_snarks.h:163-164,184-185,212-213,233-234
 *                                                               include/constantine/curves/pallas.h:125-126,146-147
 *                                                               include/constantine/curves/vesta.h:125-126,146-147
 *   parallel : bindings/c_curve_decls_parallel.nim:33-45    -> include/constantine/curves/{bls12_381,bn254_snarks,pallas,vesta}_parallel.h:20-23
 *
 * All pointers are HOST pointers, buffers are caller-owned, `r` is out-only, the functions block until the
 * result is written, return void and never report an error (same contract as the reference); there is no CPU
 * fallback, so a call of these `void` symbols that the GPU cannot serve aborts with a diagnostic.  Every symbol with a
 * return value -- the neutral spellings of the same MSM (Part 1c), Part 2, and the protocol symbols of Part 3 --
 * reports a GPU refusal through it instead and never aborts (round 5; ctt_hip_last_error() says why).  `tp` is accepted and ignored (the GPU
 * replaces the thread pool).  len == 0 yields the neutral element (undefined behaviour upstream).
 * The result is the same group element as the reference's; its (X,Y,Z) representative is the canonical one
 * with Z = 1 (neutral: jac (1,1,0), prj (0,1,0)).
 *
 * Part 2 is the device-resident interface (inputs already in HBM): the hook for Halo2-ZAL style cached bases
 * (constantine-rust/constantine-halo2-zal/src/lib.rs:60-95) and what bench.py times.
 */
#ifndef CTT_MSM_HIP_H
#define CTT_MSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types: layout-compatible with include/constantine/core/datatypes.h:42-50, curves/bigints.h:18-21,
 *      curves/bls12_381.h:19-27, bn254_snarks.h:19-27, pallas.h:19-24, vesta.h:19-24 ---------------------- */
#ifndef CTT_MSM_HIP_NO_TYPES
typedef size_t secret_word;
#define CTT_WORD_BITWIDTH (sizeof(secret_word) * 8)
#define CTT_WORDS_REQUIRED(bits) (((bits) + CTT_WORD_BITWIDTH - 1) / CTT_WORD_BITWIDTH)

typedef struct ctt_threadpool ctt_threadpool; /* opaque, core/threadpool.h:21 */

typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(255)]; } big255;
typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(254)]; } big254;

typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(255)]; } bls12_381_fr;
typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(381)]; } bls12_381_fp;
typedef struct { bls12_381_fp c[2]; } bls12_381_fp2;
typedef struct { bls12_381_fp x, y; } bls12_381_g1_aff;
typedef struct { bls12_381_fp x, y, z; } bls12_381_g1_jac;
typedef struct { bls12_381_fp x, y, z; } bls12_381_g1_prj;
typedef struct { bls12_381_fp2 x, y; } bls12_381_g2_aff;
typedef struct { bls12_381_fp2 x, y, z; } bls12_381_g2_jac;
typedef struct { bls12_381_fp2 x, y, z; } bls12_381_g2_prj;

typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(254)]; } bn254_snarks_fr;
typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(254)]; } bn254_snarks_fp;
typedef struct { bn254_snarks_fp c[2]; } bn254_snarks_fp2;
typedef struct { bn254_snarks_fp x, y; } bn254_snarks_g1_aff;
typedef struct { bn254_snarks_fp x, y, z; } bn254_snarks_g1_jac;
typedef struct { bn254_snarks_fp x, y, z; } bn254_snarks_g1_prj;
typedef struct { bn254_snarks_fp2 x, y; } bn254_snarks_g2_aff;
typedef struct { bn254_snarks_fp2 x, y, z; } bn254_snarks_g2_jac;
typedef struct { bn254_snarks_fp2 x, y, z; } bn254_snarks_g2_prj;

typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(255)]; } pallas_fr;
typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(255)]; } pallas_fp;
typedef struct { pallas_fp x, y; } pallas_ec_aff;
typedef struct { pallas_fp x, y, z; } pallas_ec_jac;
typedef struct { pallas_fp x, y, z; } pallas_ec_prj;

typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(255)]; } vesta_fr;
typedef struct { secret_word limbs[CTT_WORDS_REQUIRED(255)]; } vesta_fp;
typedef struct { vesta_fp x, y; } vesta_ec_aff;
typedef struct { vesta_fp x, y, z; } vesta_ec_jac;
typedef struct { vesta_fp x, y, z; } vesta_ec_prj;
#endif /* CTT_MSM_HIP_NO_TYPES */

/* ---- Part 1: Constantine-compatible MSM symbols -------------------------------------------------------- */
#define CTT_MSM_DECL_SERIAL(EC, AFF, BIG, FR)                                                                      \
  void ctt_##EC##_multi_scalar_mul_big_coefs_vartime(EC* r, const BIG coefs[], const AFF points[], size_t len);    \
  void ctt_##EC##_multi_scalar_mul_fr_coefs_vartime(EC* r, const FR coefs[], const AFF points[], size_t len);
#define CTT_MSM_DECL_PARALLEL(EC, AFF, BIG, FR)                                                                    \
  void ctt_##EC##_multi_scalar_mul_big_coefs_vartime_parallel(const ctt_threadpool* tp, EC* r, const BIG coefs[],  \
                                                              const AFF points[], size_t len);                     \
  void ctt_##EC##_multi_scalar_mul_fr_coefs_vartime_parallel(const ctt_threadpool* tp, EC* r, const FR coefs[],    \
                                                             const AFF points[], size_t len);

/* bls12_381.h:163-164,184-185 ; bls12_381_parallel.h:20-23 */
CTT_MSM_DECL_SERIAL(bls12_381_g1_jac, bls12_381_g1_aff, big255, bls12_381_fr)
CTT_MSM_DECL_SERIAL(bls12_381_g1_prj, bls12_381_g1_aff, big255, bls12_381_fr)
CTT_MSM_DECL_PARALLEL(bls12_381_g1_jac, bls12_381_g1_aff, big255, bls12_381_fr)
CTT_MSM_DECL_PARALLEL(bls12_381_g1_prj, bls12_381_g1_aff, big255, bls12_381_fr)
/* bls12_381.h:212-213,233-234 (G2 has no parallel variant upstream) */
CTT_MSM_DECL_SERIAL(bls12_381_g2_jac, bls12_381_g2_aff, big255, bls12_381_fr)
CTT_MSM_DECL_SERIAL(bls12_381_g2_prj, bls12_381_g2_aff, big255, bls12_381_fr)
/* bn254_snarks.h:163-164,184-185,212-213,233-234 ; bn254_snarks_parallel.h:20-23
 * ctt_bn254_snarks_g1_prj_multi_scalar_mul_fr_coefs_vartime_parallel is the Halo2-ZAL entry
 * (constantine-rust/constantine-halo2-zal/src/lib.rs:42-58). */
CTT_MSM_DECL_SERIAL(bn254_snarks_g1_jac, bn254_snarks_g1_aff, big254, bn254_snarks_fr)
CTT_MSM_DECL_SERIAL(bn254_snarks_g1_prj, bn254_snarks_g1_aff, big254, bn254_snarks_fr)
CTT_MSM_DECL_PARALLEL(bn254_snarks_g1_jac, bn254_snarks_g1_aff, big254, bn254_snarks_fr)
CTT_MSM_DECL_PARALLEL(bn254_snarks_g1_prj, bn254_snarks_g1_aff, big254, bn254_snarks_fr)
CTT_MSM_DECL_SERIAL(bn254_snarks_g2_jac, bn254_snarks_g2_aff, big254, bn254_snarks_fr)
CTT_MSM_DECL_SERIAL(bn254_snarks_g2_prj, bn254_snarks_g2_aff, big254, bn254_snarks_fr)
/* pallas.h:125-126,146-147 ; pallas_parallel.h:20-23 */
CTT_MSM_DECL_SERIAL(pallas_ec_jac, pallas_ec_aff, big255, pallas_fr)
CTT_MSM_DECL_SERIAL(pallas_ec_prj, pallas_ec_aff, big255, pallas_fr)
CTT_MSM_DECL_PARALLEL(pallas_ec_jac, pallas_ec_aff, big255, pallas_fr)
CTT_MSM_DECL_PARALLEL(pallas_ec_prj, pallas_ec_aff, big255, pallas_fr)
/* vesta.h:125-126,146-147 ; vesta_parallel.h:20-23 */
CTT_MSM_DECL_SERIAL(vesta_ec_jac, vesta_ec_aff, big255, vesta_fr)
CTT_MSM_DECL_SERIAL(vesta_ec_prj, vesta_ec_aff, big255, vesta_fr)
CTT_MSM_DECL_PARALLEL(vesta_ec_jac, vesta_ec_aff, big255, vesta_fr)
CTT_MSM_DECL_PARALLEL(vesta_ec_prj, vesta_ec_aff, big255, vesta_fr)

/* ---- Part 1b: Constantine-compatible batch conversion to affine (next row) -----------
 * bindings/c_curve_decls.nim:395-396 (`ctt_<EC>_batch_affine`); bls12_381.h:158,179,207,228, bn254_snarks.h:158,179,
 * 207,228, pallas.h:120,141, vesta.h:120,141.  dst[i] = affine(src[i]); a neutral input gives (0,0).  Host pointers. */
#define CTT_BATCH_AFFINE_DECL(EC, AFF) void ctt_##EC##_batch_affine(AFF dst[], const EC src[], size_t n);
CTT_BATCH_AFFINE_DECL(bls12_381_g1_jac, bls12_381_g1_aff)
CTT_BATCH_AFFINE_DECL(bls12_381_g1_prj, bls12_381_g1_aff)
CTT_BATCH_AFFINE_DECL(bls12_381_g2_jac, bls12_381_g2_aff)
CTT_BATCH_AFFINE_DECL(bls12_381_g2_prj, bls12_381_g2_aff)
CTT_BATCH_AFFINE_DECL(bn254_snarks_g1_jac, bn254_snarks_g1_aff)
CTT_BATCH_AFFINE_DECL(bn254_snarks_g1_prj, bn254_snarks_g1_aff)
CTT_BATCH_AFFINE_DECL(bn254_snarks_g2_jac, bn254_snarks_g2_aff)
CTT_BATCH_AFFINE_DECL(bn254_snarks_g2_prj, bn254_snarks_g2_aff)
CTT_BATCH_AFFINE_DECL(pallas_ec_jac, pallas_ec_aff)
CTT_BATCH_AFFINE_DECL(pallas_ec_prj, pallas_ec_aff)
CTT_BATCH_AFFINE_DECL(vesta_ec_jac, vesta_ec_aff)
CTT_BATCH_AFFINE_DECL(vesta_ec_prj, vesta_ec_aff)

/* ---- Part 1c: neutral host-pointer symbols (SURVEY §8b "replacement plan") ----------------------------------------
 * The same host-pointer MSM as Part 1 under names nothing in Constantine exports, for a binding INSIDE libconstantine: the
 * templates of bindings/c_curve_decls.nim:418-431 and bindings/c_curve_decls_parallel.nim:33-45 keep `libExport`ing the
 * ctt_<EC>_multi_scalar_mul_* names and `importc` these (INTEGRATION.md part B), so a static libconstantine.a defines every
 * name once.  Same argument meaning as Part 1 (host pointers, caller-owned, r out-only); the difference is the `int` they
 * return, the error channel the Constantine names lack: 0 = r holds the result; -1 = bad id, len above 2^31-1, or both
 * in-flight slots of the default context taken by tickets of Part 2; -2 = out of device memory.  r is untouched on an error
 * and the binding runs the reference's CPU path instead (also -1, never an abort: no usable HIP device, or a HIP runtime call
 * failed -- ctt_hip_last_error() tells the cases apart).  ctt_hip_msm_available(): 1 when a HIP device is present (the probe
 * a binding makes once), else 0.
 *
 * Why a call was refused -- per thread, set by every symbol of this header that returns an error value (NULL, -1, -2, or Part 3's
 * CTT_HIP_STATUS_GPU_UNAVAILABLE), and cleared when such a symbol is entered (round 6: no stale code of an earlier call):
 *    0 none   -1 refused (bad arguments: curve id, kind, length, ticket, option key, owner)   -2 out of device memory
 *   -5 busy: all in-flight slots of the context are taken -- finish a ticket, or retry (round 6; was -1)
 *   -3 no usable HIP device   -4 a HIP runtime call failed: the context it happened on is LOST -- every later call on it is
 *      refused with -4; destroy it and create a new one (the default context of the host-pointer symbols stays lost) */
int ctt_hip_last_error(void);
const char* ctt_hip_last_error_message(void);   /* the same, in words; valid until the thread's next refused call */
void ctt_hip_clear_last_error(void);
enum { CTT_HIP_BLS12_381_G1 = 0, CTT_HIP_BLS12_381_G2 = 1, CTT_HIP_BN254_SNARKS_G1 = 2,
       CTT_HIP_BN254_SNARKS_G2 = 3, CTT_HIP_PALLAS = 4, CTT_HIP_VESTA = 5 };
enum { CTT_HIP_COEF_BIG = 0, CTT_HIP_COEF_FR = 1 };
enum { CTT_HIP_OUT_AFF = 0, CTT_HIP_OUT_JAC = 1, CTT_HIP_OUT_PRJ = 2 };
int ctt_hip_msm_available(void);
int ctt_hip_msm_host(int curve, int coef_kind, int out_kind, void* r, const void* coefs, const void* points, size_t len);
#define CTT_HIP_MSM_DECL_NEUTRAL(STEM, AFF, BIG, FR)                                                        \
  int ctt_hip_msm_##STEM##_jac_big(STEM##_jac* r, const BIG coefs[], const AFF points[], size_t len);       \
  int ctt_hip_msm_##STEM##_jac_fr(STEM##_jac* r, const FR coefs[], const AFF points[], size_t len);         \
  int ctt_hip_msm_##STEM##_prj_big(STEM##_prj* r, const BIG coefs[], const AFF points[], size_t len);       \
  int ctt_hip_msm_##STEM##_prj_fr(STEM##_prj* r, const FR coefs[], const AFF points[], size_t len);
CTT_HIP_MSM_DECL_NEUTRAL(bls12_381_g1, bls12_381_g1_aff, big255, bls12_381_fr)
CTT_HIP_MSM_DECL_NEUTRAL(bls12_381_g2, bls12_381_g2_aff, big255, bls12_381_fr)
CTT_HIP_MSM_DECL_NEUTRAL(bn254_snarks_g1, bn254_snarks_g1_aff, big254, bn254_snarks_fr)
CTT_HIP_MSM_DECL_NEUTRAL(bn254_snarks_g2, bn254_snarks_g2_aff, big254, bn254_snarks_fr)
CTT_HIP_MSM_DECL_NEUTRAL(pallas_ec, pallas_ec_aff, big255, pallas_fr)
CTT_HIP_MSM_DECL_NEUTRAL(vesta_ec, vesta_ec_aff, big255, vesta_fr)

/* ---- Part 2: device-resident interface ------------------------------------------------------------------ */
typedef struct ctt_hip_msm_ctx ctt_hip_msm_ctx;


int ctt_hip_msm_abi_version(void);
/* One context = one GPU, its streams, one grow-only workspace. NULL ctx in the calls below = process default
 * context on device $CTT_HIP_DEVICE (default 0). */
ctt_hip_msm_ctx* ctt_hip_msm_ctx_create(int device);
void ctt_hip_msm_ctx_destroy(ctt_hip_msm_ctx* ctx);
/* key: "c" window bits, "K" sorted entries per accumulate lane, "S" scalars per sort-partition workgroup, "chunks" slices a
 * host-pointer call is uploaded in (the upload of slice i+1 runs underneath the accumulation of slice i),
 * "horner_bits" bits per group of the bit Horner the device runs per window (0 = 4; the host joins the groups),
 * "host_window_sums" (legacy spelling: 1 = groups of one bit, 2 = one group per window), "timings" 1 = record the
 * stage events ctt_hip_msm_last_timings reads (2 = the accumulate stage and the total only), "timings_every" k = only every
 * k-th MSM records them (the others report zeros; default 1).  value 0 = automatic / off.  Returns 0, or -1 for an unknown key.
 * A context created with $CTT_HIP_CU_TAIL = r > 0 partitions the chip: its tail stream runs on r compute units of every XCD
 * (hipExtStreamCreateWithCUMask), its main stream on the others -- an experiment of round 6, measured slower than sharing the chip
 * (profiles/cu_mask_r06.txt); such streams are BLOCKING streams: do not order them behind the legacy null stream. */
int ctt_hip_msm_set_option(ctt_hip_msm_ctx* ctx, const char* key, int value);
/* r (HOST memory, `out_kind` layout) = sum coefs[i] * points[i]; d_coefs / d_points are DEVICE pointers
 * (BigInt canonical or Fr Montgomery 32-byte scalars; affine Montgomery points, C-API struct layout).
 * Returns 0; -1 for a bad curve id, len > 2^31-1, or three tickets outstanding on the curve.  Blocks until r is written. */
int ctt_hip_msm_device(ctt_hip_msm_ctx* ctx, int curve, int coef_kind, int out_kind, void* r, const void* d_coefs,
                       const void* d_points, size_t len);
/* Split form: submit enqueues the GPU work of one MSM and returns a ticket (>= 0) at once, or -1 on bad arguments or
 * when three tickets are outstanding on the curve (two in rounds 1-5); finish waits for the ticket, runs the host tail (Horner over
 * windows, affine normalisation) and writes r (0, or -1 for a ticket that is not outstanding).  Submitting MSM i+1 before
 * finishing MSM i overlaps the host tail of i with the GPU work of i+1 (how bench.py keeps the GPU busy); a caller of SMALL MSMs
 * (up to ~2^16 pairs: the host tail and the enqueueing are a third of a step there; above, the third MSM only crowds the second one's tail)
 * keeps three outstanding -- submit i+2, then finish i. */
int ctt_hip_msm_device_submit(ctt_hip_msm_ctx* ctx, int curve, int coef_kind, const void* d_coefs, const void* d_points,
                              size_t len);
int ctt_hip_msm_device_finish(ctt_hip_msm_ctx* ctx, int ticket, int out_kind, void* r);
/* Wait for everything enqueued on the context's streams (the main one and the tail stream that carries the last,
 * latency-bound reduction passes of an MSM underneath the next MSM's conversion and sort). */
void ctt_hip_msm_sync(ctt_hip_msm_ctx* ctx);
/* Stream ordering: the engine's streams are not ordered against the caller's.  When device inputs may still be in
 * flight on `producer` (a hipStream_t), call this first: what the engine enqueues afterwards waits for the work
 * `producer` holds now (a stream of the context's own device).  Device outputs (ctt_hip_batch_affine on device arrays) are
 * complete when their call returns. */
int ctt_hip_msm_wait_stream(ctt_hip_msm_ctx* ctx, void* producer);
/* Multi-GPU: the host-pointer symbols of Part 1 shard a call by points over these devices -- the reference's msm-level
 * split (ec_multi_scalar_mul_parallel.nim:386-431; balanced chunks, threadpool/partitioners.nim:44-77) with GPUs for
 * threads: each slice is uploaded to its own GPU and the partial results are summed on the host.  `n` device ids (an
 * id may repeat: one context each); n < 2 turns sharding off.  Default: $CTT_HIP_DEVICES ("0,1,2,3" or "all"), else
 * off.  Calls below shard_min pairs per device (default 2^15, $CTT_HIP_SHARD_MIN) use fewer devices.
 * Returns 0, or -1 for a device id out of range. */
int ctt_hip_msm_set_devices(const int* devices, int n);
void ctt_hip_msm_set_shard_min(size_t pairs_per_device);
/* Cached bases -- the Halo2-ZAL descriptor hooks (constantine-halo2-zal/src/lib.rs:60-95: get_base_descriptor,
 * msm_with_cached_base): base points are uploaded and converted to the device representation once and stay resident
 * in HBM; later MSMs move only the 32-byte coefficients. `points` / `coefs` are host pointers when the
 * *_on_device flag is 0, device pointers when it is 1. ctt_hip_msm_with_bases uses the first `len` bases and must be
 * called with the context the bases were created on (-1 otherwise: the records live on that context's GPU); destroy the
 * bases before their context. */
typedef struct ctt_hip_msm_bases ctt_hip_msm_bases;
ctt_hip_msm_bases* ctt_hip_msm_bases_create(ctt_hip_msm_ctx* ctx, int curve, const void* points, size_t len,
                                            int points_on_device);
void ctt_hip_msm_bases_destroy(ctt_hip_msm_ctx* ctx, ctt_hip_msm_bases* bases);
int ctt_hip_msm_with_bases(ctt_hip_msm_ctx* ctx, const ctt_hip_msm_bases* bases, int coef_kind, int out_kind, void* r,
                           const void* coefs, size_t len, int coefs_on_device);
/* split form (device-resident coefficients): a ticket for ctt_hip_msm_device_finish, or -1 */
int ctt_hip_msm_with_bases_submit(ctt_hip_msm_ctx* ctx, const ctt_hip_msm_bases* bases, int coef_kind, const void* d_coefs,
                                  size_t len);
/* Cached bases WITH A WINDOW TABLE: besides the bases themselves their multiples 2^(c*w) * P, w = 0 .. bits/c, are computed
 * once and stay resident ((bits/c + 1) x the memory: 1.7 GB for 2^20 BLS12-381 G1 bases at c = 20).  Every Booth digit
 * of every window then selects a table row, all windows share one bucket set, and the window combine (the doublings of
 * ec_multi_scalar_mul.nim:250-254) disappears: a later ctt_hip_msm_with_bases[_submit] call costs bits/c + 1 accumulations
 * per pair with a larger c than the table-less MSM can afford (13 instead of 16 at 2^20 pairs) and ONE bucket reduction.
 * The result is the same group element.  window_bits = 0 chooses c from len; ctt_hip_msm_bases_window_bits returns the c
 * in use (0 = plain records: no c fits the sort's packed records for this len). */
ctt_hip_msm_bases* ctt_hip_msm_bases_create_table(ctt_hip_msm_ctx* ctx, int curve, const void* points, size_t len,
                                                  int points_on_device, int window_bits);
int ctt_hip_msm_bases_window_bits(const ctt_hip_msm_bases* bases);
/* HIP-event stage times (ms) of the last finished MSM: digits, sort, accumulate, merge, reduce, total.  Opt-in: set the
 * option "timings" to 1 first (the events are host time per MSM; without it the call returns zeros). */
int ctt_hip_msm_last_timings(ctt_hip_msm_ctx* ctx, float* ms, int cap);
/* plan of the last call: c, W, K, G, S, resident lanes */
int ctt_hip_msm_last_plan(ctt_hip_msm_ctx* ctx, int* out, int cap);
/* d_out[i] = [s_i]G, deterministic synthetic subgroup points (bench / test inputs) */
int ctt_hip_gen_points(ctt_hip_msm_ctx* ctx, int curve, uint64_t seed, uint64_t first, uint32_t n, void* d_out);
/* element-wise coordinate-field op on device arrays (0 mul, 1 sqr, 2 add, 3 sub, 4 neg): kernel unit tests */
int ctt_hip_field_op(ctt_hip_msm_ctx* ctx, int curve, int op, const void* d_a, const void* d_b, void* d_r, uint32_t n);
/* host-only: r (`out_kind` layout) = sum of n affine points -- combines the per-GPU partial results of a
 * sharded MSM (the `r ~+= partial` of ec_multi_scalar_mul_parallel.nim:427-429). Needs no GPU. */
int ctt_hip_ec_sum_affine(int curve, int out_kind, void* r, const void* pts_aff, size_t n);
/* r (host, `out_kind` layout) = sum of `len` affine points -- sum_reduce_vartime
 * (ec_shortweierstrass_batch_ops.nim:649-663) on the GPU; points on the host (points_on_device = 0) or in HBM (1). */
int ctt_hip_sum_reduce(ctt_hip_msm_ctx* ctx, int curve, int out_kind, void* r, const void* points, size_t len,
                       int points_on_device);
/* dst[i] = affine(src[i]), src_kind CTT_HIP_OUT_JAC (x = X/Z^2, y = Y/Z^3) or CTT_HIP_OUT_PRJ (x = X/Z, y = Y/Z) --
 * batchAffine(_vartime) (ec_shortweierstrass_batch_ops.nim:44-345).  Both arrays on the host (on_device = 0) or both
 * in HBM (1).  Blocking. */
int ctt_hip_batch_affine(ctt_hip_msm_ctx* ctx, int curve, int src_kind, void* dst, const void* src, size_t n, int on_device);
/* ok[i] = 1 when points[i] has order r (r = the curve order; the neutral passes): the subgroup check that the MSM's callers
 * run on deserialised points (eth_evm_bls12381_g1msm / g2msm, ethereum_evm_precompiles.nim:894-975; KZG commitments), for all n
 * points at once.  BLS12-381: the reference's endomorphism tests (isInSubgroup, named/constants/bls12_381_subgroups.nim:170-207);
 * other curves: [r]P = neutral.  Up to 64 host-resident points (256 for BLS12-381) are checked on host threads, more -- or
 * device-resident ones -- in one launch.  points: affine, host (points_on_device = 0) or device memory; ok: n bytes of host memory. */
int ctt_hip_subgroup_check(ctt_hip_msm_ctx* ctx, int curve, uint8_t* ok, const void* points, size_t n, int points_on_device);
/* Quotient polynomial of a KZG opening over the scalar field of `curve`, in evaluation form: the Fr-side work of kzg_prove
 * (constantine/commitments/kzg.nim:204-223 -> getQuotientPoly).  d_poly: n canonical scalars (the MSM's coefficient format),
 * d_domain: the n roots of unity as Montgomery residues, both in DEVICE memory; z: the opening point, canonical, host.  Writes
 * q_i = (p_i - y) / (w_i - z), canonical, to d_q (device: the coefficients of the proof's MSM) and y = p(z), canonical, to y
 * (host).  Blocking.  Returns 0; -2 when z is one of the roots of unity (the reference's other branch, left to the caller);
 * -1 for bad arguments or when device memory runs out. */
int ctt_hip_fr_quotient(ctt_hip_msm_ctx* ctx, int curve, void* d_q, void* y, const void* d_poly, const void* d_domain,
                        const void* z, uint32_t n);
/* the engine's hipStream_t */
void* ctt_hip_msm_stream(ctt_hip_msm_ctx* ctx);

/* ---- Part 3: the MSM's immediate callers under the reference's own names -----------------------
 * EIP-4844 KZG commitments and opening proofs over a context that caches the Lagrange SRS on the GPU, and the EIP-2537
 * BLS12_G1MSM / BLS12_G2MSM precompiles: same symbols, argument meaning and status enums as
 *   include/constantine/protocols/ethereum_eip4844_kzg.h:106 (blob_to_kzg_commitment), :126 (compute_kzg_proof),
 *       :153 (compute_blob_kzg_proof), :200 (context_new), :238 (context_delete)
 *   include/constantine/protocols/ethereum_evm_precompiles.h:386 (g1msm), :419 (g2msm)
 * (constantine/ethereum_eip4844_kzg.nim:297-444, constantine/ethereum_evm_precompiles.nim:894-1060).  Host side in C++
 * (constantine_amd/csrc/protocols.hip), MSMs / subgroup checks / quotient polynomial on the GPU.  Verification (pairings), the
 * PeerDAS cell functions and the other precompiles are out of scope and not exported.  There is no CPU fallback, and none of
 * these symbols terminates the process (round 5; rounds 1-4 aborted): a call the GPU cannot serve -- no device, out of device
 * memory, a failed HIP call -- returns CTT_HIP_STATUS_GPU_UNAVAILABLE, a value outside every status enum below (the reference's
 * *_status_to_string print "InvalidStatusCode" for it), leaves its outputs untouched, and ctt_hip_last_error() of the calling
 * thread says why.  A precompile whose context is momentarily held by another thread's tickets waits for it (up to ~2 s). */
#define CTT_HIP_STATUS_GPU_UNAVAILABLE 0xF0
#ifndef CTT_MSM_HIP_NO_PROTOCOLS
typedef uint8_t ctt_byte;   /* the reference's `byte` (constantine/core/datatypes.h) */
typedef struct ctt_eth_kzg_context_struct ctt_eth_kzg_context;
typedef struct { ctt_byte raw[48]; }        ctt_eth_kzg_commitment;
typedef struct { ctt_byte raw[48]; }        ctt_eth_kzg_proof;
typedef struct { ctt_byte raw[4096 * 32]; } ctt_eth_kzg_blob;
typedef struct { ctt_byte raw[32]; }        ctt_eth_kzg_opening_challenge;
typedef struct { ctt_byte raw[32]; }        ctt_eth_kzg_eval_at_challenge;
typedef enum __attribute__((__packed__)) {
  cttEthKzg_Success, cttEthKzg_VerificationFailure, cttEthKzg_InputsLengthsMismatch, cttEthKzg_ScalarZero,
  cttEthKzg_ScalarLargerThanCurveOrder, cttEthKzg_EccInvalidEncoding, cttEthKzg_EccCoordinateGreaterThanOrEqualModulus,
  cttEthKzg_EccPointNotOnCurve, cttEthKzg_EccPointNotInSubgroup, cttEthKzg_CellIndicesNotAscending,
} ctt_eth_kzg_status;
typedef enum __attribute__((__packed__)) {
  cttEthTS_Success, cttEthTS_MissingOrInaccessibleFile, cttEthTS_InvalidFile
} ctt_eth_trusted_setup_status;
typedef enum __attribute__((__packed__)) { cttEthTSFormat_ckzg4844 } ctt_eth_trusted_setup_format;
typedef enum __attribute__((__packed__)) {
  cttEVM_Success, cttEVM_InvalidInputSize, cttEVM_InvalidOutputSize, cttEVM_IntLargerThanModulus, cttEVM_PointNotOnCurve,
  cttEVM_PointNotInSubgroup, cttEVM_VerificationFailure,
} ctt_evm_status;

/* the c-kzg text format of the Ethereum ceremony; the SRS is cached on $CTT_HIP_DEVICE (default 0) */
ctt_eth_trusted_setup_status ctt_eth_kzg_context_new(ctt_eth_kzg_context** ctx, const char* filepath,
                                                     ctt_eth_trusted_setup_format format);
/* ethereum_eip4844_kzg.h:232 -- t, b size the reference's CPU lookup tables (PeerDAS); accepted, the same context as above */
ctt_eth_trusted_setup_status ctt_eth_kzg_context_new_with_precompute(ctt_eth_kzg_context** ctx, const char* filepath,
                                                                     ctt_eth_trusted_setup_format format, int t, int b);
void ctt_eth_kzg_context_delete(ctt_eth_kzg_context* ctx);
ctt_eth_kzg_status ctt_eth_kzg_blob_to_kzg_commitment(const ctt_eth_kzg_context* ctx, ctt_eth_kzg_commitment* dst,
                                                      const ctt_eth_kzg_blob* blob);
ctt_eth_kzg_status ctt_eth_kzg_compute_kzg_proof(const ctt_eth_kzg_context* ctx, ctt_eth_kzg_proof* proof,
                                                 ctt_eth_kzg_eval_at_challenge* y, const ctt_eth_kzg_blob* blob,
                                                 const ctt_eth_kzg_opening_challenge* z);
ctt_eth_kzg_status ctt_eth_kzg_compute_blob_kzg_proof(const ctt_eth_kzg_context* ctx, ctt_eth_kzg_proof* proof,
                                                      const ctt_eth_kzg_blob* blob, const ctt_eth_kzg_commitment* commitment);
/* include/constantine/protocols/ethereum_eip4844_kzg_parallel.h:40, :61, :73 -- the thread pool is accepted and not used */
ctt_eth_kzg_status ctt_eth_kzg_blob_to_kzg_commitment_parallel(const ctt_threadpool* tp, const ctt_eth_kzg_context* ctx,
                                                               ctt_eth_kzg_commitment* dst, const ctt_eth_kzg_blob* blob);
ctt_eth_kzg_status ctt_eth_kzg_compute_kzg_proof_parallel(const ctt_threadpool* tp, const ctt_eth_kzg_context* ctx,
                                                          ctt_eth_kzg_proof* proof, ctt_eth_kzg_eval_at_challenge* y,
                                                          const ctt_eth_kzg_blob* blob, const ctt_eth_kzg_opening_challenge* z);
ctt_eth_kzg_status ctt_eth_kzg_compute_blob_kzg_proof_parallel(const ctt_threadpool* tp, const ctt_eth_kzg_context* ctx,
                                                               ctt_eth_kzg_proof* proof, const ctt_eth_kzg_blob* blob,
                                                               const ctt_eth_kzg_commitment* commitment);
ctt_evm_status ctt_eth_evm_bls12381_g1msm(ctt_byte* r, size_t r_len, const ctt_byte* inputs, size_t inputs_len);
ctt_evm_status ctt_eth_evm_bls12381_g2msm(ctt_byte* r, size_t r_len, const ctt_byte* inputs, size_t inputs_len);

/* Not in the reference: the context from memory -- the 4096 x 48 bytes of the Lagrange-form G1 SRS in ceremony (file) order --
 * on a chosen GPU, optionally cached as a window table (ctt_hip_msm_bases_create_table).  -> ctt_eth_trusted_setup_status */
int ctt_hip_eth_kzg_context_from_srs(ctt_eth_kzg_context** ctx, const uint8_t* g1_lagrange_compressed, size_t n_points,
                                     int device, int table);
/* Host-only pieces of the above (no GPU): SHA-256; BLS12-381 G1 compressed <-> affine Montgomery {x, y} (C-API layout, no
 * subgroup check; -> ctt_eth_kzg_status); blob -> 4096 canonical little-endian scalars (-> ctt_eth_kzg_status); the Fiat-Shamir
 * challenge of compute_blob_kzg_proof (32 big-endian bytes); the quotient polynomial of an opening on the host, both branches
 * (4096 canonical little-endian scalars in and out, z and y 32 little-endian bytes). */
void ctt_hip_sha256(uint8_t out[32], const uint8_t* data, size_t len);
int ctt_hip_bls12_381_g1_decompress(uint8_t aff[96], const uint8_t in[48]);
void ctt_hip_bls12_381_g1_compress(uint8_t out[48], const uint8_t aff[96]);
int ctt_hip_eth_kzg_blob_to_scalars(uint8_t* scalars_le, const uint8_t* blob);
void ctt_hip_eth_kzg_challenge(uint8_t z_be[32], const uint8_t* blob, const uint8_t commitment[48]);
void ctt_hip_eth_kzg_quotient_host(uint8_t* q_le, uint8_t y_le[32], const uint8_t* poly_le, const uint8_t z_le[32]);
#endif /* CTT_MSM_HIP_NO_PROTOCOLS */

#ifdef __cplusplus
}
#endif
#endif /* CTT_MSM_HIP_H */
