/* tests/c_api/t_zal_cached_base.c -- the Halo2-ZAL caching hooks over the cached-base interface, in plain C.
 *
 * Upstream (constantine-rust/constantine-halo2-zal/src/lib.rs:60-95) get_base_descriptor and msm_with_cached_base are
 * pass-throughs with a "do expensive device/library specific preprocessing here" note.  INTEGRATION.md part D2 writes out the
 * Rust bodies that put the preprocessing there; this program is the same three functions in C -- descriptor creation
 * (ctt_hip_msm_bases_create_table), the MSM against it (ctt_hip_msm_with_bases, Fr coefficients in host memory, projective
 * result: exactly what CttEngine::msm hands to ctt_bn254_snarks_g1_prj_multi_scalar_mul_fr_coefs_vartime_parallel), the drop
 * (ctt_hip_msm_bases_destroy) -- with ONE descriptor reused across several coefficient vectors, the prover's access pattern.
 *
 * usage: t_zal_cached_base <in.bin> <out.bin>
 *   in.bin : u64 n | u64 m | n * bn254_snarks_g1_aff (64 B) | m vectors of n * bn254_snarks_fr (32 B, Montgomery)
 *   out.bin: m * bn254_snarks_g1_prj from msm_with_cached_base | m * bn254_snarks_g1_prj from the un-cached ZAL entry (msm)
 *            | bn254_snarks_g1_prj: a prefix (n / 2 pairs) of the cached bases | i32 window bits of the descriptor
 * Built and driven by tests/test_gpu_parity.py::test_zal_cached_base_descriptor_from_c. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ctt_msm_hip.h"

/* ---- the three trait functions (lib.rs:68-71, :82-88 and the descriptor's Drop) ------------------------------------------- */
typedef struct {
  const bn254_snarks_g1_aff* raw; /* upstream's only field: kept for msm_with_cached_scalars and as the CPU path's input */
  size_t len;
  ctt_hip_msm_bases* dev;         /* the bases resident in HBM as a window table; NULL when the GPU refused */
} zal_base_desc;

static zal_base_desc zal_get_base_descriptor(const bn254_snarks_g1_aff* base, size_t len) {
  zal_base_desc d = {base, len, NULL};
  if (ctt_hip_msm_available())
    d.dev = ctt_hip_msm_bases_create_table(NULL, CTT_HIP_BN254_SNARKS_G1, base, len, /*points_on_device=*/0, /*window_bits=*/0);
  return d; /* d.dev == NULL: no device or out of device memory (ctt_hip_last_error()) -- the caller keeps its CPU path */
}

/* upstream's msm(): the un-cached entry */
static void zal_msm(bn254_snarks_g1_prj* r, const bn254_snarks_fr* coeffs, const bn254_snarks_g1_aff* base, size_t len) {
  ctt_bn254_snarks_g1_prj_multi_scalar_mul_fr_coefs_vartime_parallel(NULL, r, coeffs, base, len);
}

static void zal_msm_with_cached_base(bn254_snarks_g1_prj* r, const bn254_snarks_fr* coeffs, size_t len, const zal_base_desc* base) {
  if (base->dev && ctt_hip_msm_with_bases(NULL, base->dev, CTT_HIP_COEF_FR, CTT_HIP_OUT_PRJ, r, coeffs, len, /*coefs_on_device=*/0) == 0)
    return;
  zal_msm(r, coeffs, base->raw, len); /* refused (r untouched): upstream's pass-through */
}

static void zal_drop_base_descriptor(zal_base_desc* d) {
  if (d->dev) ctt_hip_msm_bases_destroy(NULL, d->dev);
  d->dev = NULL;
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  uint64_t n = 0, m = 0;
  if (fread(&n, sizeof n, 1, f) != 1 || fread(&m, sizeof m, 1, f) != 1 || n == 0 || m == 0) return 4;
  bn254_snarks_g1_aff* bases = (bn254_snarks_g1_aff*)malloc(n * sizeof *bases);
  bn254_snarks_fr* coeffs = (bn254_snarks_fr*)malloc(n * m * sizeof *coeffs);
  if (fread(bases, sizeof *bases, n, f) != n) return 5;
  if (fread(coeffs, sizeof *coeffs, n * m, f) != n * m) return 6;
  fclose(f);

  zal_base_desc desc = zal_get_base_descriptor(bases, (size_t)n);
  if (!desc.dev) {
    fprintf(stderr, "no descriptor: %d %s\n", ctt_hip_last_error(), ctt_hip_last_error_message());
    return 7;
  }
  const int32_t wbits = ctt_hip_msm_bases_window_bits(desc.dev);
  bn254_snarks_g1_prj* cached = (bn254_snarks_g1_prj*)malloc(m * sizeof *cached);
  bn254_snarks_g1_prj* plain = (bn254_snarks_g1_prj*)malloc(m * sizeof *plain);
  for (uint64_t v = 0; v < m; v++) zal_msm_with_cached_base(&cached[v], coeffs + v * n, (size_t)n, &desc); /* one descriptor, m proofs' worth of MSMs */
  for (uint64_t v = 0; v < m; v++) zal_msm(&plain[v], coeffs + v * n, bases, (size_t)n);
  bn254_snarks_g1_prj prefix;
  zal_msm_with_cached_base(&prefix, coeffs, (size_t)(n / 2), &desc); /* fewer coefficients than cached bases: the first len bases */
  /* a descriptor made on another context is refused, not crashed on (-1, r untouched) */
  ctt_hip_msm_ctx* other = ctt_hip_msm_ctx_create(0);
  bn254_snarks_g1_prj untouched = cached[0];
  if (!other || ctt_hip_msm_with_bases(other, desc.dev, CTT_HIP_COEF_FR, CTT_HIP_OUT_PRJ, &untouched, coeffs, (size_t)n, 0) != -1) return 8;
  ctt_hip_msm_ctx_destroy(other);
  zal_drop_base_descriptor(&desc);

  f = fopen(argv[2], "wb");
  if (!f) return 9;
  fwrite(cached, sizeof *cached, m, f);
  fwrite(plain, sizeof *plain, m, f);
  fwrite(&prefix, sizeof prefix, 1, f);
  fwrite(&wbits, sizeof wbits, 1, f);
  fclose(f);
  return 0;
}
