/* tests/c_api/t_msm_c_abi.c -- calls the Constantine-compatible MSM symbols from plain C through
 * include/ctt_msm_hip.h, the way a C user of constantine.h would (the reference's own C tests,
 * tests/c_api/t_threadpool.c and examples-c/, never call MSM).
 *
 * usage: t_msm_c_abi <in.bin> <out.bin>
 *   in.bin : u64 n | n * big255 (32 B) | n * bls12_381_g1_aff (96 B) | u64 m | m bytes of an EIP-2537 BLS12_G1MSM input
 *   out.bin: bls12_381_g1_jac (parallel symbol, big coefs) | bls12_381_g1_prj (serial symbol, big coefs)
 *            | bls12_381_g1_jac (parallel symbol again, the call sharded over two contexts on device 0:
 *              ctt_hip_msm_set_devices) | bls12_381_g1_jac (cached bases with a window table:
 *              ctt_hip_msm_bases_create_table + ctt_hip_msm_with_bases) | bls12_381_g1_jac (neutral typed symbol
 *              ctt_hip_msm_bls12_381_g1_jac_big -- what the Nim binding of INTEGRATION.md part B imports) | bls12_381_g1_prj
 *              (neutral generic symbol ctt_hip_msm_host) | n bytes of ctt_hip_subgroup_check flags | 128 bytes: the output of
 *              ctt_eth_evm_bls12381_g1msm (header part 3: the reference's precompile symbol and status enum from plain C)
 * Built and driven by tests/test_gpu_parity.py::test_c_program_through_the_header. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ctt_msm_hip.h"

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  uint64_t n = 0;
  if (fread(&n, sizeof n, 1, f) != 1) return 4;
  big255* coefs = (big255*)malloc(n * sizeof(big255));
  bls12_381_g1_aff* points = (bls12_381_g1_aff*)malloc(n * sizeof(bls12_381_g1_aff));
  if (fread(coefs, sizeof(big255), n, f) != n) return 5;
  if (fread(points, sizeof(bls12_381_g1_aff), n, f) != n) return 6;
  uint64_t m = 0;
  if (fread(&m, sizeof m, 1, f) != 1) return 17;
  ctt_byte* evm_in = (ctt_byte*)malloc(m ? m : 1);
  if (fread(evm_in, 1, m, f) != m) return 18;
  fclose(f);

  bls12_381_g1_jac rj;
  bls12_381_g1_prj rp;
  const ctt_threadpool* tp = NULL; /* accepted and ignored by the GPU engine */
  ctt_bls12_381_g1_jac_multi_scalar_mul_big_coefs_vartime_parallel(tp, &rj, coefs, points, (size_t)n);
  ctt_bls12_381_g1_prj_multi_scalar_mul_big_coefs_vartime(&rp, coefs, points, (size_t)n);

  /* the same call cut in two slices, one context each (what CTT_HIP_DEVICES=0,1 does on a two-GPU node) */
  bls12_381_g1_jac rs;
  const int devs[2] = {0, 0};
  if (ctt_hip_msm_set_devices(devs, 2) != 0) return 8;
  ctt_hip_msm_set_shard_min(100);
  ctt_bls12_381_g1_jac_multi_scalar_mul_big_coefs_vartime_parallel(tp, &rs, coefs, points, (size_t)n);
  if (ctt_hip_msm_set_devices(devs, 0) != 0) return 9;
  uint8_t* ok = (uint8_t*)malloc(n);
  if (ctt_hip_subgroup_check(NULL, CTT_HIP_BLS12_381_G1, ok, points, (size_t)n, 0) != 0) return 10;

  /* the bases cached once with a window table (the ZAL base descriptor), then an MSM over them */
  bls12_381_g1_jac rt;
  ctt_hip_msm_bases* bases = ctt_hip_msm_bases_create_table(NULL, CTT_HIP_BLS12_381_G1, points, (size_t)n, 0, 0);
  if (!bases || ctt_hip_msm_bases_window_bits(bases) <= 0) return 11;
  if (ctt_hip_msm_with_bases(NULL, bases, CTT_HIP_COEF_BIG, CTT_HIP_OUT_JAC, &rt, coefs, (size_t)n, 0) != 0) return 12;
  ctt_hip_msm_bases_destroy(NULL, bases);

  /* the neutral symbols: same call, status returned (0 = done); a bad curve id is refused, not aborted on */
  bls12_381_g1_jac rn;
  bls12_381_g1_prj rg;
  if (ctt_hip_msm_available() != 1) return 13;
  if (ctt_hip_msm_bls12_381_g1_jac_big(&rn, coefs, points, (size_t)n) != 0) return 14;
  if (ctt_hip_msm_host(CTT_HIP_BLS12_381_G1, CTT_HIP_COEF_BIG, CTT_HIP_OUT_PRJ, &rg, coefs, points, (size_t)n) != 0) return 15;
  if (ctt_hip_msm_host(17, CTT_HIP_COEF_BIG, CTT_HIP_OUT_PRJ, &rg, coefs, points, (size_t)n) != -1) return 16;

  /* the precompile through the reference's signature: status enum, explicit buffer lengths */
  ctt_byte evm_out[128];
  if (ctt_eth_evm_bls12381_g1msm(evm_out, sizeof evm_out, evm_in, (size_t)m) != cttEVM_Success) return 19;
  if (ctt_eth_evm_bls12381_g1msm(evm_out, 64, evm_in, (size_t)m) != cttEVM_InvalidOutputSize) return 20;
  if (ctt_eth_evm_bls12381_g1msm(evm_out, sizeof evm_out, evm_in, (size_t)m - 1) != cttEVM_InvalidInputSize) return 21;
  _Static_assert(sizeof(ctt_evm_status) == 1 && sizeof(ctt_eth_kzg_status) == 1 && sizeof(ctt_eth_kzg_blob) == 4096 * 32, "reference ABI");

  f = fopen(argv[2], "wb");
  if (!f) return 7;
  fwrite(&rj, sizeof rj, 1, f);
  fwrite(&rp, sizeof rp, 1, f);
  fwrite(&rs, sizeof rs, 1, f);
  fwrite(&rt, sizeof rt, 1, f);
  fwrite(&rn, sizeof rn, 1, f);
  fwrite(&rg, sizeof rg, 1, f);
  fwrite(ok, 1, n, f);
  fwrite(evm_out, 1, sizeof evm_out, f);
  fclose(f);
  free(evm_in);
  free(ok);
  free(coefs);
  free(points);
  return 0;
}
