// generators.h -- subgroup generators (Montgomery form) for the synthetic-input generator.
// constantine/named/constants/bls12_381_generators.nim:20-36, bn254_snarks_generators.nim:23-45;
// Pallas/Vesta use (-1, 2).
#pragma once
#include "msm_bodies.h"

namespace ctt {

template <class F, class G>
CTT_HD F gen_coord_fp(int which) {
  F r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.l[i] = which == 0 ? G::C0[i] : which == 1 ? G::C1[i] : 0u;
  return r;
}
template <class F, class G>
CTT_HD F gen_coord_fp4(int which) {
  F r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.l[i] = which == 0 ? G::C0[i] : which == 1 ? G::C1[i] : which == 2 ? G::C2[i] : G::C3[i];
  return r;
}

template <class C> CTT_HD Affine<typename C::F> generator();

template <> CTT_HD Affine<Bls12381G1::F> generator<Bls12381G1>() {
  using F = Bls12381G1::F;
  return {gen_coord_fp<F, GenBls12381G1>(0), gen_coord_fp<F, GenBls12381G1>(1)};
}
template <> CTT_HD Affine<Bn254G1::F> generator<Bn254G1>() {
  using F = Bn254G1::F;
  return {gen_coord_fp<F, GenBn254G1>(0), gen_coord_fp<F, GenBn254G1>(1)};
}
template <> CTT_HD Affine<PallasEc::F> generator<PallasEc>() {
  using F = PallasEc::F;
  return {gen_coord_fp<F, GenPallas>(0), gen_coord_fp<F, GenPallas>(1)};
}
template <> CTT_HD Affine<VestaEc::F> generator<VestaEc>() {
  using F = VestaEc::F;
  return {gen_coord_fp<F, GenVesta>(0), gen_coord_fp<F, GenVesta>(1)};
}
template <> CTT_HD Affine<Bls12381G2::F> generator<Bls12381G2>() {
  using B = Bls12381G2::F::Base;
  using G = GenBls12381G2;
  return {{gen_coord_fp4<B, G>(0), gen_coord_fp4<B, G>(1)}, {gen_coord_fp4<B, G>(2), gen_coord_fp4<B, G>(3)}};
}
template <> CTT_HD Affine<Bn254G2::F> generator<Bn254G2>() {
  using B = Bn254G2::F::Base;
  using G = GenBn254G2;
  return {{gen_coord_fp4<B, G>(0), gen_coord_fp4<B, G>(1)}, {gen_coord_fp4<B, G>(2), gen_coord_fp4<B, G>(3)}};
}

}  // namespace ctt
