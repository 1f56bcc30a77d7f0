// generators.hip.h -- the fixed generators of G1 and G2 as device constants (g1.rs:197-217, g2.rs:210-250).
#pragma once
#include "curve.hip.h"

namespace bls {

template <class F> DEV Aff<F> generator();
template <> DEV Aff<FpPolicy> generator<FpPolicy>() {
  constexpr PLimbs gx = {BLS_G1_GEN_X}, gy = {BLS_G1_GEN_Y};
  Aff<FpPolicy> g; g.x = fe1_const(gx); g.y = fe1_const(gy); return g;
}
template <> DEV Aff<Fp2Policy> generator<Fp2Policy>() {
  constexpr PLimbs x0 = {BLS_G2_GEN_X0}, x1 = {BLS_G2_GEN_X1}, y0 = {BLS_G2_GEN_Y0}, y1 = {BLS_G2_GEN_Y1};
  Aff<Fp2Policy> g; g.x.c0 = fe1_const(x0); g.x.c1 = fe1_const(x1); g.y.c0 = fe1_const(y0); g.y.c1 = fe1_const(y1); return g;
}

}  // namespace bls
