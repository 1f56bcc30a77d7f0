// Exercises include/bls12_381.hpp (the C++ mirror of the reference API) against properties the reference's own
// tests use (src/pairings.rs:835-921, src/g1.rs:1558-1595).  Argument 1: 576-byte file with the RELIC Gt constant.
#include <cstdio>
#include <cstdlib>
#include "bls12_381.hpp"
using namespace bls;
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  Gt want;
  if (argc > 1) { FILE* f = std::fopen(argv[1], "rb"); REQUIRE(f && std::fread(want.f.data(), 1, 576, f) == 576); std::fclose(f); }
  Gt g = pairing(G1Affine::generator(), G2Affine::generator());
  if (argc > 1) REQUIRE(g == want);                                  // RELIC KAT, src/tests/mod.rs:78-231
  REQUIRE(Gt::generator() == g);
  Scalar a = Scalar::from_u64(0x1234567), b = Scalar::from_u64(0x7654321), ab = Scalar::from_u64(0x1234567ull * 0x7654321ull);
  G1Affine ga = (G1Affine::generator() * a).to_affine();
  G2Affine hb = (G2Affine::generator() * b).to_affine();
  Gt p = pairing(ga, hb);
  REQUIRE(p == pairing((G1Affine::generator() * ab).to_affine(), G2Affine::generator()));   // bilinearity
  REQUIRE(pairing(G1Affine::identity(), hb) == Gt::identity());
  REQUIRE((p + (-p)) == Gt::identity());
  REQUIRE(g * ab == p && g * Scalar::from_u64(2) == g.dbl());          // Gt * Scalar, src/pairings.rs:297-322
  auto ml = multi_miller_loop({{ga, G2Prepared(hb)}, {G1Affine::identity(), G2Prepared(hb)}, {G1Affine::generator(), G2Prepared(G2Affine::generator())}});
  REQUIRE(ml.final_exponentiation() == p + g);
  REQUIRE(MillerLoopResult::default_value().final_exponentiation() == Gt::identity());
  G1Projective gp = G1Projective::generator();
  REQUIRE((gp * a) * b == gp * ab);
  REQUIRE(G1Projective::sum({gp, gp.dbl(), G1Projective::identity()}) == gp * Scalar::from_u64(3));
  REQUIRE(G1Projective::batch_normalize({gp, G1Projective::identity()})[1].is_identity());
  REQUIRE((gp + G1Affine::generator()) == gp.dbl());
  auto m = msm<1>({G1Affine::generator(), ga}, {b, Scalar::from_u64(1)});
  REQUIRE(m == gp * b + G1Projective{(G1Affine::generator() * a).xyz});
  // hash_to_curve: deterministic, in the subgroup ([r]P = O is implied by pairing bilinearity below), encode != hash
  auto h = hash_to_curve<1>({"abc", "abc", ""}, "QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_");
  REQUIRE(h[0] == h[1] && !(h[0] == h[2]));
  auto h2 = hash_to_curve<2>({"abc"}, "QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_");
  REQUIRE(pairing((h[0] * a).to_affine(), h2[0].to_affine()) == pairing(h[0].to_affine(), (h2[0] * a).to_affine()));
  // Fr: x * x^-1 = 1 (R in Montgomery form), NTT round trip
  FrLimbs one = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};     // scalar.rs:159-164
  std::vector<FrLimbs> x = {{5, 6, 7, 8}, one, {1, 0, 0, 0}, {9, 9, 9, 9}};
  auto xi = fr_op(FrOp::Invert, x);
  REQUIRE(fr_op(FrOp::Mul, x, xi)[0] == one && fr_op(FrOp::Mul, x, xi)[3] == one);
  auto y = x; fr_ntt(y); REQUIRE(!(y == x)); fr_ntt(y, true); REQUIRE(y == x);
  // round 4: N equations in one call; the same operations over a device group (two logical members on device 0)
  auto eqs = multi_miller_loop_many({{{ga, G2Prepared(hb)}, {G1Affine::generator(), G2Prepared(G2Affine::generator())}}, {}, {{ga, G2Prepared(hb)}}});
  REQUIRE(eqs.size() == 3 && eqs[0] == p + g && eqs[1] == Gt::identity() && eqs[2] == p);
  {
    Group grp({0, 0});
    REQUIRE(grp.size() == 2);
    REQUIRE(grp.msm<1>({G1Affine::generator(), ga, ga}, {b, Scalar::from_u64(1), Scalar::from_u64(2)}) == gp * b + G1Projective{(G1Affine::generator() * a).xyz} + (G1Projective{(G1Affine::generator() * a).xyz}).dbl());
    REQUIRE(grp.multi_miller_loop_final_exp({{ga, G2Prepared(hb)}, {G1Affine::identity(), G2Prepared(hb)}, {G1Affine::generator(), G2Prepared(G2Affine::generator())}}) == p + g);
    auto pb = grp.pairing_batch({ga, G1Affine::generator(), G1Affine::identity()}, {hb, G2Affine::generator(), hb});
    REQUIRE(pb[0] == p && pb[1] == g && pb[2] == Gt::identity());
  }
  // round 5: G2Prepared resident on the device (`From<G2Affine> for G2Prepared` once, evaluated by every later multi_miller_loop)
  {
    auto res = G2Prepared::resident_many({hb, G2Affine::generator()});
    REQUIRE(res[0].coeffs().size() == 68 * 36);
    auto ml2 = multi_miller_loop({{ga, res[0]}, {G1Affine::identity(), res[0]}, {G1Affine::generator(), res[1]}});
    REQUIRE(ml2.f == ml.f);                                           // the raw Miller value, limb for limb
    auto mixed = multi_miller_loop({{ga, G2Prepared(hb)}, {G1Affine::generator(), res[1]}});
    REQUIRE(mixed.final_exponentiation() == p + g);
    auto eq2 = multi_miller_loop_many({{{ga, res[0]}, {G1Affine::generator(), res[1]}}, {}, {{ga, G2Prepared(hb)}}});
    REQUIRE(eq2.size() == 3 && eq2[0] == p + g && eq2[1] == Gt::identity() && eq2[2] == p);
  }
  std::printf("host mirror ok\n");
  return 0;
}
