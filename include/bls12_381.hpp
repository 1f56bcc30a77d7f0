// bls12_381.hpp -- header-only C++ mirror of the zkcrypto/bls12_381 hot-path API over libblsgpu.so.
//
// Same names and semantics as the reference's Rust surface (/root/reference/src/lib.rs:49-83): Scalar, G1Affine,
// G1Projective, G2Affine, G2Projective, Gt, MillerLoopResult, G2Prepared, pairing(), multi_miller_loop().  Every
// group / pairing operation is executed by the HIP kernels through the C ABI (bls12_381_hip.h); values are the
// reference's in-memory limbs (canonical Montgomery form, R = 2^384), so they can be memcpy'd to/from a Rust caller.
// Errors from the ABI are thrown as std::runtime_error (the reference's functions are infallible; there is no
// CPU fallback here).
#pragma once
#include <array>
#include <memory>
#include <optional>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bls12_381_hip.h"

namespace bls {

inline void check(int rc, const char* what) {
  if (rc != BLSGPU_OK) throw std::runtime_error(std::string(what) + ": " + blsgpu_last_error());
}

class Context {
 public:
  explicit Context(int device = 0) { check(blsgpu_create(device, &h_), "blsgpu_create"); }
  ~Context() { blsgpu_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  blsgpu_ctx* handle() const { return h_; }
  static Context& instance() { static Context c(0); return c; }
 private:
  blsgpu_ctx* h_ = nullptr;
};

// src/scalar.rs: only the byte form is on the hot path (Scalar::to_bytes, :284-296)
struct Scalar {
  std::array<uint8_t, 32> bytes{};                       // little-endian canonical integer in [0, r)
  static Scalar from_u64(uint64_t v) { Scalar s; std::memcpy(s.bytes.data(), &v, 8); return s; }
  // scalar.rs:256-280: CtOption::none for a value >= r (here: std::nullopt)
  static std::optional<Scalar> from_bytes(const uint8_t b[32]) {
    static constexpr uint8_t kR[32] = {0x01, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0x02, 0xa4, 0xbd, 0x53,
                                       0x05, 0xd8, 0xa1, 0x09, 0x08, 0xd8, 0x39, 0x33, 0x48, 0x7d, 0x9d, 0x29, 0x53, 0xa7, 0xed, 0x73};
    for (int i = 31; i >= 0; i--) {
      if (b[i] < kR[i]) { Scalar s; std::memcpy(s.bytes.data(), b, 32); return s; }
      if (b[i] > kR[i]) return std::nullopt;
    }
    return std::nullopt;               // == r
  }
  const std::array<uint8_t, 32>& to_bytes() const { return bytes; }
};

namespace detail {
constexpr uint64_t kOne[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
constexpr uint64_t kG1Gen[12] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
constexpr uint64_t kG2Gen[24] = {0xf5f28fa202940a10ull, 0xb3f5fb2687b4961aull, 0xa1a893b53e2ae580ull, 0x9894999d1a3caee9ull, 0x6f67b7631863366bull, 0x058191924350bcd7ull, 0xa5a9c0759e23f606ull, 0xaaa0c59dbccd60c3ull, 0x3bb17e18e2867806ull, 0x1b1ab6cc8541b367ull, 0xc2b6ed0ef2158547ull, 0x11922a097360edf3ull, 0x4c730af860494c4aull, 0x597cfa1f5e369c5aull, 0xe7e6856caa0a635aull, 0xbbefb5e96e0d495full, 0x07d3a975f0ef25a2ull, 0x0083fd8e7e80dae5ull, 0xadc0fc92df64b05dull, 0x18aa270a2b1461dcull, 0x86adac6a3be4eba0ull, 0x79495c4ec93da33aull, 0xe7175850a43ccaedull, 0x0b2bc2a163de1bf2ull};
}  // namespace detail

template <int G> struct Projective;

// src/g1.rs:28-32 / src/g2.rs: affine point = x | y limbs + infinity flag
template <int G> struct Affine {
  static constexpr int W = G == 1 ? 12 : 24;
  std::array<uint64_t, W> xy{};
  bool infinity = false;
  static Affine identity() {                                 // (0, 1, infinity)   src/g1.rs:187-193
    Affine a; a.infinity = true; std::memcpy(a.xy.data() + W / 2, detail::kOne, 48); return a;
  }
  static Affine generator() {                                // src/g1.rs:197-217, src/g2.rs:210-250
    Affine a; std::memcpy(a.xy.data(), G == 1 ? detail::kG1Gen : detail::kG2Gen, W * 8); return a;
  }
  bool is_identity() const { return infinity; }
  bool operator==(const Affine& o) const { return (infinity && o.infinity) || (infinity == o.infinity && xy == o.xy); }
  Projective<G> operator*(const Scalar& s) const;            // `&G1Affine * &Scalar`, src/g1.rs:573-579
};

// src/g1.rs:442-446: homogeneous projective (X:Y:Z), identity (0:1:0)
template <int G> struct Projective {
  static constexpr int W = G == 1 ? 18 : 36;
  std::array<uint64_t, W> xyz{};
  static Projective identity() { Projective p; std::memcpy(p.xyz.data() + W / 3, detail::kOne, 48); return p; }
  static Projective generator() {
    Projective p; auto g = Affine<G>::generator();
    std::memcpy(p.xyz.data(), g.xy.data(), g.xy.size() * 8); std::memcpy(p.xyz.data() + 2 * W / 3, detail::kOne, 48); return p;
  }
  Affine<G> to_affine() const {                              // `G1Affine::from(&G1Projective)`, src/g1.rs:49-63
    Affine<G> a; uint8_t inf = 0;
    check((G == 1 ? blsgpu_g1_batch_normalize : blsgpu_g2_batch_normalize)(Context::instance().handle(), xyz.data(), 1, a.xy.data(), &inf), "batch_normalize");
    a.infinity = inf != 0; return a;
  }
  static std::vector<Affine<G>> batch_normalize(const std::vector<Projective>& p) {      // src/g1.rs:806-839
    std::vector<Affine<G>> out(p.size());
    if (p.empty()) return out;
    std::vector<uint64_t> in(p.size() * W), xy(p.size() * Affine<G>::W); std::vector<uint8_t> inf(p.size());
    for (size_t i = 0; i < p.size(); i++) std::memcpy(in.data() + i * W, p[i].xyz.data(), W * 8);
    check((G == 1 ? blsgpu_g1_batch_normalize : blsgpu_g2_batch_normalize)(Context::instance().handle(), in.data(), p.size(), xy.data(), inf.data()), "batch_normalize");
    for (size_t i = 0; i < p.size(); i++) { std::memcpy(out[i].xy.data(), xy.data() + i * Affine<G>::W, Affine<G>::W * 8); out[i].infinity = inf[i] != 0; }
    return out;
  }
  Projective operator+(const Projective& o) const {          // RCB15 Alg. 7, src/g1.rs:670-712
    Projective r; check(blsgpu_point_op(Context::instance().handle(), G, 0, xyz.data(), o.xyz.data(), nullptr, 1, r.xyz.data()), "point add"); return r;
  }
  Projective operator+(const Affine<G>& o) const {           // RCB15 Alg. 8, src/g1.rs:715-752
    Projective r; uint8_t inf = o.infinity;
    check(blsgpu_point_op(Context::instance().handle(), G, 2, xyz.data(), o.xy.data(), &inf, 1, r.xyz.data()), "point add_mixed"); return r;
  }
  Projective dbl() const {                                   // `double`, src/g1.rs:638-667
    Projective r; check(blsgpu_point_op(Context::instance().handle(), G, 1, xyz.data(), nullptr, nullptr, 1, r.xyz.data()), "point double"); return r;
  }
  Projective operator*(const Scalar& s) const { return to_affine() * s; }                // src/g1.rs:556-562
  bool operator==(const Projective& o) const { return to_affine() == o.to_affine(); }    // src/g1.rs:479-496
  static Projective sum(const std::vector<Projective>& p) {                              // `Sum`, src/g1.rs:161-171
    Projective r = identity();
    if (p.empty()) return r;
    std::vector<uint64_t> in(p.size() * W);
    for (size_t i = 0; i < p.size(); i++) std::memcpy(in.data() + i * W, p[i].xyz.data(), W * 8);
    check((G == 1 ? blsgpu_g1_sum : blsgpu_g2_sum)(Context::instance().handle(), in.data(), p.size(), r.xyz.data()), "sum"); return r;
  }
};

// bases.iter().zip(scalars).map(|(p, s)| p * s).sum()
template <int G> Projective<G> msm(const std::vector<Affine<G>>& bases, const std::vector<Scalar>& scalars) {
  if (bases.size() != scalars.size()) throw std::invalid_argument("msm: bases and scalars differ in length");
  size_t n = bases.size();
  std::vector<uint64_t> xy(n * Affine<G>::W); std::vector<uint8_t> inf(n), s(n * 32);
  for (size_t i = 0; i < n; i++) {
    std::memcpy(xy.data() + i * Affine<G>::W, bases[i].xy.data(), Affine<G>::W * 8);
    inf[i] = bases[i].infinity; std::memcpy(s.data() + 32 * i, scalars[i].bytes.data(), 32);
  }
  Projective<G> r;
  check((G == 1 ? blsgpu_g1_msm_host : blsgpu_g2_msm_host)(Context::instance().handle(), xy.data(), inf.data(), s.data(), n, r.xyz.data()), "msm");
  return r;
}
// points.iter().zip(scalars).map(|(p, s)| p * s) collected: `Mul<&Scalar>` over slices (src/g1.rs:573-579, src/g2.rs:626-632)
template <int G> std::vector<Projective<G>> mul_batch(const std::vector<Affine<G>>& points, const std::vector<Scalar>& scalars) {
  if (points.size() != scalars.size()) throw std::invalid_argument("mul_batch: points and scalars differ in length");
  size_t n = points.size();
  std::vector<uint64_t> xy(n * Affine<G>::W), out(n * Projective<G>::W); std::vector<uint8_t> inf(n), s(n * 32);
  for (size_t i = 0; i < n; i++) {
    std::memcpy(xy.data() + i * Affine<G>::W, points[i].xy.data(), Affine<G>::W * 8);
    inf[i] = points[i].infinity; std::memcpy(s.data() + 32 * i, scalars[i].bytes.data(), 32);
  }
  check((G == 1 ? blsgpu_g1_mul_batch : blsgpu_g2_mul_batch)(Context::instance().handle(), xy.data(), inf.data(), s.data(), n, out.data()), "mul_batch");
  std::vector<Projective<G>> r(n);
  for (size_t i = 0; i < n; i++) std::memcpy(r[i].xyz.data(), out.data() + i * Projective<G>::W, Projective<G>::W * 8);
  return r;
}
template <int G> Projective<G> Affine<G>::operator*(const Scalar& s) const { return mul_batch<G>({*this}, {s})[0]; }

using G1Affine = Affine<1>;
using G2Affine = Affine<2>;
using G1Projective = Projective<1>;
using G2Projective = Projective<2>;

// src/pairings.rs:204-337 -- written additively like the reference: + is the Fp12 product, unary - the conjugate
struct Gt {
  std::array<uint64_t, 72> f{};
  static Gt identity() { Gt g; std::memcpy(g.f.data(), detail::kOne, 48); return g; }
  Gt operator+(const Gt& o) const { Gt r; check(blsgpu_fp12_op(Context::instance().handle(), 0, f.data(), o.f.data(), 1, r.f.data()), "Gt add"); return r; }
  Gt operator-() const { Gt r; check(blsgpu_fp12_op(Context::instance().handle(), 8, f.data(), nullptr, 1, r.f.data()), "Gt neg"); return r; }
  Gt dbl() const { Gt r; check(blsgpu_fp12_op(Context::instance().handle(), 3, f.data(), nullptr, 1, r.f.data()), "Gt double"); return r; }
  Gt operator*(const Scalar& s) const {                      // `&Gt * &Scalar`, src/pairings.rs:297-322
    Gt r; check(blsgpu_gt_mul_scalar_batch(Context::instance().handle(), f.data(), s.bytes.data(), 1, r.f.data()), "Gt mul"); return r;
  }
  bool operator==(const Gt& o) const { return f == o.f; }
  static Gt generator();
};

// src/pairings.rs:26 -- no equality on purpose (:21-26)
struct MillerLoopResult {
  std::array<uint64_t, 72> f{};
  static MillerLoopResult default_value() { MillerLoopResult m; std::memcpy(m.f.data(), detail::kOne, 48); return m; }
  Gt final_exponentiation() const {                          // src/pairings.rs:48-176
    Gt g; check(blsgpu_final_exponentiation_batch(Context::instance().handle(), f.data(), 1, g.f.data()), "final_exponentiation"); return g;
  }
  MillerLoopResult operator+(const MillerLoopResult& o) const {                          // src/pairings.rs:179-186
    MillerLoopResult r; check(blsgpu_fp12_op(Context::instance().handle(), 0, f.data(), o.f.data(), 1, r.f.data()), "MillerLoopResult add"); return r;
  }
};

// m `G2Prepared` values resident on the device (blsgpu_g2_prepared): shared by the G2Prepared objects built from it
class PreparedTable {
 public:
  explicit PreparedTable(const std::vector<G2Affine>& pts) {
    std::vector<uint64_t> xy(pts.size() * 24 + 1); std::vector<uint8_t> inf(pts.size() + 1);
    for (size_t i = 0; i < pts.size(); i++) { std::memcpy(xy.data() + 24 * i, pts[i].xy.data(), 192); inf[i] = pts[i].infinity; }
    check(blsgpu_g2_prepare(Context::instance().handle(), xy.data(), inf.data(), pts.size(), &h_), "g2_prepare");
  }
  ~PreparedTable() { blsgpu_g2_prepared_free(h_); }
  PreparedTable(const PreparedTable&) = delete;
  PreparedTable& operator=(const PreparedTable&) = delete;
  const blsgpu_g2_prepared* handle() const { return h_; }
  size_t size() const { return blsgpu_g2_prepared_len(h_); }
 private:
  blsgpu_g2_prepared* h_ = nullptr;
};
// src/pairings.rs:487-546: opaque in the reference.  G2Prepared(q) keeps the affine point (its lines are computed on the fly in every
// Miller loop); G2Prepared::resident(q) / resident_many(points) do what `From<G2Affine>` does in the reference: the 68 coefficient triples
// are computed ONCE, into a device-resident table, and every later multi_miller_loop only evaluates them.
struct G2Prepared {
  G2Affine q;
  std::shared_ptr<PreparedTable> table;                    // null: not resident
  uint32_t index = BLSGPU_UNPREPARED;
  explicit G2Prepared(const G2Affine& p) : q(p) {}
  static std::vector<G2Prepared> resident_many(const std::vector<G2Affine>& pts) {
    auto t = std::make_shared<PreparedTable>(pts);
    std::vector<G2Prepared> out;
    for (size_t i = 0; i < pts.size(); i++) { G2Prepared g(pts[i]); g.table = t; g.index = (uint32_t)i; out.push_back(g); }
    return out;
  }
  static G2Prepared resident(const G2Affine& p) { return resident_many({p})[0]; }
  // `coeffs: Vec<(Fp2, Fp2, Fp2)>` of a resident value: 68 x 3 x 12 u64 in the reference's value format
  std::vector<uint64_t> coeffs() const {
    if (!table) throw std::invalid_argument("G2Prepared::coeffs: not resident");
    std::vector<uint64_t> c(68 * 36); uint8_t inf = 0;
    check(blsgpu_g2_prepared_coeffs(Context::instance().handle(), table->handle(), index, c.data(), &inf), "g2_prepared_coeffs");
    return c;
  }
};
namespace detail {
// the table shared by the resident terms (the first one found) and the per-term indices; terms of another table stay unprepared
template <class It> const PreparedTable* resident_indices(It begin, It end, std::vector<uint32_t>& qi) {
  const PreparedTable* t = nullptr;
  for (It i = begin; i != end; ++i) if (i->second.table) { t = i->second.table.get(); break; }
  if (!t) return nullptr;
  for (It i = begin; i != end; ++i) qi.push_back(i->second.table.get() == t ? i->second.index : BLSGPU_UNPREPARED);
  return t;
}
}  // namespace detail

inline Gt pairing(const G1Affine& p, const G2Affine& q) {    // src/pairings.rs:607-653
  Gt g; uint8_t i1 = p.infinity, i2 = q.infinity;
  check(blsgpu_pairing_batch(Context::instance().handle(), p.xy.data(), &i1, q.xy.data(), &i2, 1, g.f.data()), "pairing"); return g;
}
inline std::vector<Gt> pairing_batch(const std::vector<G1Affine>& p, const std::vector<G2Affine>& q) {
  if (p.size() != q.size()) throw std::invalid_argument("pairing_batch: length mismatch");
  size_t n = p.size(); std::vector<Gt> out(n);
  if (!n) return out;
  std::vector<uint64_t> a(n * 12), b(n * 24), o(n * 72); std::vector<uint8_t> fa(n), fb(n);
  for (size_t i = 0; i < n; i++) { std::memcpy(a.data() + 12 * i, p[i].xy.data(), 96); std::memcpy(b.data() + 24 * i, q[i].xy.data(), 192); fa[i] = p[i].infinity; fb[i] = q[i].infinity; }
  check(blsgpu_pairing_batch(Context::instance().handle(), a.data(), fa.data(), b.data(), fb.data(), n, o.data()), "pairing_batch");
  for (size_t i = 0; i < n; i++) std::memcpy(out[i].f.data(), o.data() + 72 * i, 576);
  return out;
}
inline MillerLoopResult multi_miller_loop(const std::vector<std::pair<G1Affine, G2Prepared>>& terms) {   // src/pairings.rs:554-603
  size_t n = terms.size();
  std::vector<uint64_t> a(n * 12), b(n * 24); std::vector<uint8_t> fa(n), fb(n);
  for (size_t i = 0; i < n; i++) {
    std::memcpy(a.data() + 12 * i, terms[i].first.xy.data(), 96); std::memcpy(b.data() + 24 * i, terms[i].second.q.xy.data(), 192);
    fa[i] = terms[i].first.infinity; fb[i] = terms[i].second.q.infinity;
  }
  MillerLoopResult m;
  std::vector<uint32_t> qi;
  if (const PreparedTable* t = detail::resident_indices(terms.begin(), terms.end(), qi))
    check(blsgpu_multi_miller_loop_prepared(Context::instance().handle(), a.data(), fa.data(), b.data(), fb.data(), qi.data(), t->handle(), n, m.f.data()), "multi_miller_loop_prepared");
  else
    check(blsgpu_multi_miller_loop(Context::instance().handle(), a.data(), fa.data(), b.data(), fb.data(), n, m.f.data()), "multi_miller_loop");
  return m;
}
// N independent `multi_miller_loop(..).final_exponentiation()` in one device call: the bulk form of signature verification
// (src/pairings.rs:554-603, 817-824; blsgpu_multi_miller_loop_many)
inline std::vector<Gt> multi_miller_loop_many(const std::vector<std::vector<std::pair<G1Affine, G2Prepared>>>& equations) {
  size_t n = 0;
  std::vector<uint64_t> off(equations.size() + 1, 0);
  for (size_t s = 0; s < equations.size(); s++) { n += equations[s].size(); off[s + 1] = n; }
  std::vector<uint64_t> a(n * 12 + 1), b(n * 24 + 1), o(equations.size() * 72 + 1); std::vector<uint8_t> fa(n + 1), fb(n + 1);
  size_t i = 0;
  for (const auto& e : equations)
    for (const auto& t : e) {
      std::memcpy(a.data() + 12 * i, t.first.xy.data(), 96); std::memcpy(b.data() + 24 * i, t.second.q.xy.data(), 192);
      fa[i] = t.first.infinity; fb[i] = t.second.q.infinity; i++;
    }
  std::vector<std::pair<G1Affine, G2Prepared>> flat;
  for (const auto& e : equations) flat.insert(flat.end(), e.begin(), e.end());
  std::vector<uint32_t> qi;
  if (const PreparedTable* t = detail::resident_indices(flat.begin(), flat.end(), qi))
    check(blsgpu_multi_miller_loop_prepared_many(Context::instance().handle(), a.data(), fa.data(), b.data(), fb.data(), qi.data(), t->handle(), off.data(), equations.size(), 1,
                                                 o.data()), "multi_miller_loop_prepared_many");
  else
    check(blsgpu_multi_miller_loop_many(Context::instance().handle(), a.data(), fa.data(), b.data(), fb.data(), off.data(), equations.size(), 1, o.data()), "multi_miller_loop_many");
  std::vector<Gt> out(equations.size());
  for (size_t s = 0; s < equations.size(); s++) std::memcpy(out[s].f.data(), o.data() + 72 * s, 576);
  return out;
}
inline Gt Gt::generator() { return pairing(G1Affine::generator(), G2Affine::generator()); }            // src/pairings.rs:359-475

// Bulk signature verification from bytes (blsgpu_bls_verify_batch): compressed keys, signatures and messages in, one verdict byte each
// out (1 valid, 0 invalid, 2 bad key encoding, 3 bad signature encoding); keys_in_g1 = true: 48-byte keys / 96-byte signatures
inline std::vector<uint8_t> bls_verify_batch(bool keys_in_g1, const std::vector<uint8_t>& pk_bytes, const std::vector<uint8_t>& sig_bytes, const std::vector<std::string>& msgs,
                                             const std::string& dst) {
  const size_t n = msgs.size();
  if (pk_bytes.size() != n * (keys_in_g1 ? 48 : 96) || sig_bytes.size() != n * (keys_in_g1 ? 96 : 48)) throw std::invalid_argument("bls_verify_batch: byte lengths");
  std::vector<uint8_t> verdict(n);
  if (!n) return verdict;
  std::vector<uint64_t> offs(n + 1, 0); std::string blob;
  for (size_t i = 0; i < n; i++) { blob += msgs[i]; offs[i + 1] = blob.size(); }
  check(blsgpu_bls_verify_batch(Context::instance().handle(), keys_in_g1 ? 0 : 1, pk_bytes.data(), sig_bytes.data(), (const uint8_t*)blob.data(), offs.data(), n,
                                (const uint8_t*)dst.data(), dst.size(), verdict.data()), "bls_verify_batch");
  return verdict;
}

// src/hash_to_curve/mod.rs:86-108 with ExpandMsgXmd<Sha256>: `G::hash_to_curve(msg, dst)` / `G::encode_to_curve(msg, dst)` for a batch
template <int G> std::vector<Projective<G>> hash_to_curve(const std::vector<std::string>& msgs, const std::string& dst, bool encode_only = false) {
  std::vector<Projective<G>> out(msgs.size());
  if (msgs.empty()) return out;
  std::vector<uint64_t> offs(msgs.size() + 1, 0); std::string blob;
  for (size_t i = 0; i < msgs.size(); i++) { blob += msgs[i]; offs[i + 1] = blob.size(); }
  std::vector<uint64_t> xyz(msgs.size() * Projective<G>::W);
  check((G == 1 ? blsgpu_g1_hash_to_curve_batch : blsgpu_g2_hash_to_curve_batch)(Context::instance().handle(), (const uint8_t*)blob.data(), offs.data(), msgs.size(),
                                                                                 (const uint8_t*)dst.data(), dst.size(), encode_only ? 1 : 0, xyz.data()), "hash_to_curve");
  for (size_t i = 0; i < msgs.size(); i++) std::memcpy(out[i].xyz.data(), xyz.data() + i * Projective<G>::W, Projective<G>::W * 8);
  return out;
}

// src/scalar.rs: vectors of `Scalar([u64; 4])` (Montgomery limbs) -- element-wise arithmetic and the radix-2 transform built on ROOT_OF_UNITY
using FrLimbs = std::array<uint64_t, 4>;
enum class FrOp { Mul = 0, Add = 1, Sub = 2, Square = 3, Invert = 4, Neg = 5, Double = 6 };
inline std::vector<FrLimbs> fr_op(FrOp op, const std::vector<FrLimbs>& a, const std::vector<FrLimbs>& b = {}) {
  std::vector<FrLimbs> out(a.size());
  if (a.empty()) return out;
  if ((int)op <= 2 && b.size() != a.size()) throw std::invalid_argument("fr_op: operands differ in length");
  check(blsgpu_fr_op(Context::instance().handle(), (int)op, a[0].data(), b.empty() ? nullptr : b[0].data(), a.size(), out[0].data(), nullptr), "fr_op");
  return out;
}
inline void fr_ntt(std::vector<FrLimbs>& v, bool inverse = false) {         // natural order in and out; v.size() a power of two
  if (v.empty() || (v.size() & (v.size() - 1))) throw std::invalid_argument("fr_ntt: length must be a power of two");
  int log_n = 0; while (((size_t)1 << log_n) < v.size()) log_n++;
  check(blsgpu_fr_ntt(Context::instance().handle(), v[0].data(), log_n, inverse ? 1 : 0), "fr_ntt");
}

// The same operations sharded over several GPUs of one node from this process (blsgpu_group: one context + one host thread per
// listed device; partial results folded with `Sum` / `MillerLoopResult + MillerLoopResult`, src/g1.rs:161-171, src/pairings.rs:179-186)
class Group {
 public:
  explicit Group(const std::vector<int>& devices) { check(blsgpu_group_create(devices.data(), (int)devices.size(), &h_), "blsgpu_group_create"); }
  ~Group() { blsgpu_group_destroy(h_); }
  Group(const Group&) = delete;
  Group& operator=(const Group&) = delete;
  blsgpu_group* handle() const { return h_; }
  int size() const { return blsgpu_group_size(h_); }
  // sum_i scalars[i] * bases[i] with bases and scalars dealt to the members in contiguous slices
  template <int G> Projective<G> msm(const std::vector<Affine<G>>& bases, const std::vector<Scalar>& scalars) const {
    if (bases.size() != scalars.size()) throw std::invalid_argument("msm: length mismatch");
    const size_t n = bases.size();
    std::vector<uint64_t> xy(n * Affine<G>::W + 1); std::vector<uint8_t> inf(n + 1), sb(n * 32 + 1);
    for (size_t i = 0; i < n; i++) { std::memcpy(xy.data() + i * Affine<G>::W, bases[i].xy.data(), Affine<G>::W * 8); inf[i] = bases[i].infinity; std::memcpy(sb.data() + 32 * i, scalars[i].bytes.data(), 32); }
    blsgpu_group_bases* b = nullptr;
    check(blsgpu_group_bases_upload(h_, G, xy.data(), inf.data(), n, &b), "group_bases_upload");
    Projective<G> r;
    int rc = G == 1 ? blsgpu_g1_msm_sharded(h_, b, sb.data(), n, r.xyz.data()) : blsgpu_g2_msm_sharded(h_, b, sb.data(), n, r.xyz.data());
    blsgpu_group_bases_free(b);
    check(rc, "msm_sharded");
    return r;
  }
  // multi_miller_loop(terms).final_exponentiation() with the terms dealt to the members and ONE final exponentiation
  Gt multi_miller_loop_final_exp(const std::vector<std::pair<G1Affine, G2Prepared>>& terms) const {
    const size_t n = terms.size();
    std::vector<uint64_t> a(n * 12 + 1), b(n * 24 + 1); std::vector<uint8_t> fa(n + 1), fb(n + 1);
    for (size_t i = 0; i < n; i++) {
      std::memcpy(a.data() + 12 * i, terms[i].first.xy.data(), 96); std::memcpy(b.data() + 24 * i, terms[i].second.q.xy.data(), 192);
      fa[i] = terms[i].first.infinity; fb[i] = terms[i].second.q.infinity;
    }
    Gt g;
    check(blsgpu_multi_miller_loop_sharded(h_, a.data(), fa.data(), b.data(), fb.data(), n, 1, g.f.data()), "multi_miller_loop_sharded");
    return g;
  }
  std::vector<Gt> pairing_batch(const std::vector<G1Affine>& p, const std::vector<G2Affine>& q) const {
    if (p.size() != q.size()) throw std::invalid_argument("pairing_batch: length mismatch");
    const size_t n = p.size(); std::vector<Gt> out(n);
    if (!n) return out;
    std::vector<uint64_t> a(n * 12), b(n * 24), o(n * 72); std::vector<uint8_t> fa(n), fb(n);
    for (size_t i = 0; i < n; i++) { std::memcpy(a.data() + 12 * i, p[i].xy.data(), 96); std::memcpy(b.data() + 24 * i, q[i].xy.data(), 192); fa[i] = p[i].infinity; fb[i] = q[i].infinity; }
    check(blsgpu_pairing_batch_sharded(h_, a.data(), fa.data(), b.data(), fb.data(), n, o.data()), "pairing_batch_sharded");
    for (size_t i = 0; i < n; i++) std::memcpy(out[i].f.data(), o.data() + 72 * i, 576);
    return out;
  }
 private:
  blsgpu_group* h_ = nullptr;
};

struct Bls12 {                                              // `pairing::Engine` / `MultiMillerLoop`, src/pairings.rs:790-824
  static Gt pairing(const G1Affine& p, const G2Affine& q) { return bls::pairing(p, q); }
  static MillerLoopResult multi_miller_loop(const std::vector<std::pair<G1Affine, G2Prepared>>& t) { return bls::multi_miller_loop(t); }
};

}  // namespace bls
