// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Wire formats and key digests, restated from:
//   * src/digest.rs:22-77        Digestible / SimpleDigestible / DigestComputer: SHA-256 over the byte stream of `write_bytes`; for plain
//                                serde types that stream is bincode `DefaultOptions` + little-endian + fixint (:33-41)
//   * src/spartan.rs:62-104      SpartanVerifierKey: bincode(vk_ee) || bincode(ck_s) || S.write_bytes()
//   * src/r1cs/mod.rs:775-794    SplitR1CSShape::write_bytes (ten u64 + three raw matrices); src/r1cs/sparse.rs:398-417 write_digest_bytes
//   * src/neutronnova_zk.rs:1290-1333  NeutronNovaVerifierKey::write_bytes
//   * struct field orders        SpartanSNARK src/spartan.rs:125-137, SplitR1CSInstance src/r1cs/mod.rs:797-806, SumcheckProof src/sumcheck.rs:39-43,
//                                CompressedUniPoly src/polys/univariate.rs:34-37, HyraxCommitmentKey / VerifierKey / Commitment / Blind /
//                                EvaluationArgument src/provider/pcs/hyrax_pc.rs:56-131, InnerProductArgumentLinear src/provider/pcs/ipa.rs:103-114,
//                                NeutronNovaZkSNARK src/neutronnova_zk.rs:1373-1385, SplitMultiRoundR1CSInstance src/r1cs/mod.rs:1424-1430,
//                                NovaNIFS src/nifs.rs:21-25, RelaxedR1CSInstance src/r1cs/mod.rs:211-218, RelaxedR1CSSpartanProof
//                                src/spartan_relaxed.rs:79-91, SplitMultiRoundR1CSShape src/r1cs/mod.rs:1401-1419, R1CSShape :169-179,
//                                SparseMatrix src/r1cs/sparse.rs:383-394
//
// bincode 1.3 with fixint + little-endian: usize / u64 = 8 bytes LE; Vec<T> = u64 length, then the elements; Option<T> = one byte 0 / 1, then T;
// structs and tuples = their fields in declaration order, nothing else; fixed-size arrays = their elements, no length.
//
// THE ONE ASSUMPTION (third-party, not in /root/reference; Cargo.toml:41-46 `halo2curves 0.10` with `derive_serde`): a field element
// serialises as its `to_repr()` bytes — 32 bytes, canonical value, little-endian — and is rejected on read when the value is >= the modulus; an
// affine point is the struct {x, y} (64 bytes, identity = (0, 0)), a projective point (`E::GE`) the struct {x, y, z} (96 bytes, identity z = 0).
// The reference serialises a projective point in whatever Jacobian representative its last operation left; this build always WRITES the
// normalised representative (x, y, 1) (identity: (0, 0, 0)) and READS any representative. A reference-produced proof therefore decodes to the
// same group elements here, but its commitment bytes are not claimed equal byte for byte: parity of these bytes against a reference build is
// UNPINNED (no reference vector exists; the reference cannot be built in this image). What IS pinned: SHA-256 against FIPS 180-4 vectors and
// hashlib, the framing against an independent Python writer (tests/pywire.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "hyrax.hpp"
#include "sparse.hpp"
#include "sumcheck.hpp"

namespace oracle {

// ---- SHA-256 (FIPS 180-4) -----------------------------------------------------------------------------------------------------------------
struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t total = 0;
  size_t fill = 0;
  Sha256() {
    static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h, iv, 32);
  }
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), maj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + maj;
      hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
    }
    h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
  }
  void update(const uint8_t* p, size_t n) {
    total += n;
    if (fill) {
      size_t k = std::min(n, 64 - fill);
      memcpy(buf + fill, p, k);
      fill += k, p += k, n -= k;
      if (fill < 64) return;
      block(buf);
      fill = 0;
    }
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) memcpy(buf, p, n), fill = n;
  }
  void finalize(uint8_t out[32]) {
    uint64_t bits = total * 8;
    uint8_t pad[72] = {0x80};
    size_t padlen = (fill < 56 ? 56 : 120) - fill;
    for (int i = 0; i < 8; ++i) pad[padlen + i] = (uint8_t)(bits >> (56 - 8 * i));
    update(pad, padlen + 8);
    for (int i = 0; i < 8; ++i) out[4 * i] = h[i] >> 24, out[4 * i + 1] = h[i] >> 16, out[4 * i + 2] = h[i] >> 8, out[4 * i + 3] = h[i];
  }
};

// ---- writer: into a byte vector or straight into the hasher (DigestComputer streams, src/digest.rs:62-76) ---------------------------------
struct WireWriter {
  std::vector<uint8_t>* buf = nullptr;
  Sha256* sha = nullptr;
  explicit WireWriter(std::vector<uint8_t>* b) : buf(b) {}
  explicit WireWriter(Sha256* s) : sha(s) {}
  void raw(const void* p, size_t n) {
    if (buf) buf->insert(buf->end(), (const uint8_t*)p, (const uint8_t*)p + n);
    else sha->update((const uint8_t*)p, n);
  }
  void u8(uint8_t v) { raw(&v, 1); }
  void u64(uint64_t v) { raw(&v, 8); }  // little-endian host
  template <class F>
  void fe(const F& f) {
    uint8_t b[32];
    f.to_repr(b);
    raw(b, 32);
  }
  void scalars(const std::vector<Fq>& v) {  // Vec<E::Scalar>
    u64(v.size());
    for (const Fq& f : v) fe(f);
  }
  void usizes(const std::vector<size_t>& v) {
    u64(v.size());
    for (size_t x : v) u64(x);
  }
  void affine(const Affine& a) {
    fe(a.x);
    fe(a.y);
  }
  void point_affine(const Affine& a) {  // E::GE, normalised representative
    if (a.is_identity()) {
      fe(Fp::zero()), fe(Fp::zero()), fe(Fp::zero());
    } else {
      fe(a.x), fe(a.y), fe(Fp::one());
    }
  }
  void point(const Jac& p) { point_affine(p.to_affine()); }
  void commitment(const HyraxCommitment& c) {  // HyraxCommitment { comm: Vec<E::GE> }
    u64(c.size());
    for (const Affine& a : batch_affine(c)) point_affine(a);
  }
  void option_commitment(const HyraxCommitment& c) {  // Option<Commitment<E>>: present exactly when the segment has rows
    u8(c.empty() ? 0 : 1);
    if (!c.empty()) commitment(c);
  }
  void hyrax_key(const HyraxKey& k) {  // HyraxCommitmentKey and HyraxVerifierKey serialise the same three fields (tables are #[serde(skip)])
    u64(k.num_cols);
    u64(k.ck.size());
    for (const Affine& a : k.ck) affine(a);
    point(k.h);
  }
  void sumcheck(const SumcheckProof<Fq>& p) {  // SumcheckProof { compressed_polys: Vec<CompressedUniPoly { coeffs_except_linear_term: Vec<F> }> }
    u64(p.compressed_polys.size());
    for (const auto& c : p.compressed_polys) scalars(c);
  }
  void ipa(const IpaProof& a) {  // HyraxEvaluationArgument { ipa: InnerProductArgumentLinear { delta, beta, z_vec, z_delta, z_beta } }
    point(a.delta);
    point(a.beta);
    scalars(a.z_vec);
    fe(a.z_delta);
    fe(a.z_beta);
  }
  void matrix_bincode(const SparseMatrix<Fq>& M) {  // derived Serialize of SparseMatrix: data, indices, indptr, cols
    scalars(M.data);
    usizes(M.indices);
    usizes(M.indptr);
    u64(M.cols);
  }
  void matrix_digest_bytes(const SparseMatrix<Fq>& M) {  // src/r1cs/sparse.rs:398-417
    u64(M.data.size());
    u64(M.indices.size());
    u64(M.indptr.size());
    u64(M.cols);
    for (const Fq& d : M.data) fe(d);
    for (size_t i : M.indices) u64(i);
    for (size_t p : M.indptr) u64(p);
  }
  void shape_dims(const SplitR1CSShape<Fq>& S) {
    u64(S.num_cons), u64(S.num_cons_unpadded), u64(S.num_shared_unpadded), u64(S.num_precommitted_unpadded), u64(S.num_rest_unpadded);
    u64(S.num_shared), u64(S.num_precommitted), u64(S.num_rest), u64(S.num_public), u64(S.num_challenges);
  }
  void shape_digest_bytes(const SplitR1CSShape<Fq>& S) {  // SplitR1CSShape::write_bytes, src/r1cs/mod.rs:775-794
    shape_dims(S);
    matrix_digest_bytes(S.A), matrix_digest_bytes(S.B), matrix_digest_bytes(S.C);
  }
  void shape_bincode(const SplitR1CSShape<Fq>& S) {  // derived Serialize (the OnceCell fields are skipped), src/r1cs/mod.rs:742-773
    shape_dims(S);
    matrix_bincode(S.A), matrix_bincode(S.B), matrix_bincode(S.C);
  }
  template <class MRS>
  void multiround_shape(const MRS& S) {  // src/r1cs/mod.rs:1401-1419
    u64(S.num_cons), u64(S.num_cons_unpadded), u64(S.num_rounds);
    usizes(S.vars_unpadded), usizes(S.vars_padded), usizes(S.chals_per_round);
    u64(S.num_public), u64(S.width);
    matrix_bincode(S.A), matrix_bincode(S.B), matrix_bincode(S.C);
  }
  template <class MRS>
  void regular_shape_of(const MRS& S) {  // to_regular_shape (:1659-1672) -> R1CSShape { num_cons, num_vars, num_io, A, B, C }
    u64(S.num_cons), u64(S.total_vars()), u64(S.num_io());
    matrix_bincode(S.A), matrix_bincode(S.B), matrix_bincode(S.C);
  }
};

inline void sha256(const uint8_t* p, size_t n, uint8_t out[32]) {
  Sha256 h;
  h.update(p, n);
  h.finalize(out);
}

// ---- reader ---------------------------------------------------------------------------------------------------------------------------------
struct WireReader {
  const uint8_t *p, *end;
  WireReader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  void need(size_t n) const {
    if ((size_t)(end - p) < n) throw std::runtime_error("wire: unexpected end of input");
  }
  uint8_t u8() {
    need(1);
    return *p++;
  }
  uint64_t u64() {
    need(8);
    uint64_t v;
    memcpy(&v, p, 8);
    p += 8;
    return v;
  }
  size_t len(size_t elem_bytes) {  // a length prefix that the remaining input can actually hold
    uint64_t n = u64();
    if (elem_bytes && n > (uint64_t)(end - p) / elem_bytes) throw std::runtime_error("wire: length prefix exceeds the input");
    return (size_t)n;
  }
  template <class F>
  F fe() {
    need(32);
    uint64_t v[4];
    memcpy(v, p, 32);
    p += 32;
    if (cmp256(v, F::P().p.l) >= 0) throw std::runtime_error("wire: non-canonical field element");
    return F::from_canonical(v);
  }
  std::vector<Fq> scalars() {
    size_t n = len(32);
    std::vector<Fq> v(n);
    for (auto& f : v) f = fe<Fq>();
    return v;
  }
  std::vector<size_t> usizes() {
    size_t n = len(8);
    std::vector<size_t> v(n);
    for (auto& x : v) x = (size_t)u64();
    return v;
  }
  Affine affine() {
    Affine a;
    a.x = fe<Fp>();
    a.y = fe<Fp>();
    if (!on_curve(a)) throw std::runtime_error("wire: point not on the curve");
    return a;
  }
  Jac point() {  // any Jacobian representative
    Jac j;
    j.x = fe<Fp>(), j.y = fe<Fp>(), j.z = fe<Fp>();
    if (j.z.is_zero()) return Jac::identity();
    if (!on_curve(j.to_affine())) throw std::runtime_error("wire: point not on the curve");
    return j;
  }
  HyraxCommitment commitment() {
    size_t n = len(96);
    HyraxCommitment c(n);
    for (auto& j : c) j = point();
    return c;
  }
  HyraxCommitment option_commitment() {
    uint8_t tag = u8();
    if (tag > 1) throw std::runtime_error("wire: bad Option tag");
    return tag ? commitment() : HyraxCommitment();
  }
  SumcheckProof<Fq> sumcheck() {
    SumcheckProof<Fq> s;
    size_t n = len(8);
    s.compressed_polys.resize(n);
    for (auto& c : s.compressed_polys) c = scalars();
    return s;
  }
  IpaProof ipa() {
    IpaProof a;
    a.delta = point();
    a.beta = point();
    a.z_vec = scalars();
    a.z_delta = fe<Fq>();
    a.z_beta = fe<Fq>();
    return a;
  }
  void done() const {
    if (p != end) throw std::runtime_error("wire: trailing bytes");  // bincode DefaultOptions rejects trailing bytes
  }
};

}  // namespace oracle
