// libspartan_hip.so — wire formats and key digests. Host code only: no kernel, no device call.
//
//   * the byte sink / source a Rust caller's serde layer has in `bincode::DefaultOptions::new().with_little_endian().with_fixint_encoding()`
//     (src/digest.rs:33-41): usize = 8 bytes LE, Vec<T> = u64 length + elements, Option<T> = tag byte + T, structs = fields in order;
//   * SHA-256 as DigestComputer uses it (src/digest.rs:49-77): the sink can stream into the hasher instead of collecting bytes;
//   * the two typed objects of the Spartan path: SpartanVerifierKey::write_bytes -> digest (src/spartan.rs:73-104, src/r1cs/mod.rs:775-794,
//     src/r1cs/sparse.rs:398-417) and SpartanSNARK in struct field order (src/spartan.rs:125-137). The NeutronNova objects are composed from the
//     same primitives by host/neutronnova_zk.cpp.
//
// Third-party layouts (halo2curves 0.10 `derive_serde`, not in the reference tree; the one documented assumption): a field
// element is its 32 `to_repr()` bytes (canonical value, little-endian; values >= the modulus are rejected on read), an affine point {x, y}, a
// projective point {x, y, z} — written normalised ((x, y, 1); identity (0, 0, 0)), any Jacobian representative accepted on read.
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "core.hpp"
#include "curve.hpp"
#include "sha256.hpp"

using sp::fail;

typedef FqP SC;  // scalar field of the bench engine

struct sp_wire {
  bool hashing = false;
  sp::Sha256 sha;
  std::vector<uint8_t> buf;
  uint64_t written = 0;
  uint8_t stage[1 << 16];  // BufWriter::with_capacity(64 * 1024, ..) of src/digest.rs:66 (and the batching unit of the collecting mode)
  size_t fill = 0;
  void flush() {
    if (!fill) return;
    if (hashing) sha.update(stage, fill);
    else buf.insert(buf.end(), stage, stage + fill);
    fill = 0;
  }
  void put(const void* p, size_t n) {
    written += n;
    if (n >= sizeof stage) {
      flush();
      if (hashing) sha.update(p, n);
      else buf.insert(buf.end(), (const uint8_t*)p, (const uint8_t*)p + n);
      return;
    }
    if (fill + n > sizeof stage) flush();
    memcpy(stage + fill, p, n);
    fill += n;
  }
  uint8_t* reserve(size_t n) {  // n <= sizeof stage: bytes the caller fills in place
    written += n;
    if (fill + n > sizeof stage) flush();
    uint8_t* p = stage + fill;
    fill += n;
    return p;
  }
  void u64(uint64_t v) { put(&v, 8); }
};

struct sp_unwire {
  const uint8_t *p, *end;
  size_t left() const { return (size_t)(end - p); }
};

namespace {

inline fe_t load_fe(const uint64_t* w) {
  fe_t f;
  memcpy(f.v, w, 32);
  return f;
}
template <class FP>
inline bool canonical_below_p(const fe_t& c) {
  for (int i = 7; i >= 0; --i) {
    if (c.v[i] < FP::P(i)) return true;
    if (c.v[i] > FP::P(i)) return false;
  }
  return false;
}
template <class FP>
inline bool read_fe(sp_unwire* r, fe_t* out) {  // 32 to_repr bytes -> Montgomery limbs; false when short or not canonical
  if (r->left() < 32) return false;
  fe_t c;
  memcpy(c.v, r->p, 32);
  r->p += 32;
  if (!canonical_below_p<FP>(c)) return false;
  *out = fe_from_canonical<FP>(c);
  return true;
}
inline void write_point(sp_wire* w, const aff_t& a) {  // E::GE, normalised representative
  uint8_t* o = w->reserve(96);
  if (aff_is_identity(a)) {
    memset(o, 0, 96);
    return;
  }
  sp::fe_to_le_bytes<B>(a.x, o);
  sp::fe_to_le_bytes<B>(a.y, o + 32);
  memset(o + 64, 0, 32);
  o[64] = 1;
}
inline int len_prefix(sp_wire* w, size_t n, int with_len) {
  if (with_len) w->u64(n);
  return SP_OK;
}

}  // namespace

extern "C" {

int sp_sha256(const uint8_t* data, size_t n, uint8_t out[32]) {
  if ((!data && n) || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_sha256: null argument");
  sp::Sha256 h;
  h.update(data, n);
  h.finish(out);
  return SP_OK;
}
int sp_sha256_accelerated(void) { return sp::Sha256::accelerated() ? 1 : 0; }

// ---- sink --------------------------------------------------------------------------------------------------------------------------------
int sp_wire_new(int hashing, sp_wire** out) {
  if (!out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_new: null out");
  sp_wire* w = new sp_wire();
  w->hashing = hashing != 0;
  *out = w;
  return SP_OK;
}
void sp_wire_free(sp_wire* w) { delete w; }
int sp_wire_raw(sp_wire* w, const uint8_t* bytes, size_t n) {
  if (!w || (!bytes && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_raw: null argument");
  w->put(bytes, n);
  return SP_OK;
}
int sp_wire_u8(sp_wire* w, uint8_t v) {
  if (!w) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_u8: null sink");
  w->put(&v, 1);
  return SP_OK;
}
int sp_wire_u64s(sp_wire* w, const uint64_t* v, size_t n, int with_len) {
  if (!w || (!v && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_u64s: null argument");
  len_prefix(w, n, with_len);
  if (n) w->put(v, 8 * n);  // little-endian host
  return SP_OK;
}
int sp_wire_u32s_as_u64(sp_wire* w, const uint32_t* v, size_t n, int with_len) {
  if (!w || (!v && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_u32s_as_u64: null argument");
  len_prefix(w, n, with_len);
  for (size_t i = 0; i < n;) {
    const size_t k = std::min<size_t>(n - i, 4096);
    uint64_t* o = (uint64_t*)w->reserve(8 * k);
    uint64_t tmp[4096];
    for (size_t j = 0; j < k; ++j) tmp[j] = v[i + j];
    memcpy(o, tmp, 8 * k);
    i += k;
  }
  return SP_OK;
}
int sp_wire_scalars(sp_wire* w, const uint64_t* f, size_t n, int with_len) {
  if (!w || (!f && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_scalars: null argument");
  len_prefix(w, n, with_len);
  // R1CS coefficient arrays repeat a handful of values (+-1, +-2^k): remember the last conversion and a small table of earlier ones
  struct Key {
    uint64_t l[4];
    bool operator==(const Key& o) const { return memcmp(l, o.l, 32) == 0; }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const { return (size_t)(k.l[0] * 0x9e3779b97f4a7c15ull ^ k.l[1] ^ (k.l[2] << 1) ^ (k.l[3] << 2)); }
  };
  std::unordered_map<Key, Key, KeyHash> memo;
  Key last_in{}, last_out{};
  bool have_last = false;
  for (size_t i = 0; i < n; ++i) {
    Key k;
    memcpy(k.l, f + 4 * i, 32);
    uint8_t* o = w->reserve(32);
    if (have_last && k == last_in) {
      memcpy(o, last_out.l, 32);
      continue;
    }
    Key v;
    auto it = n >= 64 ? memo.find(k) : memo.end();
    if (it != memo.end()) {
      v = it->second;
    } else {
      uint8_t b[32];
      sp::fe_to_le_bytes<SC>(load_fe(k.l), b);
      memcpy(v.l, b, 32);
      if (n >= 64 && memo.size() < 4096) memo.emplace(k, v);
    }
    memcpy(o, v.l, 32);
    last_in = k, last_out = v, have_last = true;
  }
  return SP_OK;
}
int sp_wire_affines(sp_wire* w, const uint64_t* aff, size_t n, int with_len) {
  if (!w || (!aff && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_affines: null argument");
  len_prefix(w, n, with_len);
  for (size_t i = 0; i < n; ++i) {
    uint8_t* o = w->reserve(64);
    sp::fe_to_le_bytes<B>(load_fe(aff + 8 * i), o);
    sp::fe_to_le_bytes<B>(load_fe(aff + 8 * i + 4), o + 32);
  }
  return SP_OK;
}
int sp_wire_points(sp_wire* w, const uint64_t* aff, size_t n, int with_len) {
  if (!w || (!aff && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_points: null argument");
  len_prefix(w, n, with_len);
  for (size_t i = 0; i < n; ++i) {
    aff_t a;
    a.x = load_fe(aff + 8 * i);
    a.y = load_fe(aff + 8 * i + 4);
    write_point(w, a);
  }
  return SP_OK;
}
// HyraxCommitmentKey / HyraxVerifierKey { num_cols, ck: Vec<Affine>, h: GE } (src/provider/pcs/hyrax_pc.rs:56-108; the tables are #[serde(skip)])
int sp_wire_hyrax_key(sp_wire* w, const uint64_t* ck, size_t num_cols, const uint64_t* h) {
  if (!w || !ck || !h) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_hyrax_key: null argument");
  w->u64(num_cols);
  int rc = sp_wire_affines(w, ck, num_cols, 1);
  return rc ? rc : sp_wire_points(w, h, 1, 0);
}
// SparseMatrix: digest_form != 0 -> write_digest_bytes (src/r1cs/sparse.rs:398-417: the three lengths and cols, then the raw arrays);
// digest_form == 0 -> the derived Serialize (data, indices, indptr as length-prefixed Vecs, then cols; :383-394)
int sp_wire_matrix(sp_wire* w, const sp_csr* M, size_t rows, size_t cols, int digest_form) {
  if (!w || !M || !M->indptr) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_matrix: null argument");
  const size_t nnz = (size_t)M->indptr[rows];
  if (nnz && (!M->data || !M->indices)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_matrix: null entries");
  int rc;
  if (digest_form) {
    w->u64(nnz), w->u64(nnz), w->u64(rows + 1), w->u64(cols);
    if ((rc = sp_wire_scalars(w, M->data, nnz, 0)) || (rc = sp_wire_u32s_as_u64(w, M->indices, nnz, 0)) || (rc = sp_wire_u64s(w, M->indptr, rows + 1, 0))) return rc;
  } else {
    if ((rc = sp_wire_scalars(w, M->data, nnz, 1)) || (rc = sp_wire_u32s_as_u64(w, M->indices, nnz, 1)) || (rc = sp_wire_u64s(w, M->indptr, rows + 1, 1))) return rc;
    w->u64(cols);
  }
  return SP_OK;
}
// SplitR1CSShape: digest_form != 0 -> write_bytes (src/r1cs/mod.rs:775-794), else the derived Serialize (:742-773); same ten dimensions first
int sp_wire_shape(sp_wire* w, const sp_dims* d, const sp_csr* A, const sp_csr* Bm, const sp_csr* C, int digest_form) {
  if (!w || !d || !A || !Bm || !C) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_shape: null argument");
  const uint64_t dims[10] = {d->num_cons, d->num_cons_unpadded, d->num_shared_unpadded, d->num_precommitted_unpadded, d->num_rest_unpadded,
                             d->num_shared, d->num_precommitted, d->num_rest,           d->num_public,                d->num_challenges};
  w->put(dims, sizeof dims);
  const size_t cols = d->num_shared + d->num_precommitted + d->num_rest + 1 + d->num_public + d->num_challenges;  // src/r1cs/mod.rs:824
  for (const sp_csr* M : {A, Bm, C}) {
    int rc = sp_wire_matrix(w, M, d->num_cons, cols, digest_form);
    if (rc) return rc;
  }
  return SP_OK;
}
size_t sp_wire_len(const sp_wire* w) { return w ? (size_t)w->written : 0; }
int sp_wire_bytes(sp_wire* w, uint8_t* out, size_t cap) {
  if (!w || w->hashing) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_bytes: not a collecting sink");
  w->flush();
  if (cap < w->buf.size() || (!out && !w->buf.empty())) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_bytes: buffer too small");
  if (!w->buf.empty()) memcpy(out, w->buf.data(), w->buf.size());
  return SP_OK;
}
int sp_wire_digest(sp_wire* w, uint8_t out[32]) {  // consumes the hasher
  if (!w || !w->hashing || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_wire_digest: not a hashing sink");
  w->flush();
  w->sha.finish(out);
  w->sha = sp::Sha256();
  return SP_OK;
}

// ---- source ------------------------------------------------------------------------------------------------------------------------------
int sp_unwire_new(const uint8_t* bytes, size_t n, sp_unwire** out) {
  if (!out || (!bytes && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_unwire_new: null argument");
  *out = new sp_unwire{bytes, bytes + n};
  return SP_OK;
}
void sp_unwire_free(sp_unwire* r) { delete r; }
size_t sp_unwire_left(const sp_unwire* r) { return r ? r->left() : 0; }
int sp_unwire_u8(sp_unwire* r, uint8_t* out) {
  if (!r || r->left() < 1) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: unexpected end of input");
  *out = *r->p++;
  return SP_OK;
}
int sp_unwire_u64s(sp_unwire* r, size_t n, uint64_t* out) {
  if (!r || r->left() / 8 < n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: unexpected end of input");
  memcpy(out, r->p, 8 * n);
  r->p += 8 * n;
  return SP_OK;
}
// a Vec length prefix, refused when the remaining input cannot hold that many elements of at least elem_bytes each
int sp_unwire_len(sp_unwire* r, size_t elem_bytes, size_t* out) {
  uint64_t n;
  int rc = sp_unwire_u64s(r, 1, &n);
  if (rc) return rc;
  if (elem_bytes && n > r->left() / elem_bytes) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: length prefix exceeds the input");
  *out = (size_t)n;
  return SP_OK;
}
int sp_unwire_scalars(sp_unwire* r, size_t n, uint64_t* out) {
  if (!r || r->left() / 32 < n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: unexpected end of input");
  for (size_t i = 0; i < n; ++i) {
    fe_t f;
    if (!read_fe<SC>(r, &f)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: non-canonical field element");
    memcpy(out + 4 * i, f.v, 32);
  }
  return SP_OK;
}
int sp_unwire_affines(sp_unwire* r, size_t n, uint64_t* out) {
  if (!r || r->left() / 64 < n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: unexpected end of input");
  for (size_t i = 0; i < n; ++i) {
    aff_t a;
    if (!read_fe<B>(r, &a.x) || !read_fe<B>(r, &a.y)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: non-canonical coordinate");
    if (!aff_on_curve(a)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: point not on the curve");
    memcpy(out + 8 * i, a.x.v, 32);
    memcpy(out + 8 * i + 4, a.y.v, 32);
  }
  return SP_OK;
}
int sp_unwire_points(sp_unwire* r, size_t n, uint64_t* out) {  // {x, y, z}: any representative -> affine, (0,0) for the identity
  if (!r || r->left() / 96 < n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: unexpected end of input");
  for (size_t i = 0; i < n; ++i) {
    jac_t j;
    if (!read_fe<B>(r, &j.x) || !read_fe<B>(r, &j.y) || !read_fe<B>(r, &j.z)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: non-canonical coordinate");
    aff_t a;
    if (fe_is_zero(j.z)) {
      a.x = fe_zero(), a.y = fe_zero();
    } else if (fe_eq(j.z, fe_one<B>())) {
      a.x = j.x, a.y = j.y;
    } else {
      a = jac_to_affine(j);
    }
    if (!aff_on_curve(a)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: point not on the curve");
    memcpy(out + 8 * i, a.x.v, 32);
    memcpy(out + 8 * i + 4, a.y.v, 32);
  }
  return SP_OK;
}
int sp_unwire_done(const sp_unwire* r) {  // bincode's DefaultOptions reject trailing bytes
  if (!r || r->left() != 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: trailing bytes");
  return SP_OK;
}

// ---- SpartanVerifierKey digest (src/spartan.rs:73-104) ---------------------------------------------------------------------------------------
int sp_vk_digest(const sp_dims* dims, const sp_csr* A, const sp_csr* Bm, const sp_csr* C, const uint64_t* ck, size_t num_cols, const uint64_t* h,
                 const uint64_t* ck_s, size_t num_cols_s, const uint64_t* h_s, uint8_t out[32]) {
  if (!out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_vk_digest: null out");
  sp_wire* w = nullptr;
  sp_wire_new(1, &w);
  int rc = sp_wire_hyrax_key(w, ck, num_cols, h);            // vk_ee
  if (!rc) rc = sp_wire_hyrax_key(w, ck_s, num_cols_s, h_s);  // ck_s
  if (!rc) rc = sp_wire_shape(w, dims, A, Bm, C, 1);          // S.write_bytes()
  if (!rc) rc = sp_wire_digest(w, out);
  sp_wire_free(w);
  return rc;
}

// ---- SpartanSNARK (src/spartan.rs:125-137) <-> the flat word layout of DESIGN.md section 4 ----------------------------------------------------
static size_t spartan_flat_words(const sp_spartan_layout* L) {
  return 8 * (L->rows_shared + L->rows_precommitted + L->rows_rest) + 4 * (L->num_public + L->num_challenges) + 12 * L->rounds_x + 12 + 8 * L->rounds_y +
         8 + 16 + 4 * L->z_len + 8;
}
size_t sp_proof_words(const sp_spartan_layout* L) { return L ? spartan_flat_words(L) : 0; }
int sp_proof_serialize(const sp_spartan_layout* L, const uint64_t* words, size_t nwords, uint8_t* out, size_t cap, size_t* len) {
  if (!L || !words || !len) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_proof_serialize: null argument");
  if (nwords != spartan_flat_words(L)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_proof_serialize: the word count does not match the layout");
  sp_wire* w = nullptr;
  int rc = sp_wire_new(0, &w);
  if (rc) return rc;
  const uint64_t* p = words;
  auto keep = [&rc](int r) {  // the first failure of a nested sink call is the call's result: a truncated stream is never returned as a proof
    if (!rc) rc = r;
  };
  auto option_commitment = [&](size_t rows) {  // Option<HyraxCommitment { comm: Vec<GE> }>: Some exactly when the segment has rows
    keep(sp_wire_u8(w, rows ? 1 : 0));
    if (rows) keep(sp_wire_points(w, p, rows, 1));
    p += 8 * rows;
  };
  auto scalars = [&](size_t n, int with_len) {
    keep(sp_wire_scalars(w, p, n, with_len));
    p += 4 * n;
  };
  auto sumcheck = [&](size_t rounds, size_t per) {  // Vec<CompressedUniPoly { coeffs_except_linear_term: Vec<Scalar> }>
    w->u64(rounds);
    for (size_t i = 0; i < rounds; ++i) scalars(per, 1);
  };
  // U: SplitR1CSInstance (src/r1cs/mod.rs:797-806)
  option_commitment(L->rows_shared);
  option_commitment(L->rows_precommitted);
  keep(sp_wire_points(w, p, L->rows_rest, 1));
  p += 8 * L->rows_rest;
  scalars(L->num_public, 1);
  scalars(L->num_challenges, 1);
  sumcheck(L->rounds_x, 3);
  scalars(3, 0);  // claims_outer: a tuple
  sumcheck(L->rounds_y, 2);
  scalars(1, 0);  // eval_W
  scalars(1, 1);  // blind_eval_W: HyraxBlind { blind: Vec<Scalar> }, one row
  // eval_arg: HyraxEvaluationArgument { ipa: InnerProductArgumentLinear { delta, beta, z_vec, z_delta, z_beta } } (src/provider/pcs/ipa.rs:103-114)
  keep(sp_wire_points(w, p, 2, 0));
  p += 16;
  scalars(L->z_len, 1);
  scalars(2, 0);
  *len = sp_wire_len(w);
  if (!rc && out) rc = sp_wire_bytes(w, out, cap);
  sp_wire_free(w);
  return rc;
}
int sp_proof_deserialize(const uint8_t* bytes, size_t n, sp_spartan_layout* L, uint64_t* words, size_t cap_words, size_t* nwords) {
  if (!L || !nwords || (!bytes && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_proof_deserialize: null argument");
  sp_unwire r{bytes, bytes + n};
  std::vector<uint64_t> out;
  auto points = [&](size_t cnt) {
    out.resize(out.size() + 8 * cnt);
    return sp_unwire_points(&r, cnt, out.data() + out.size() - 8 * cnt);
  };
  auto scalars = [&](size_t cnt) {
    out.resize(out.size() + 4 * cnt);
    return sp_unwire_scalars(&r, cnt, out.data() + out.size() - 4 * cnt);
  };
  int rc;
#define WIRE_TRY(e) \
  if ((rc = (e)) != SP_OK) return rc
  auto commitment = [&](uint64_t* rows) -> int {
    size_t cnt;
    int rc2 = sp_unwire_len(&r, 96, &cnt);
    if (rc2) return rc2;
    *rows = cnt;
    return points(cnt);
  };
  auto option_commitment = [&](uint64_t* rows) -> int {
    uint8_t tag;
    int rc2 = sp_unwire_u8(&r, &tag);
    if (rc2) return rc2;
    if (tag > 1) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: bad Option tag");
    *rows = 0;
    if (!tag) return SP_OK;
    rc2 = commitment(rows);
    if (!rc2 && *rows == 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: Some(commitment) without rows");
    return rc2;
  };
  auto vec_scalars = [&](uint64_t* cnt_out) -> int {
    size_t cnt;
    int rc2 = sp_unwire_len(&r, 32, &cnt);
    if (rc2) return rc2;
    *cnt_out = cnt;
    return scalars(cnt);
  };
  auto sumcheck = [&](uint64_t* rounds, size_t per) -> int {
    size_t cnt;
    int rc2 = sp_unwire_len(&r, 8 + 32 * per, &cnt);
    if (rc2) return rc2;
    *rounds = cnt;
    for (size_t i = 0; i < cnt; ++i) {
      uint64_t k;
      if ((rc2 = vec_scalars(&k))) return rc2;
      if (k != per) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: compressed polynomial of the wrong degree");
    }
    return SP_OK;
  };
  WIRE_TRY(option_commitment(&L->rows_shared));
  WIRE_TRY(option_commitment(&L->rows_precommitted));
  WIRE_TRY(commitment(&L->rows_rest));
  WIRE_TRY(vec_scalars(&L->num_public));
  WIRE_TRY(vec_scalars(&L->num_challenges));
  WIRE_TRY(sumcheck(&L->rounds_x, 3));
  WIRE_TRY(scalars(3));
  WIRE_TRY(sumcheck(&L->rounds_y, 2));
  WIRE_TRY(scalars(1));
  uint64_t one_row;
  WIRE_TRY(vec_scalars(&one_row));
  if (one_row != 1) return fail(SP_ERR_INVALID_INPUT_LENGTH, "wire: blind_eval_W must hold one row");
  WIRE_TRY(points(2));
  WIRE_TRY(vec_scalars(&L->z_len));
  WIRE_TRY(scalars(2));
  WIRE_TRY(sp_unwire_done(&r));
#undef WIRE_TRY
  *nwords = out.size();
  if (words) {
    if (cap_words < out.size()) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_proof_deserialize: word buffer too small");
    memcpy(words, out.data(), 8 * out.size());
  }
  return SP_OK;
}

}  // extern "C"
