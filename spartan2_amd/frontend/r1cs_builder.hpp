// Integer R1CS builder + SHA-256 circuit generator (CPU frontend; input generation for the prover,
// not part of the accelerated path and independent of both the HIP library and the oracle).
//
// Stands in for the reference's bellpepper frontend (src/bellpepper/r1cs.rs:134-287 ShapeCS ->
// SplitR1CSShape, solver.rs SatisfyingAssignment) and the third-party gadget
// bellpepper::gadgets::sha256 (bellpepper 0.4.0, NOT under /root/reference; call site
// benches/sha256_spartan.rs:99). The gadget's constraint/variable ORDER is third-party and
// unpinned; this generator reproduces its published construction (Boolean XOR/AND/ch/maj
// constraints, UInt32::addmany with deferred additions, constant folding) and is self-checked
// against a bit-level SHA-256 plus the reference's "~26,352 constraints per compression" note
// (benches/sha256_neutronnova.rs:159).
//
// Output convention (== what add_constraint emits, r1cs.rs:234-287): column j < num_aux is aux
// variable j; column num_aux + i is input i, input 0 being the constant ONE. Coefficients and
// witness values are plain int64/uint64 (everything in these circuits is a bit or a small
// power-of-two multiple), so no field arithmetic lives here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

namespace sp_frontend {

static const uint32_t INPUT_FLAG = 0x80000000u;

struct Term {
  uint32_t var;  // aux index, or INPUT_FLAG | input index (input 0 == ONE)
  int64_t coeff;
};
typedef std::vector<Term> LC;

struct CsrInt {
  std::vector<int64_t> data;
  std::vector<uint32_t> indices;  // pre-padding column ids, inputs still flagged
  std::vector<uint64_t> indptr{0};
};

struct ConstraintSystem {
  std::vector<uint64_t> aux;     // aux assignment
  std::vector<uint64_t> inputs;  // input assignment, inputs[0] == 1
  CsrInt A, B, C;
  size_t num_constraints = 0;
  // bookkeeping for the comparison with bellpepper's count (see MultiEqSim): equality rows emitted by UInt32::addmany inside SHA-256 compressions, and
  // the rows bellpepper's MultiEq would have packed them into
  size_t stat_addmany_rows = 0, stat_multieq_rows = 0;
  ConstraintSystem() { inputs.push_back(1); }

  uint32_t alloc_aux(uint64_t v) {
    aux.push_back(v);
    return (uint32_t)(aux.size() - 1);
  }
  uint32_t alloc_input(uint64_t v) {
    inputs.push_back(v);
    return INPUT_FLAG | (uint32_t)(inputs.size() - 1);
  }
  static uint32_t one() { return INPUT_FLAG | 0; }

  static void normalize(LC& lc) {  // merge duplicate variables, drop zero coefficients
    std::sort(lc.begin(), lc.end(), [](const Term& a, const Term& b) { return a.var < b.var; });
    size_t w = 0;
    for (size_t i = 0; i < lc.size();) {
      Term t = lc[i];
      size_t j = i + 1;
      while (j < lc.size() && lc[j].var == t.var) t.coeff += lc[j++].coeff;
      if (t.coeff != 0) lc[w++] = t;
      i = j;
    }
    lc.resize(w);
  }
  static void push(CsrInt& M, LC lc) {
    normalize(lc);
    for (const Term& t : lc) {
      M.data.push_back(t.coeff);
      M.indices.push_back(t.var);
    }
    M.indptr.push_back(M.indices.size());
  }
  void enforce(const LC& a, const LC& b, const LC& c) {
    push(A, a);
    push(B, b);
    push(C, c);
    ++num_constraints;
  }
  uint64_t value(uint32_t var) const { return (var & INPUT_FLAG) ? inputs[var & ~INPUT_FLAG] : aux[var]; }
  // check A z o B z == C z over the integers mod 2^64 is NOT sound for big coefficients, so callers
  // use is_satisfied only on these small circuits (all intermediate sums < 2^40).
  bool is_satisfied() const {
    for (size_t r = 0; r < num_constraints; ++r) {
      auto dot = [&](const CsrInt& M) {
        __int128 s = 0;
        for (uint64_t k = M.indptr[r]; k < M.indptr[r + 1]; ++k) s += (__int128)M.data[k] * (__int128)value(M.indices[k]);
        return s;
      };
      if (dot(A) * dot(B) != dot(C)) return false;
    }
    return true;
  }
};

// ---- Boolean gadget (bellpepper-core boolean.rs construction) ---------------------------------
struct Boolean {
  enum Kind { Const, Is, Not } kind;
  uint32_t var;  // valid for Is/Not
  bool cval;     // valid for Const
  bool val;      // assigned value
  static Boolean constant(bool b) { return Boolean{Const, 0, b, b}; }
  Boolean negate() const {
    if (kind == Const) return constant(!cval);
    return Boolean{kind == Is ? Not : Is, var, false, !val};
  }
  // add coeff * self to an LC
  void lc_add(LC& lc, int64_t coeff) const {
    if (kind == Const) {
      if (cval) lc.push_back(Term{ConstraintSystem::one(), coeff});
    } else if (kind == Is) {
      lc.push_back(Term{var, coeff});
    } else {
      lc.push_back(Term{ConstraintSystem::one(), coeff});
      lc.push_back(Term{var, -coeff});
    }
  }
};

inline Boolean alloc_bit(ConstraintSystem& cs, bool v) {  // AllocatedBit::alloc: (1 - a) * a = 0
  uint32_t var = cs.alloc_aux(v ? 1 : 0);
  cs.enforce({{ConstraintSystem::one(), 1}, {var, -1}}, {{var, 1}}, {});
  return Boolean{Boolean::Is, var, false, v};
}
inline Boolean alloc_bit_unchecked(ConstraintSystem& cs, bool v) {
  uint32_t var = cs.alloc_aux(v ? 1 : 0);
  return Boolean{Boolean::Is, var, false, v};
}

inline Boolean bool_xor(ConstraintSystem& cs, const Boolean& a, const Boolean& b) {
  if (a.kind == Boolean::Const) return a.cval ? b.negate() : b;
  if (b.kind == Boolean::Const) return b.cval ? a.negate() : a;
  // reduce Not/Not and Is/Not to the Is/Is case
  bool flip = (a.kind == Boolean::Not) ^ (b.kind == Boolean::Not);
  bool av = a.kind == Boolean::Is ? a.val : !a.val, bv = b.kind == Boolean::Is ? b.val : !b.val;
  Boolean c = alloc_bit_unchecked(cs, av ^ bv);
  // (a + a) * b = a + b - c
  cs.enforce({{a.var, 2}}, {{b.var, 1}}, {{a.var, 1}, {b.var, 1}, {c.var, -1}});
  return flip ? c.negate() : c;
}

inline Boolean bool_and(ConstraintSystem& cs, const Boolean& a, const Boolean& b) {
  if (a.kind == Boolean::Const) return a.cval ? b : Boolean::constant(false);
  if (b.kind == Boolean::Const) return b.cval ? a : Boolean::constant(false);
  Boolean c = alloc_bit_unchecked(cs, a.val && b.val);
  LC la, lb;
  a.lc_add(la, 1);
  b.lc_add(lb, 1);
  cs.enforce(la, lb, {{c.var, 1}});  // and / and_not / nor in one form
  return c;
}

// ch(a,b,c) = (a & b) ^ (!a & c): a * (b - c) = ch - c
inline Boolean sha256_ch(ConstraintSystem& cs, const Boolean& a, const Boolean& b, const Boolean& c) {
  if (a.kind == Boolean::Const) return a.cval ? b : c;
  if (b.kind == Boolean::Const && c.kind == Boolean::Const) {
    if (b.cval == c.cval) return b;
    return b.cval ? a : a.negate();
  }
  if (b.kind == Boolean::Const) {  // bellpepper falls back to and/xor combinations here
    if (b.cval) return bool_and(cs, a.negate(), c.negate()).negate();  // a | c
    return bool_and(cs, a.negate(), c);
  }
  if (c.kind == Boolean::Const) {
    if (c.cval) return bool_and(cs, a, b.negate()).negate();  // !a | b
    return bool_and(cs, a, b);
  }
  bool v = (a.val && b.val) ^ (!a.val && c.val);
  Boolean ch = alloc_bit_unchecked(cs, v);
  LC la, lb, lc;
  a.lc_add(la, 1);
  b.lc_add(lb, 1);
  c.lc_add(lb, -1);
  lc.push_back({ch.var, 1});
  c.lc_add(lc, -1);
  cs.enforce(la, lb, lc);
  return ch;
}

// maj(a,b,c): bc = b & c ; a * (b + c - 2 bc) = maj - bc
inline Boolean sha256_maj(ConstraintSystem& cs, const Boolean& a, const Boolean& b, const Boolean& c) {
  if (a.kind == Boolean::Const) return a.cval ? bool_and(cs, b.negate(), c.negate()).negate() : bool_and(cs, b, c);
  if (b.kind == Boolean::Const) return b.cval ? bool_and(cs, a.negate(), c.negate()).negate() : bool_and(cs, a, c);
  if (c.kind == Boolean::Const) return c.cval ? bool_and(cs, a.negate(), b.negate()).negate() : bool_and(cs, a, b);
  Boolean bc = bool_and(cs, b, c);
  bool v = (a.val && b.val) ^ (a.val && c.val) ^ (b.val && c.val);
  Boolean maj = alloc_bit_unchecked(cs, v);
  LC la, lb, lc;
  a.lc_add(la, 1);
  b.lc_add(lb, 1);
  c.lc_add(lb, 1);
  bc.lc_add(lb, -2);
  lc.push_back({maj.var, 1});
  bc.lc_add(lc, -1);
  cs.enforce(la, lb, lc);
  return maj;
}

// ---- UInt32 gadget ------------------------------------------------------------------------------
struct UInt32 {
  Boolean bits[32];  // little-endian
  uint32_t value() const {
    uint32_t v = 0;
    for (int i = 0; i < 32; ++i) v |= (uint32_t)bits[i].val << i;
    return v;
  }
  bool is_constant() const {
    for (int i = 0; i < 32; ++i)
      if (bits[i].kind != Boolean::Const) return false;
    return true;
  }
  static UInt32 constant(uint32_t v) {
    UInt32 u;
    for (int i = 0; i < 32; ++i) u.bits[i] = Boolean::constant((v >> i) & 1);
    return u;
  }
  static UInt32 from_bits_be(const Boolean* b) {  // b[0] is the MSB
    UInt32 u;
    for (int i = 0; i < 32; ++i) u.bits[i] = b[31 - i];
    return u;
  }
  void into_bits_be(Boolean* out) const {
    for (int i = 0; i < 32; ++i) out[i] = bits[31 - i];
  }
  UInt32 rotr(int by) const {
    UInt32 u;
    for (int i = 0; i < 32; ++i) u.bits[i] = bits[(i + by) % 32];
    return u;
  }
  UInt32 shr(int by) const {
    UInt32 u;
    for (int i = 0; i < 32; ++i) u.bits[i] = (i + by < 32) ? bits[i + by] : Boolean::constant(false);
    return u;
  }
};

inline UInt32 u32_xor(ConstraintSystem& cs, const UInt32& a, const UInt32& b) {
  UInt32 u;
  for (int i = 0; i < 32; ++i) u.bits[i] = bool_xor(cs, a.bits[i], b.bits[i]);
  return u;
}

// bellpepper's MultiEq (gadgets/multieq.rs; used by sha256_compression_function) packs the equality constraints of successive UInt32::addmany calls
// into ONE row while the accumulated bit widths stay below the field's CAPACITY (255 for the 256-bit scalar field of the bench engine):
// enforce_equal(num_bits, lhs, rhs) adds 2^bits_used * (lhs - rhs) to the pending row and flushes it first when CAPACITY <= bits_used + num_bits; the
// last row is flushed when the compression ends. Packed rows carry coefficients up to 2^254, which this integer frontend (int64 coefficients) cannot
// represent, so the generator keeps ONE row per addmany — the same equalities, unpacked — and only SIMULATES the packing to count the rows the
// reference's synthesizer would emit: reference count = num_constraints - (stat_addmany_rows - stat_multieq_rows).
struct MultiEqSim {
  static constexpr size_t CAPACITY = 255;
  size_t bits_used = 0, rows = 0;
  void enforce_equal(size_t num_bits) {
    if (CAPACITY <= bits_used + num_bits) flush();
    bits_used += num_bits;
  }
  void flush() {
    if (bits_used > 0) ++rows;
    bits_used = 0;
  }
};

// UInt32::addmany: sum operands as an LC, allocate the result bits (with carries), one equality (through `me`, when given, as bellpepper does).
inline UInt32 u32_addmany(ConstraintSystem& cs, const std::vector<UInt32>& ops, MultiEqSim* me = nullptr) {
  bool all_const = true;
  uint64_t sum = 0, max_value = (uint64_t)ops.size() * 0xffffffffULL;
  LC lc;
  for (const UInt32& op : ops) {
    sum += op.value();
    if (!op.is_constant()) all_const = false;
    for (int i = 0; i < 32; ++i) op.bits[i].lc_add(lc, (int64_t)1 << i);
  }
  if (all_const) return UInt32::constant((uint32_t)sum);
  LC res_lc;
  UInt32 out;
  int i = 0;
  while (max_value != 0) {
    Boolean b = alloc_bit(cs, (sum >> i) & 1);
    res_lc.push_back({b.var, (int64_t)1 << i});
    if (i < 32) out.bits[i] = b;
    max_value >>= 1;
    ++i;
  }
  cs.enforce(lc, {{ConstraintSystem::one(), 1}}, res_lc);
  if (me) {
    me->enforce_equal((size_t)i);
    ++cs.stat_addmany_rows;
  }
  return out;
}

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t SHA256_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

// one compression: 512 input bits (big-endian words) + current state -> new state. Statement order, deferred additions (`Maybe`) and the operand lists
// are those of bellpepper::gadgets::sha256::sha256_compression_function (bellpepper 0.4.0, a fork of bellman's gadget; the crate is not under
// /root/reference): the working variables a and e are kept as operand lists and only become bits at the start of the next round — or, after the last
// round, together with the chaining value they are added to (h0 and h4 are ONE addition of 8 / 7 operands each).
inline void sha256_compression(ConstraintSystem& cs, const Boolean* input512, UInt32 state[8]) {
  MultiEqSim me;
  std::vector<UInt32> w(64);
  for (int i = 0; i < 16; ++i) w[i] = UInt32::from_bits_be(input512 + 32 * i);
  for (int i = 16; i < 64; ++i) {
    UInt32 s0 = u32_xor(cs, u32_xor(cs, w[i - 15].rotr(7), w[i - 15].rotr(18)), w[i - 15].shr(3));
    UInt32 s1 = u32_xor(cs, u32_xor(cs, w[i - 2].rotr(17), w[i - 2].rotr(19)), w[i - 2].shr(10));
    w[i] = u32_addmany(cs, {w[i - 16], s0, w[i - 7], s1}, &me);
  }
  struct Maybe {  // Deferred(operands) | Concrete(value)
    bool deferred = false;
    std::vector<UInt32> ops;
    UInt32 value;
    UInt32 compute(ConstraintSystem& cs, MultiEqSim& me, const std::vector<UInt32>& others) {
      if (!deferred) return value;
      std::vector<UInt32> v = ops;
      v.insert(v.end(), others.begin(), others.end());
      return u32_addmany(cs, v, &me);
    }
  };
  Maybe a, e;
  a.value = state[0];
  e.value = state[4];
  UInt32 b = state[1], c = state[2], d = state[3], f = state[5], g = state[6], h = state[7];
  for (int i = 0; i < 64; ++i) {
    const UInt32 new_e = e.compute(cs, me, {});
    UInt32 s1 = u32_xor(cs, u32_xor(cs, new_e.rotr(6), new_e.rotr(11)), new_e.rotr(25));
    UInt32 ch;
    for (int k = 0; k < 32; ++k) ch.bits[k] = sha256_ch(cs, new_e.bits[k], f.bits[k], g.bits[k]);
    const std::vector<UInt32> temp1 = {h, s1, ch, UInt32::constant(SHA256_K[i]), w[i]};
    const UInt32 new_a = a.compute(cs, me, {});
    UInt32 s0 = u32_xor(cs, u32_xor(cs, new_a.rotr(2), new_a.rotr(13)), new_a.rotr(22));
    UInt32 maj;
    for (int k = 0; k < 32; ++k) maj.bits[k] = sha256_maj(cs, new_a.bits[k], b.bits[k], c.bits[k]);
    h = g;
    g = f;
    f = new_e;
    e.deferred = true;
    e.ops = {d};
    e.ops.insert(e.ops.end(), temp1.begin(), temp1.end());
    d = c;
    c = b;
    b = new_a;
    a.deferred = true;
    a.ops = temp1;
    a.ops.push_back(s0);
    a.ops.push_back(maj);
  }
  UInt32 out[8];
  out[0] = a.compute(cs, me, {state[0]});
  out[1] = u32_addmany(cs, {state[1], b}, &me);
  out[2] = u32_addmany(cs, {state[2], c}, &me);
  out[3] = u32_addmany(cs, {state[3], d}, &me);
  out[4] = e.compute(cs, me, {state[4]});
  out[5] = u32_addmany(cs, {state[5], f}, &me);
  out[6] = u32_addmany(cs, {state[6], g}, &me);
  out[7] = u32_addmany(cs, {state[7], h}, &me);
  for (int i = 0; i < 8; ++i) state[i] = out[i];
  me.flush();
  cs.stat_multieq_rows += me.rows;
}

// bellpepper::gadgets::sha256::sha256: pad, iterate compressions, output 256 bits big-endian
inline std::vector<Boolean> sha256_gadget(ConstraintSystem& cs, const std::vector<Boolean>& input) {
  if (input.size() % 8) throw std::runtime_error("sha256: input must be whole bytes");
  std::vector<Boolean> padded = input;
  uint64_t plen = padded.size();
  padded.push_back(Boolean::constant(true));
  while ((padded.size() + 64) % 512 != 0) padded.push_back(Boolean::constant(false));
  for (int i = 63; i >= 0; --i) padded.push_back(Boolean::constant((plen >> i) & 1));
  UInt32 state[8];
  for (int i = 0; i < 8; ++i) state[i] = UInt32::constant(SHA256_IV[i]);
  for (size_t blk = 0; blk < padded.size() / 512; ++blk) sha256_compression(cs, padded.data() + 512 * blk, state);
  std::vector<Boolean> out(256);
  for (int i = 0; i < 8; ++i) state[i].into_bits_be(out.data() + 32 * i);
  return out;
}

// plain SHA-256 for the self-check
inline void sha256_plain(const uint8_t* msg, size_t len, uint8_t out[32]) {
  std::vector<uint8_t> m(msg, msg + len);
  m.push_back(0x80);
  while ((m.size() + 8) % 64) m.push_back(0);
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 7; i >= 0; --i) m.push_back((uint8_t)(bits >> (8 * i)));
  uint32_t H[8];
  for (int i = 0; i < 8; ++i) H[i] = SHA256_IV[i];
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  for (size_t blk = 0; blk < m.size(); blk += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)m[blk + 4 * i] << 24 | (uint32_t)m[blk + 4 * i + 1] << 16 | (uint32_t)m[blk + 4 * i + 2] << 8 | m[blk + 4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
    for (int i = 0; i < 64; ++i) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = h + S1 + ch + SHA256_K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    H[0] += a; H[1] += b; H[2] += c; H[3] += d; H[4] += e; H[5] += f; H[6] += g; H[7] += h;
  }
  for (int i = 0; i < 8; ++i)
    for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint8_t)(H[i] >> (24 - 8 * k));
}

// The integer R1CS instance handed to setup (== the arguments of SplitR1CSShape::new,
// src/r1cs/mod.rs:810-820, before padding) plus the satisfying assignment.
struct R1CSInstanceInt {
  size_t num_cons = 0, num_shared = 0, num_precommitted = 0, num_rest = 0, num_public = 0, num_challenges = 0;
  CsrInt A, B, C;                 // indices already mapped: aux j -> j, input i -> num_aux + i
  std::vector<uint64_t> witness;  // aux assignment (shared | precommitted | rest), unpadded
  std::vector<uint64_t> publics;  // input assignment without ONE
  size_t stat_addmany_rows = 0, stat_multieq_rows = 0;  // see MultiEqSim
};

inline R1CSInstanceInt finalize(ConstraintSystem& cs, size_t num_shared, size_t num_precommitted) {
  R1CSInstanceInt R;
  size_t num_aux = cs.aux.size();
  R.num_cons = cs.num_constraints;
  R.num_shared = num_shared;
  R.num_precommitted = num_precommitted;
  R.num_rest = num_aux - num_shared - num_precommitted;
  R.num_public = cs.inputs.size() - 1;
  R.stat_addmany_rows = cs.stat_addmany_rows;
  R.stat_multieq_rows = cs.stat_multieq_rows;
  auto remap = [&](CsrInt& M) {
    for (uint32_t& c : M.indices) c = (c & INPUT_FLAG) ? (uint32_t)(num_aux + (c & ~INPUT_FLAG)) : c;
  };
  remap(cs.A);
  remap(cs.B);
  remap(cs.C);
  R.A = std::move(cs.A);
  R.B = std::move(cs.B);
  R.C = std::move(cs.C);
  R.witness = std::move(cs.aux);
  R.publics.assign(cs.inputs.begin() + 1, cs.inputs.end());
  return R;
}

// Sha256Circuit of benches/sha256_spartan.rs:36-152: all preimage bits (MSB first per byte) are
// precommitted witness bits; digest bits are the 256 public inputs tied by `bit * 1 = num`.
inline R1CSInstanceInt sha256_spartan_circuit(const std::vector<uint8_t>& preimage) {
  ConstraintSystem cs;
  std::vector<Boolean> bits;
  for (uint8_t byte : preimage)
    for (int i = 7; i >= 0; --i) bits.push_back(alloc_bit(cs, (byte >> i) & 1));
  std::vector<Boolean> hash = sha256_gadget(cs, bits);
  uint8_t expect[32];
  sha256_plain(preimage.data(), preimage.size(), expect);
  for (int i = 0; i < 256; ++i) {
    bool e = (expect[i / 8] >> (7 - i % 8)) & 1;
    if (hash[i].val != e) throw std::runtime_error("sha256 gadget disagrees with plain SHA-256");
    uint32_t n = cs.alloc_input(e ? 1 : 0);
    LC la;
    hash[i].lc_add(la, 1);
    cs.enforce(la, {{ConstraintSystem::one(), 1}}, {{n, 1}});
  }
  size_t num_aux = cs.aux.size();
  return finalize(cs, 0, num_aux);  // shared = 0, everything precommitted, rest = 0 (benches :71-76,139-151)
}

// Sha256StepCircuit / CoreCircuit of benches/sha256_neutronnova.rs:49-183: the 512 block bits (MSB first per byte) are precommitted witness bits,
// the chaining value is the constant IV, ONE compression (no padding block), then x = 0 allocated and inputized (AllocatedNum::inputize:
// input * 1 = x). The core circuit is the same shape on 512 zero bits (:161-182).
inline R1CSInstanceInt sha256_step_circuit(const uint8_t block[64]) {
  ConstraintSystem cs;
  std::vector<Boolean> bits;
  for (int k = 0; k < 64; ++k)
    for (int i = 7; i >= 0; --i) bits.push_back(alloc_bit(cs, (block[k] >> i) & 1));
  UInt32 state[8];
  for (int i = 0; i < 8; ++i) state[i] = UInt32::constant(SHA256_IV[i]);
  sha256_compression(cs, bits.data(), state);
  const uint32_t x = cs.alloc_aux(0);
  const uint32_t in = cs.alloc_input(0);
  cs.enforce({{in, 1}}, {{ConstraintSystem::one(), 1}}, {{x, 1}});
  size_t num_aux = cs.aux.size();
  return finalize(cs, 0, num_aux);
}

// Sha256Circuit of the reference's own NeutronNova test (src/neutronnova_zk.rs:2357-2418): nothing shared or precommitted — the whole circuit sits in
// SpartanCircuit::synthesize, so every variable is a REST variable: the preimage bits (LSB first per byte, :2396-2401), the sha256 gadget with padding, then
// x = 0 allocated and inputized. test_neutron_sha256 (:2480-2503) folds 2, 7, 32 and 64 of them over 32- and 64-byte preimages [i; len].
inline R1CSInstanceInt sha256_rest_circuit(const std::vector<uint8_t>& preimage) {
  ConstraintSystem cs;
  std::vector<Boolean> bits;
  for (uint8_t byte : preimage)
    for (int i = 0; i < 8; ++i) bits.push_back(alloc_bit(cs, (byte >> i) & 1));
  sha256_gadget(cs, bits);
  const uint32_t x = cs.alloc_aux(0);
  const uint32_t in = cs.alloc_input(0);
  cs.enforce({{in, 1}}, {{ConstraintSystem::one(), 1}}, {{x, 1}});
  return finalize(cs, 0, 0);
}

// Small seeded synthetic circuit in SHA-like proportions (SURVEY.md 8(d) fallback shapes):
// booleanity, AND, XOR and 32-bit pack rows over random bits; `n_groups` groups of 100 rows.
inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
// `shared_permille` / `precommitted_permille`: where to cut the aux list into shared | precommitted | rest segments (the
// segments of SpartanCircuit::{shared, precommitted, synthesize}, src/traits/circuit.rs); 0/1000 = everything precommitted.
// `witness_seed` != 0 draws the 64 free input bits from a second stream: same matrices, a different satisfying assignment
// (several instances of one step shape, as NeutronNova folds them).
inline R1CSInstanceInt synthetic_circuit(size_t n_groups, uint64_t seed, size_t num_public, unsigned shared_permille = 0,
                                         unsigned precommitted_permille = 1000, uint64_t witness_seed = 0) {
  ConstraintSystem cs;
  uint64_t s = seed, ws = witness_seed;
  std::vector<Boolean> pool;
  for (int i = 0; i < 64; ++i) {
    const uint64_t structural = splitmix64(s);
    pool.push_back(alloc_bit(cs, (witness_seed ? splitmix64(ws) : structural) & 1));
  }
  for (size_t g = 0; g < n_groups; ++g) {
    for (int k = 0; k < 30; ++k) {
      Boolean a = pool[splitmix64(s) % pool.size()], b = pool[splitmix64(s) % pool.size()];
      pool.push_back(bool_xor(cs, a, (splitmix64(s) & 1) ? b.negate() : b));
    }
    for (int k = 0; k < 20; ++k) {
      Boolean a = pool[splitmix64(s) % pool.size()], b = pool[splitmix64(s) % pool.size()];
      pool.push_back(bool_and(cs, (splitmix64(s) & 1) ? a.negate() : a, b));
    }
    UInt32 x, y;
    for (int i = 0; i < 32; ++i) {
      x.bits[i] = pool[splitmix64(s) % pool.size()];
      y.bits[i] = pool[splitmix64(s) % pool.size()];
    }
    UInt32 z = u32_addmany(cs, {x, y, UInt32::constant((uint32_t)splitmix64(s))});
    for (int i = 0; i < 32; ++i) pool.push_back(z.bits[i]);
    if (pool.size() > 4096) pool.erase(pool.begin(), pool.begin() + 2048);
  }
  for (size_t i = 0; i < num_public; ++i) {
    Boolean b = pool[pool.size() - 1 - i];
    uint32_t n = cs.alloc_input(b.val ? 1 : 0);
    LC la;
    b.lc_add(la, 1);
    cs.enforce(la, {{ConstraintSystem::one(), 1}}, {{n, 1}});
  }
  size_t num_aux = cs.aux.size();
  size_t ns = num_aux * shared_permille / 1000, np = num_aux * precommitted_permille / 1000;
  if (ns + np > num_aux) np = num_aux - ns;
  return finalize(cs, ns, np);
}

// CubicCircuit of the reference's end-to-end test (src/spartan.rs:587-651): x^3 + x + 5 = y with x = 2; every variable is
// allocated in `synthesize`, i.e. in the REST segment; one public output (15).
inline R1CSInstanceInt cubic_circuit() {
  ConstraintSystem cs;
  const uint32_t x = cs.alloc_aux(2), x_sq = cs.alloc_aux(4), x_cu = cs.alloc_aux(8), y = cs.alloc_aux(15);
  cs.enforce({{x, 1}}, {{x, 1}}, {{x_sq, 1}});        // AllocatedNum::square
  cs.enforce({{x_sq, 1}}, {{x, 1}}, {{x_cu, 1}});     // AllocatedNum::mul
  cs.enforce({{x_cu, 1}, {x, 1}, {ConstraintSystem::one(), 5}}, {{ConstraintSystem::one(), 1}}, {{y, 1}});  // y = x^3 + x + 5
  const uint32_t out = cs.alloc_input(15);             // AllocatedNum::inputize
  cs.enforce({{out, 1}}, {{ConstraintSystem::one(), 1}}, {{y, 1}});
  return finalize(cs, 0, 0);
}

}  // namespace sp_frontend
