// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Restates src/provider/msm.rs: cpu_msm_serial signed-digit Pippenger (:59-178), msm (:187-222),
// msm_small dispatch -> msm_binary / msm_10 / msm_small_rest (:367-620), FixedBaseMul 8-bit window
// tables (:637-774). Group results are compared as canonical affine points, so bucket order and the
// Bucket::{None,Affine,Projective} state machine (:24-57) are restated by value.
#pragma once
#include <array>
#include <cmath>
#include <vector>

#include "curve.hpp"

namespace oracle {

inline size_t msm_window_bits(size_t n) {  // msm.rs:60-66
  if (n < 4) return 1;
  if (n < 32) return 3;
  return (size_t)std::ceil(std::log((double)(uint32_t)n));
}

inline size_t get_at(size_t segment, size_t c, const uint8_t bytes[32]) {  // msm.rs:68-86
  size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
  if (skip_bytes >= 32) return 0;
  uint8_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < 8 && skip_bytes + i < 32; ++i) v[i] = bytes[skip_bytes + i];
  uint64_t tmp;
  memcpy(&tmp, v, 8);
  tmp >>= skip_bits - skip_bytes * 8;
  tmp %= (uint64_t)1 << c;
  return (size_t)tmp;
}

// msm.rs:59-178
inline Jac cpu_msm_serial(const Fq* coeffs, const Affine* bases, size_t len) {
  size_t c = msm_window_bits(len);
  Jac boolean_sum = Jac::identity();
  std::vector<std::array<uint8_t, 32>> reprs;
  std::vector<Affine> nb;
  Fq one = Fq::one();
  for (size_t i = 0; i < len; ++i) {
    if (coeffs[i] == one) {
      boolean_sum = boolean_sum.add_mixed(bases[i]);
    } else if (!coeffs[i].is_zero()) {
      std::array<uint8_t, 32> r;
      coeffs[i].to_repr(r.data());
      reprs.push_back(r);
      nb.push_back(bases[i]);
    }
  }
  if (reprs.empty()) return boolean_sum;
  size_t n = reprs.size();
  size_t segments = 256 / c + 1;
  size_t half = (size_t)1 << (c - 1), full = (size_t)1 << c;
  std::vector<int16_t> digits((segments + 1) * n, 0);
  std::vector<uint8_t> carry(n, 0);
  for (size_t seg = 0; seg < segments; ++seg) {
    for (size_t j = 0; j < n; ++j) {
      size_t raw = get_at(seg, c, reprs[j].data()) + carry[j];
      carry[j] = 0;
      if (raw >= half) {
        digits[seg * n + j] = (int16_t) - (int)(full - raw);
        carry[j] = 1;
      } else {
        digits[seg * n + j] = (int16_t)raw;
      }
    }
  }
  size_t total_segments = segments;
  bool any = false;
  for (size_t j = 0; j < n; ++j) any |= carry[j] != 0;
  if (any) {
    for (size_t j = 0; j < n; ++j) digits[segments * n + j] = carry[j];
    total_segments = segments + 1;
  }
  // window sums (independent, so they may run on several threads), then the Horner over windows (:150-175)
  std::vector<Jac> wsum(total_segments);
#pragma omp parallel for schedule(dynamic) if (n >= 256)
  for (size_t seg = 0; seg < total_segments; ++seg) {
    std::vector<Jac> buckets(half, Jac::identity());
    for (size_t j = 0; j < n; ++j) {
      int d = digits[seg * n + j];
      if (d > 0) buckets[d - 1] = buckets[d - 1].add_mixed(nb[j]);
      else if (d < 0) buckets[-d - 1] = buckets[-d - 1].add_mixed(affine_neg(nb[j]));
    }
    Jac running = Jac::identity(), w = Jac::identity();
    for (size_t k = half; k-- > 0;) {
      running = running.add(buckets[k]);
      w = w.add(running);
    }
    wsum[seg] = w;
  }
  Jac acc = Jac::identity();
  for (size_t seg = total_segments; seg-- > 0;) {
    for (size_t k = 0; k < c; ++k) acc = acc.dbl();
    acc = acc.add(wsum[seg]);
  }
  return boolean_sum.add(acc);
}

// msm.rs:187-222. `threads` restates rayon's chunk split (result is chunking-independent).
inline Jac msm(const Fq* coeffs, const Affine* bases, size_t n, size_t threads = 1) {
  if (threads <= 1 || n < 1024 || n <= threads) return cpu_msm_serial(coeffs, bases, n);
  size_t chunk = n / threads;
  Jac sum = Jac::identity();
  for (size_t s = 0; s < n; s += chunk) {
    size_t len = std::min(chunk, n - s);
    sum = sum.add(cpu_msm_serial(coeffs + s, bases + s, len));
  }
  return sum;
}

inline Jac msm_binary(const uint64_t* s, const Affine* bases, size_t n) {  // msm.rs:418-451
  Jac acc = Jac::identity();
  for (size_t i = 0; i < n; ++i)
    if (s[i]) acc = acc.add_mixed(bases[i]);
  return acc;
}
inline Jac msm_10(const uint64_t* s, const Affine* bases, size_t n, size_t max_bits) {  // msm.rs:454-502
  std::vector<Jac> buckets((size_t)1 << max_bits, Jac::identity());
  for (size_t i = 0; i < n; ++i)
    if (s[i]) buckets[s[i]] = buckets[s[i]].add_mixed(bases[i]);
  Jac result = Jac::identity(), running = Jac::identity();
  for (size_t k = buckets.size(); k-- > 1;) {
    running = running.add(buckets[k]);
    result = result.add(running);
  }
  return result;
}
inline size_t compute_ln(size_t a) {  // msm.rs:622-630
  if (a == 0) return 0;
  size_t lg = 0;
  while ((a >> (lg + 1)) != 0) ++lg;
  return lg * 69 / 100;
}
inline Jac msm_small_rest(const uint64_t* s, const Affine* bases, size_t n, size_t max_bits) {  // msm.rs:504-620
  size_t c = n < 32 ? 3 : compute_ln(n) + 2;
  std::vector<Jac> window_sums;
  for (size_t w_start = 0; w_start < max_bits; w_start += c) {
    Jac res = Jac::identity();
    std::vector<Jac> buckets(((size_t)1 << c) - 1, Jac::identity());
    for (size_t i = 0; i < n; ++i) {
      uint64_t sc = s[i];
      if (sc == 0) continue;
      if (sc == 1) {
        if (w_start == 0) res = res.add_mixed(bases[i]);
      } else {
        sc >>= w_start;
        sc %= (uint64_t)1 << c;
        if (sc) buckets[sc - 1] = buckets[sc - 1].add_mixed(bases[i]);
      }
    }
    Jac running = Jac::identity();
    for (size_t k = buckets.size(); k-- > 0;) {
      running = running.add(buckets[k]);
      res = res.add(running);
    }
    window_sums.push_back(res);
  }
  Jac lowest = window_sums.empty() ? Jac::identity() : window_sums[0];
  Jac total = Jac::identity();
  for (size_t k = window_sums.size(); k-- > 1;) {
    total = total.add(window_sums[k]);
    for (size_t d = 0; d < c; ++d) total = total.dbl();
  }
  return lowest.add(total);
}
inline size_t num_bits_of(uint64_t v) {
  size_t b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}
// msm.rs:367-409
inline Jac msm_small(const uint64_t* s, const Affine* bases, size_t n) {
  uint64_t mx = 0;
  for (size_t i = 0; i < n; ++i) mx = std::max(mx, s[i]);
  size_t mb = num_bits_of(mx);
  if (mb == 0) return Jac::identity();
  if (mb == 1) return msm_binary(s, bases, n);
  if (mb <= 10) return msm_10(s, bases, n, mb);
  return msm_small_rest(s, bases, n, mb);
}

// msm.rs:637-734: tables[j][d-1] = d * 2^(8j) * P, d in 1..255, j in 0..31
struct FixedBaseMul {
  std::vector<std::vector<Affine>> tables;
  size_t window_bits = 8;
  static FixedBaseMul precompute(const Jac& p, size_t window_bits = 8) {
    FixedBaseMul t;
    t.window_bits = window_bits;
    size_t num_windows = (256 + window_bits - 1) / window_bits;
    size_t per = ((size_t)1 << window_bits) - 1;
    std::vector<Jac> all;
    Jac base = p;
    for (size_t w = 0; w < num_windows; ++w) {
      Jac acc = base;
      all.push_back(acc);
      for (size_t d = 1; d < per; ++d) {
        acc = acc.add(base);
        all.push_back(acc);
      }
      for (size_t k = 0; k < window_bits; ++k) base = base.dbl();
    }
    std::vector<Affine> aff = batch_affine(all);
    for (size_t w = 0; w < num_windows; ++w) t.tables.emplace_back(aff.begin() + w * per, aff.begin() + (w + 1) * per);
    return t;
  }
  Jac mul(const Fq& scalar) const {  // msm.rs:691-725 (w == 8: one byte per window)
    uint8_t bytes[32];
    scalar.to_repr(bytes);
    Jac acc = Jac::identity();
    for (size_t j = 0; j < tables.size(); ++j) {
      size_t digit = get_at(j, window_bits, bytes);
      if (digit) acc = acc.add_mixed(tables[j][digit - 1]);
    }
    return acc;
  }
};

}  // namespace oracle
