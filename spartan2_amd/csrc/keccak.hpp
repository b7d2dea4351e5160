// Keccak-f[1600] / Keccak-256 and the Fiat-Shamir transcript of the reference
// (src/provider/keccak.rs:18-105) for the product path. __host__ __device__ so the same sponge can run in a
// single-wave device kernel (SURVEY.md kernel K15) as well as in the host-side round loop.
#pragma once
#include <stdint.h>
#include <string.h>

#include "field.hpp"

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#endif
namespace sp {

SP_HD uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// Plane-per-plane formulation (theta, then rho+pi into a second state, then chi+iota): the form the device compiles, and the host's reference for the
// unrolled one below.
SP_HD void keccak_permute_loop(uint64_t a[25]) {
  constexpr uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                               0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                               0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                               0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                               0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  // rotation offsets r[x][y] indexed as [x + 5*y]
  constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) {
        uint64_t v = a[x + 5 * y] ^ d[x];
        int r = RHO[x + 5 * y];
        // pi: B[y][2x+3y] = rot(A[x][y])
        b[y + 5 * ((2 * x + 3 * y) % 5)] = r ? rotl64(v, r) : v;
      }
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[rnd];
  }
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host form: the transcript is one dependent chain of permutations (241 of them in front of a config-2 prove's first challenge, 77 K in a config-5
// NIFS), so the host's time per permutation is on the critical path. Fully unrolled, one output plane at a time - the five rotated lanes of a plane
// live in registers only, two rounds per iteration ping-pong between two states - and compiled a second time for BMI (andn, rorx) where the CPU has
// it: 395 -> 368 -> 303 ns on the build container's 2.1 GHz Xeon. (AVX-512 auto-vectorisation of either form is slower: 515 ns.)
namespace keccak_host {
constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
constexpr uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                             0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                             0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                             0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                             0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#define SP_KH_INLINE static inline __attribute__((always_inline))
// The round function over a word type W: uint64_t (one state), or two states side by side in an xmm register (W2 below: the two final permutations of a
// squeeze - Keccak(.. || 0) and Keccak(.. || 1), keccak.rs:33-54 - are independent and run for the price of one).
SP_KH_INLINE uint64_t w_xor(uint64_t a, uint64_t b) { return a ^ b; }
SP_KH_INLINE uint64_t w_chi(uint64_t b0, uint64_t b1, uint64_t b2) { return b0 ^ (~b1 & b2); }
template <int R>
SP_KH_INLINE uint64_t w_rol(uint64_t x) {
  if constexpr (R == 0) return x;
  else return (x << R) | (x >> (64 - R));
}
SP_KH_INLINE uint64_t w_rc(uint64_t, uint64_t rc) { return rc; }
// two independent states at once: word i of both in one 128-bit vector (a generic vector type: inside the AVX-512VL function below the shifts become
// vprolq and the xor / and-not chains vpternlogq)
typedef unsigned long long W2 __attribute__((vector_size(16)));
SP_KH_INLINE W2 w_xor(W2 a, W2 b) { return a ^ b; }
SP_KH_INLINE W2 w_chi(W2 b0, W2 b1, W2 b2) { return b0 ^ (~b1 & b2); }
template <int R>
SP_KH_INLINE W2 w_rol(W2 x) {
  if constexpr (R == 0) return x;
  else return (x << R) | (x >> (64 - R));
}
SP_KH_INLINE W2 w_rc(W2, uint64_t rc) { return W2{rc, rc}; }
// lane (X, Y) after rho and pi is lane (x, y) of the input with y = X and 2x + 3y = Y (mod 5), i.e. x = X + 3Y
template <int X, int Y, class W>
SP_KH_INLINE W moved(const W* a, const W* d) {
  constexpr int x = (X + 3 * Y) % 5, y = X;
  return w_rol<RHO[x + 5 * y]>(w_xor(a[x + 5 * y], d[x]));
}
template <int Y, class W>
SP_KH_INLINE void plane(const W* a, const W* d, W* e) {
  const W b0 = moved<0, Y>(a, d), b1 = moved<1, Y>(a, d), b2 = moved<2, Y>(a, d), b3 = moved<3, Y>(a, d), b4 = moved<4, Y>(a, d);
  e[0 + 5 * Y] = w_chi(b0, b1, b2);
  e[1 + 5 * Y] = w_chi(b1, b2, b3);
  e[2 + 5 * Y] = w_chi(b2, b3, b4);
  e[3 + 5 * Y] = w_chi(b3, b4, b0);
  e[4 + 5 * Y] = w_chi(b4, b0, b1);
}
template <class W>
SP_KH_INLINE void round(const W* a, W* e, uint64_t rc) {
  W c[5], d[5];
  for (int x = 0; x < 5; ++x) c[x] = w_xor(w_xor(w_xor(a[x], a[x + 5]), w_xor(a[x + 10], a[x + 15])), a[x + 20]);
  d[0] = w_xor(c[4], w_rol<1>(c[1]));
  d[1] = w_xor(c[0], w_rol<1>(c[2]));
  d[2] = w_xor(c[1], w_rol<1>(c[3]));
  d[3] = w_xor(c[2], w_rol<1>(c[4]));
  d[4] = w_xor(c[3], w_rol<1>(c[0]));
  plane<0>(a, d, e);
  plane<1>(a, d, e);
  plane<2>(a, d, e);
  plane<3>(a, d, e);
  plane<4>(a, d, e);
  e[0] = w_xor(e[0], w_rc(e[0], rc));
}
template <class W>
SP_KH_INLINE void body(W a[25]) {
  W e[25];
  for (int r = 0; r < 24; r += 2) {
    round(a, e, RC[r]);
    round(e, a, RC[r + 1]);
  }
}
static void permute_generic(uint64_t a[25]) { body(a); }
__attribute__((target("bmi,bmi2"))) static void permute_bmi(uint64_t a[25]) { body(a); }
inline void permute(uint64_t a[25]) {
  static void (*const f)(uint64_t*) = (__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2")) ? permute_bmi : permute_generic;
  f(a);
}
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512vl"))) static void permute2_avx512vl(uint64_t a[25], uint64_t b[25]) {
  W2 s[25];
  for (int i = 0; i < 25; ++i) s[i] = W2{a[i], b[i]};
  body(s);
  for (int i = 0; i < 25; ++i) {
    a[i] = s[i][0];
    b[i] = s[i][1];
  }
}
#endif
// the permutation applied to two independent states
inline void permute2(uint64_t a[25], uint64_t b[25]) {
#if defined(__x86_64__)
  static const bool two_way = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
  if (two_way) {
    permute2_avx512vl(a, b);
    return;
  }
#endif
  permute(a);
  permute(b);
}
#undef SP_KH_INLINE
}  // namespace keccak_host
#endif
SP_HD void keccak_permute(uint64_t a[25]) {
#if defined(__HIP_DEVICE_COMPILE__)
  keccak_permute_loop(a);
#else
  keccak_host::permute(a);
#endif
}

// Incremental Keccak-256 (rate 136, pad 0x01 .. 0x80 — the pre-NIST padding sha3::Keccak256 uses).
struct Keccak256State {
  uint64_t a[25];
  uint8_t buf[136];
  uint32_t fill;
  SP_HD void init() {
    for (int i = 0; i < 25; ++i) a[i] = 0;
    fill = 0;
  }
  SP_HD void block() {
    for (int i = 0; i < 17; ++i) {
      uint64_t w = 0;
#if !defined(__HIP_DEVICE_COMPILE__)
      memcpy(&w, buf + 8 * i, 8);  // (x86-64: little-endian)
#else
      for (int k = 0; k < 8; ++k) w |= (uint64_t)buf[8 * i + k] << (8 * k);
#endif
      a[i] ^= w;
    }
    keccak_permute(a);
    fill = 0;
  }
  SP_HD void update(const uint8_t* p, size_t n) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // host: whole blocks straight from the input (seventeen 64-bit little-endian loads), the rest through the buffer in one copy
    if (fill) {
      const size_t take = n < 136 - fill ? n : 136 - fill;
      memcpy(buf + fill, p, take);
      fill += (uint32_t)take;
      p += take;
      n -= take;
      if (fill == 136) block();
    }
    for (; n >= 136; p += 136, n -= 136) {
      for (int i = 0; i < 17; ++i) {
        uint64_t w;
        memcpy(&w, p + 8 * i, 8);
        a[i] ^= w;
      }
      keccak_permute(a);
    }
    if (n) {
      memcpy(buf, p, n);
      fill = (uint32_t)n;
    }
#else
    for (size_t i = 0; i < n; ++i) {
      buf[fill++] = p[i];
      if (fill == 136) block();
    }
#endif
  }
#if !defined(__HIP_DEVICE_COMPILE__)
  // finish() of two states whose last blocks are permuted side by side (host)
  static void finish2(Keccak256State& x, Keccak256State& y, uint8_t out_x[32], uint8_t out_y[32]) {
    Keccak256State* st[2] = {&x, &y};
    for (Keccak256State* k : st) {
      for (uint32_t i = k->fill; i < 136; ++i) k->buf[i] = 0;
      k->buf[k->fill] ^= 0x01;
      k->buf[135] ^= 0x80;
      for (int i = 0; i < 17; ++i) {
        uint64_t w;
        memcpy(&w, k->buf + 8 * i, 8);
        k->a[i] ^= w;
      }
      k->fill = 0;
    }
    keccak_host::permute2(x.a, y.a);
    for (int i = 0; i < 32; ++i) {
      out_x[i] = (uint8_t)(x.a[i >> 3] >> (8 * (i & 7)));
      out_y[i] = (uint8_t)(y.a[i >> 3] >> (8 * (i & 7)));
    }
  }
#endif
  SP_HD void finish(uint8_t out[32]) {
    for (uint32_t i = fill; i < 136; ++i) buf[i] = 0;
    buf[fill] ^= 0x01;
    buf[135] ^= 0x80;
    block();
    for (int i = 0; i < 32; ++i) out[i] = (uint8_t)(a[i >> 3] >> (8 * (i & 7)));
  }
};

// SHAKE256 (rate 136, pad 0x1f): the XOF stream generators are derived from (src/provider/traits.rs:205-214)
struct Shake256State {
  Keccak256State k;
  bool squeezing;
  uint32_t pos;
  SP_HD void init() {
    k.init();
    squeezing = false;
    pos = 0;
  }
  SP_HD void update(const uint8_t* p, size_t n) { k.update(p, n); }
  SP_HD void read(uint8_t* out, size_t n) {
    if (!squeezing) {
      for (uint32_t i = k.fill; i < 136; ++i) k.buf[i] = 0;
      k.buf[k.fill] ^= 0x1f;
      k.buf[135] ^= 0x80;
      k.block();
      squeezing = true;
      pos = 0;
    }
    for (size_t i = 0; i < n; ++i) {
      if (pos == 136) {
        keccak_permute(k.a);
        pos = 0;
      }
      out[i] = (uint8_t)(k.a[pos >> 3] >> (8 * (pos & 7)));
      ++pos;
    }
  }
};

// Keccak256Transcript<E> (src/provider/keccak.rs:26-105)
struct Transcript {
  uint16_t round;
  uint8_t state[64];
  Keccak256State h;

  // compute_updated_state (keccak.rs:33-54): Keccak(running || input || 0) || Keccak(running || input || 1)
  SP_HD static void updated_state(Keccak256State base, const uint8_t* in, size_t n, uint8_t out[64]) {
    base.update(in, n);
    Keccak256State lo = base, hi = base;
    const uint8_t z = 0, o = 1;
    lo.update(&z, 1);
    hi.update(&o, 1);
#if !defined(__HIP_DEVICE_COMPILE__)
    Keccak256State::finish2(lo, hi, out, out + 32);
#else
    lo.finish(out);
    hi.finish(out + 32);
#endif
  }
  SP_HD void init(const uint8_t* label, size_t n) {  // new (keccak.rs:57-68): state = f("NoTR" || label)
    Keccak256State k;
    k.init();
    const uint8_t tag[4] = {'N', 'o', 'T', 'R'};
    k.update(tag, 4);
    updated_state(k, label, n, state);
    round = 0;
    h.init();
  }
  SP_HD void absorb(const uint8_t* label, size_t ln, const uint8_t* bytes, size_t n) {  // keccak.rs:96-99
    h.update(label, ln);
    h.update(bytes, n);
  }
  SP_HD void dom_sep(const uint8_t* bytes, size_t n) {  // keccak.rs:101-104
    const uint8_t tag[4] = {'N', 'o', 'D', 'S'};
    h.update(tag, 4);
    h.update(bytes, n);
  }
  // squeeze (keccak.rs:70-94). Returns false on round-counter overflow (InternalTranscriptError).
  SP_HD bool squeeze_bytes(const uint8_t* label, size_t ln, uint8_t out[64]) {
    Keccak256State k = h;
    const uint8_t hdr[6] = {'N', 'o', 'D', 'S', (uint8_t)(round & 0xff), (uint8_t)(round >> 8)};
    k.update(hdr, 6);
    k.update(state, 64);
    updated_state(k, label, ln, out);
    if (round == 0xffff) return false;
    round = (uint16_t)(round + 1);
    for (int i = 0; i < 64; ++i) state[i] = out[i];
    h.init();
    return true;
  }
  template <class FP>
  SP_HD bool squeeze(const uint8_t* label, size_t ln, fe_t* out) {
    uint8_t b[64];
    if (!squeeze_bytes(label, ln, b)) return false;
    *out = fe_from_uniform<FP>(b);  // PrimeFieldExt::from_uniform (src/provider/traits.rs:275-280)
    return true;
  }
};

// scalar -> transcript bytes: to_repr reversed, i.e. big-endian canonical (src/provider/traits.rs:282-286)
template <class FP>
SP_HD void fe_to_be_bytes(const fe_t& a, uint8_t out[32]) {
  fe_t c = fe_to_canonical<FP>(a);
  for (int i = 0; i < 8; ++i) {
    uint32_t w = c.v[7 - i];
    out[4 * i] = (uint8_t)(w >> 24);
    out[4 * i + 1] = (uint8_t)(w >> 16);
    out[4 * i + 2] = (uint8_t)(w >> 8);
    out[4 * i + 3] = (uint8_t)w;
  }
}
// to_repr(): little-endian canonical (UniPoly coefficients enter the transcript this way, src/polys/univariate.rs:182-190)
template <class FP>
SP_HD void fe_to_le_bytes(const fe_t& a, uint8_t out[32]) {
  fe_t c = fe_to_canonical<FP>(a);
  for (int i = 0; i < 8; ++i) {
    uint32_t w = c.v[i];
    out[4 * i] = (uint8_t)w;
    out[4 * i + 1] = (uint8_t)(w >> 8);
    out[4 * i + 2] = (uint8_t)(w >> 16);
    out[4 * i + 3] = (uint8_t)(w >> 24);
  }
}

}  // namespace sp
