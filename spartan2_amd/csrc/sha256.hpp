// SHA-256 (FIPS 180-4) for the key digests of the wire formats (DigestComputer, src/digest.rs:49-77: `Sha256::new()` fed by a buffered writer).
// Host only. Two block functions: a portable one and one on the x86 SHA extensions, picked once at run time (the digest of a config-2 verifier
// key hashes ~230 MB of matrix bytes at setup: 0.7 s portable, ~0.12 s with the extensions).
#pragma once
#include <cpuid.h>
#include <immintrin.h>

#include <cstdint>
#include <cstring>

namespace sp {

class Sha256 {
  uint32_t h_[8];
  uint8_t buf_[64];
  uint64_t total_ = 0;
  size_t fill_ = 0;

  static const uint32_t* K() {
    static const uint32_t k[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    return k;
  }
  static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

  static void blocks_portable(uint32_t st[8], const uint8_t* p, size_t nblocks) {
    const uint32_t* k = K();
    for (; nblocks; --nblocks, p += 64) {
      uint32_t w[16], v[8];
      for (int i = 0; i < 16; ++i) {
        uint32_t x;
        memcpy(&x, p + 4 * i, 4);
        w[i] = __builtin_bswap32(x);
      }
      memcpy(v, st, 32);
      for (int i = 0; i < 64; ++i) {
        if (i >= 16) {  // message schedule in a 16-word ring
          const uint32_t a = w[(i + 1) & 15], b = w[(i + 14) & 15];
          w[i & 15] += (ror(a, 7) ^ ror(a, 18) ^ (a >> 3)) + w[(i + 9) & 15] + (ror(b, 17) ^ ror(b, 19) ^ (b >> 10));
        }
        const uint32_t e = v[4], a = v[0];
        const uint32_t t1 = v[7] + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & v[5]) ^ (~e & v[6])) + k[i] + w[i & 15];
        const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & v[1]) ^ (a & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6], v[6] = v[5], v[5] = v[4], v[4] = v[3] + t1, v[3] = v[2], v[2] = v[1], v[1] = v[0], v[0] = t1 + t2;
      }
      for (int i = 0; i < 8; ++i) st[i] += v[i];
    }
  }

  // x86 SHA extensions: state kept as (ABEF, CDGH); four rounds per sha256rnds2 pair, the schedule by sha256msg1 / sha256msg2
  __attribute__((target("sha,sse4.1,ssse3"))) static void blocks_shani(uint32_t st[8], const uint8_t* p, size_t nblocks) {
    const __m128i* kv = (const __m128i*)K();
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bll, 0x0405060700010203ll);
    __m128i t = _mm_loadu_si128((const __m128i*)st), s1 = _mm_loadu_si128((const __m128i*)(st + 4));
    t = _mm_shuffle_epi32(t, 0xB1);                 // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);               // EFGH
    __m128i s0 = _mm_alignr_epi8(t, s1, 8);         // ABEF
    s1 = _mm_blend_epi16(s1, t, 0xF0);              // CDGH
    for (; nblocks; --nblocks, p += 64) {
      const __m128i save0 = s0, save1 = s1;
      __m128i m[4];
      for (int i = 0; i < 4; ++i) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), bswap);
      for (int g = 0; g < 16; ++g) {  // group g = rounds 4g .. 4g+3 on message words m[g & 3]
        __m128i wk = _mm_add_epi32(m[g & 3], _mm_loadu_si128(kv + g));
        s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
        s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(wk, 0x0E));
        if (g < 12) {  // words of group g + 4 from groups g .. g + 3
          __m128i x = _mm_sha256msg1_epu32(m[g & 3], m[(g + 1) & 3]);
          x = _mm_add_epi32(x, _mm_alignr_epi8(m[(g + 3) & 3], m[(g + 2) & 3], 4));
          m[g & 3] = _mm_sha256msg2_epu32(x, m[(g + 3) & 3]);
        }
      }
      s0 = _mm_add_epi32(s0, save0);
      s1 = _mm_add_epi32(s1, save1);
    }
    t = _mm_shuffle_epi32(s0, 0x1B);                // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);               // DCHG
    s0 = _mm_blend_epi16(t, s1, 0xF0);              // DCBA
    s1 = _mm_alignr_epi8(s1, t, 8);                 // HGFE
    _mm_storeu_si128((__m128i*)st, s0);
    _mm_storeu_si128((__m128i*)(st + 4), s1);
  }

  static bool have_shani() {
    static const bool v = [] {
      if (getenv("SPARTAN_SHA_PORTABLE")) return false;
      unsigned a, b, c, d;
      if (!__get_cpuid_count(7, 0, &a, &b, &c, &d) || !(b & (1u << 29))) return false;  // CPUID.7.0:EBX.SHA
      if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
      return (c & (1u << 19)) && (c & (1u << 9));  // SSE4.1, SSSE3
    }();
    return v;
  }
  void blocks(const uint8_t* p, size_t n) {
    if (have_shani()) blocks_shani(h_, p, n);
    else blocks_portable(h_, p, n);
  }

 public:
  Sha256() {
    static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h_, iv, 32);
  }
  static bool accelerated() { return have_shani(); }
  void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    total_ += n;
    if (fill_) {
      const size_t k = n < 64 - fill_ ? n : 64 - fill_;
      memcpy(buf_ + fill_, p, k);
      fill_ += k, p += k, n -= k;
      if (fill_ < 64) return;
      blocks(buf_, 1);
      fill_ = 0;
    }
    if (n >= 64) {
      blocks(p, n / 64);
      p += n & ~(size_t)63;
      n &= 63;
    }
    if (n) memcpy(buf_, p, n), fill_ = n;
  }
  void finish(uint8_t out[32]) {
    const uint64_t bits = total_ * 8;
    uint8_t tail[128];
    memset(tail, 0, sizeof tail);
    memcpy(tail, buf_, fill_);
    tail[fill_] = 0x80;
    const size_t len = fill_ < 56 ? 64 : 128;
    for (int i = 0; i < 8; ++i) tail[len - 1 - i] = (uint8_t)(bits >> (8 * i));
    blocks(tail, len / 64);
    for (int i = 0; i < 8; ++i) {
      const uint32_t x = __builtin_bswap32(h_[i]);
      memcpy(out + 4 * i, &x, 4);
    }
  }
};

}  // namespace sp
