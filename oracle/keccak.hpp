// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Keccak-f[1600], Keccak-256 (pad 0x01, as sha3::Keccak256 in the reference, Cargo.toml sha3 0.10),
// SHAKE256 (pad 0x1f) and the reference's Fiat-Shamir transcript (src/provider/keccak.rs:18-105).
// Pinned by the reference KATs src/provider/keccak.rs:146-163 (see tests/test_oracle_kats.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "field.hpp"

namespace oracle {

inline void keccak_f1600(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int round = 0; round < 24; ++round) {
    uint64_t bc[5];
    for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; ++i) {
      uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; ++i) {
      int j = PIL[i];
      uint64_t b = st[j];
      st[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
      for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= RC[round];
  }
}

// Incremental sponge with rate 136 bytes (Keccak-256 and SHAKE256 share it).
struct Sponge136 {
  uint64_t st[25];
  uint8_t buf[136];
  size_t pos;
  Sponge136() { reset(); }
  void reset() {
    memset(st, 0, sizeof st);
    pos = 0;
  }
  void absorb_block() {
    for (int i = 0; i < 17; ++i) {
      uint64_t w;
      memcpy(&w, buf + 8 * i, 8);
      st[i] ^= w;
    }
    keccak_f1600(st);
    pos = 0;
  }
  void update(const uint8_t* data, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      buf[pos++] = data[i];
      if (pos == 136) absorb_block();
    }
  }
  void pad(uint8_t domain) {
    memset(buf + pos, 0, 136 - pos);
    buf[pos] ^= domain;
    buf[135] ^= 0x80;
    absorb_block();
  }
};

struct Keccak256 : Sponge136 {
  void finalize(uint8_t out[32]) {
    pad(0x01);
    memcpy(out, st, 32);
  }
};

struct Shake256 : Sponge136 {
  bool squeezing = false;
  size_t out_pos = 0;
  void read(uint8_t* out, size_t n) {
    if (!squeezing) {
      pad(0x1f);
      squeezing = true;
      out_pos = 0;
    }
    for (size_t i = 0; i < n; ++i) {
      if (out_pos == 136) {
        keccak_f1600(st);
        out_pos = 0;
      }
      out[i] = ((const uint8_t*)st)[out_pos++];
    }
  }
};

// src/provider/keccak.rs:18-105
struct Transcript {
  uint16_t round = 0;
  uint8_t state[64];
  Keccak256 hasher;

  static void compute_updated_state(Keccak256 h, const uint8_t* input, size_t n, uint8_t out[64]) {
    h.update(input, n);
    Keccak256 lo = h, hi = h;
    uint8_t b0 = 0, b1 = 1;
    lo.update(&b0, 1);
    hi.update(&b1, 1);
    lo.finalize(out);
    hi.finalize(out + 32);
  }
  explicit Transcript(const char* label) {  // keccak.rs:57-68
    std::vector<uint8_t> in;
    const char* tag = "NoTR";
    in.insert(in.end(), tag, tag + 4);
    in.insert(in.end(), label, label + strlen(label));
    compute_updated_state(Keccak256(), in.data(), in.size(), state);
  }
  void absorb_bytes(const char* label, const uint8_t* bytes, size_t n) {  // keccak.rs:96-99
    hasher.update((const uint8_t*)label, strlen(label));
    hasher.update(bytes, n);
  }
  void dom_sep(const char* bytes) {  // keccak.rs:101-104
    hasher.update((const uint8_t*)"NoDS", 4);
    hasher.update((const uint8_t*)bytes, strlen(bytes));
  }
  // keccak.rs:70-94; returns the 64 uniform bytes (callers reduce them into their field)
  void squeeze_bytes(const char* label, uint8_t out[64]) {
    std::vector<uint8_t> in;
    in.insert(in.end(), (const uint8_t*)"NoDS", (const uint8_t*)"NoDS" + 4);
    in.push_back((uint8_t)(round & 0xff));
    in.push_back((uint8_t)(round >> 8));
    in.insert(in.end(), state, state + 64);
    in.insert(in.end(), label, label + strlen(label));
    compute_updated_state(hasher, in.data(), in.size(), out);
    round = (uint16_t)(round + 1);
    memcpy(state, out, 64);
    hasher.reset();
  }
  template <class F>
  F squeeze(const char* label) {
    uint8_t out[64];
    squeeze_bytes(label, out);
    return F::from_uniform(out);
  }
  template <class F>
  void absorb_scalar(const char* label, const F& s) {
    uint8_t be[32];
    s.to_be_bytes(be);
    absorb_bytes(label, be, 32);
  }
  template <class F>
  void absorb_scalars(const char* label, const F* s, size_t n) {  // src/traits/transcript.rs:35-42
    std::vector<uint8_t> b(32 * n);
    for (size_t i = 0; i < n; ++i) s[i].to_be_bytes(b.data() + 32 * i);
    absorb_bytes(label, b.data(), b.size());
  }
};

}  // namespace oracle
