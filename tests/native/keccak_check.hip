// Host-only check: the unrolled host permutation (keccak.hpp keccak_host::permute, generic and BMI builds behind a CPU check) against the loop form the
// device compiles, and Keccak-256("") / Keccak-256("abc") against their published digests. Built and run by tests/test_field_host.py.
#include <cstdio>
#include <cstring>
#include <ctime>

#include "../../spartan2_amd/csrc/keccak.hpp"

int main() {
#if !defined(__HIP_DEVICE_COMPILE__)
  uint64_t a[25], b[25], c[25];
  for (int i = 0; i < 25; ++i) a[i] = b[i] = c[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull + 7;
  int bad = 0;
  for (int it = 0; it < 5000; ++it) {
    sp::keccak_permute_loop(a);
    sp::keccak_permute(b);
    sp::keccak_host::permute_generic(c);
    for (int i = 0; i < 25; ++i) bad += a[i] != b[i] || a[i] != c[i];
  }
  static const unsigned char empty[32] = {0xc5, 0xd2, 0x46, 0x01, 0x86, 0xf7, 0x23, 0x3c, 0x92, 0x7e, 0x7d, 0xb2, 0xdc, 0xc7, 0x03, 0xc0,
                                          0xe5, 0x00, 0xb6, 0x53, 0xca, 0x82, 0x27, 0x3b, 0x7b, 0xfa, 0xd8, 0x04, 0x5d, 0x85, 0xa4, 0x70};
  static const unsigned char abc[32] = {0x4e, 0x03, 0x65, 0x7a, 0xea, 0x45, 0xa9, 0x4f, 0xc7, 0xd4, 0x7b, 0xa8, 0x26, 0xc8, 0xd6, 0x67,
                                        0xc0, 0xd1, 0xe6, 0xe3, 0x3a, 0x64, 0xa0, 0x36, 0xec, 0x44, 0xf5, 0x8f, 0xa1, 0x2d, 0x6c, 0x45};
  uint8_t out[32];
  sp::Keccak256State k;
  k.init();
  k.finish(out);
  bad += memcmp(out, empty, 32) != 0;
  k.init();
  k.update(reinterpret_cast<const uint8_t*>("abc"), 3);
  k.finish(out);
  bad += memcmp(out, abc, 32) != 0;
  // absorbs of any chunking give one digest (whole blocks go straight from the input, the rest through the buffer)
  {
    uint8_t msg[1000], ref[32];
    for (int i = 0; i < 1000; ++i) msg[i] = (uint8_t)(i * 131 + 7);
    k.init();
    for (int i = 0; i < 1000; ++i) k.update(msg + i, 1);
    k.finish(ref);
    const int chunks[] = {1000, 7, 135, 136, 137, 272, 300};
    for (int c : chunks) {
      k.init();
      for (int off = 0; off < 1000; off += c) k.update(msg + off, off + c <= 1000 ? c : 1000 - off);
      k.finish(out);
      bad += memcmp(out, ref, 32) != 0;
    }
  }
  {  // two states side by side (the two final permutations of a transcript squeeze) against one at a time
    uint64_t p[25], q[25], p2[25], q2[25];
    for (int i = 0; i < 25; ++i) {
      p[i] = p2[i] = (uint64_t)(i + 3) * 0xD1B54A32D192ED03ull;
      q[i] = q2[i] = ~(uint64_t)i * 0x9E3779B97F4A7C15ull;
    }
    for (int it = 0; it < 2000; ++it) {
      sp::keccak_permute_loop(p);
      sp::keccak_permute_loop(q);
      sp::keccak_host::permute2(p2, q2);
      for (int i = 0; i < 25; ++i) bad += p[i] != p2[i] || q[i] != q2[i];
    }
    // and through the transcript: squeeze == the two hashes done one after the other
    sp::Transcript t;
    t.init(reinterpret_cast<const uint8_t*>("chk"), 3);
    uint8_t msg[300];
    for (int i = 0; i < 300; ++i) msg[i] = (uint8_t)(7 * i + 1);
    for (int n : {0, 1, 59, 60, 61, 135, 136, 137, 300}) {
      t.absorb(reinterpret_cast<const uint8_t*>("m"), 1, msg, (size_t)n);
      sp::Keccak256State base = t.h;
      uint8_t in[80], want[64], got[64];
      const uint8_t tag[4] = {'N', 'o', 'D', 'S'};
      memcpy(in, tag, 4);
      in[4] = (uint8_t)t.round;
      in[5] = (uint8_t)(t.round >> 8);
      memcpy(in + 6, t.state, 64);
      in[70] = 'c';
      base.update(in, 71);
      sp::Keccak256State lo = base, hi = base;
      const uint8_t z = 0, o = 1;
      lo.update(&z, 1);
      hi.update(&o, 1);
      lo.finish(want);
      hi.finish(want + 32);
      bad += !t.squeeze_bytes(reinterpret_cast<const uint8_t*>("c"), 1, got);
      bad += memcmp(got, want, 64) != 0;
    }
  }
  {  // ns per PAIR of permutations: one after the other, and side by side
    uint64_t x[25], y[25];
    for (int i = 0; i < 25; ++i) x[i] = y[i] = i;
    timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 0; it < 100000; ++it) {
      sp::keccak_host::permute(x);
      sp::keccak_host::permute(y);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int it = 0; it < 100000; ++it) sp::keccak_host::permute2(x, y);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    auto ns = [](const timespec& a, const timespec& b) { return ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) / 100000.0; };
    printf("keccak: two permutations one after the other %.0f ns, side by side %.0f ns (%llx)\n", ns(t0, t1), ns(t1, t2), (unsigned long long)(x[0] ^ y[0]));
  }
  printf("keccak: %d mismatches\n", bad);
  return bad != 0;
#else
  return 0;
#endif
}
