// Helpers shared by the host-side drivers above the C ABI (spartan_snark.cpp, neutronnova_nifs.cpp): error plumbing, the transcript wrapper over
// sp_transcript_*, the transcript encodings of points / commitments, the randomness tape.
#pragma once
#include <type_traits>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spartan_hip.h"
#include "../csrc/curve.hpp"
#include "../csrc/keccak.hpp"

namespace spartan2 {

typedef FqP S;
static const size_t DEFAULT_COMMITMENT_WIDTH = 2048;  // src/lib.rs:63

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
static void ck(int rc, const char* what) {
  if (rc != SP_OK) throw Error(rc, std::string(what) + ": " + sp_last_error());
}

static inline const uint64_t* u64p(const fe_t* p) { return reinterpret_cast<const uint64_t*>(p); }
static inline uint64_t* u64p(fe_t* p) { return reinterpret_cast<uint64_t*>(p); }

// ---- transcript helpers over the C ABI ---------------------------------------------------------------------------------
struct Tr {
  sp_transcript* t = nullptr;
  explicit Tr(sp_ctx* ctx, const char* label) { ck(sp_transcript_new(ctx, (const uint8_t*)label, strlen(label), &t), "transcript_new"); }
  explicit Tr(const sp_transcript* prefix) { ck(sp_transcript_clone(prefix, &t), "transcript_clone"); }
  struct Adopt {};
  Tr(sp_transcript* owned, Adopt) : t(owned) {}
  Tr(const Tr&) = delete;
  Tr& operator=(const Tr&) = delete;
  ~Tr() { sp_transcript_free(t); }
  void absorb(const char* label, const uint8_t* b, size_t n) { ck(sp_transcript_absorb(t, (const uint8_t*)label, strlen(label), b, n), "absorb"); }
  void absorb_scalars(const char* label, const fe_t* s, size_t n) {  // BE encoding (src/provider/traits.rs:282-286), slices concatenated
    std::vector<uint8_t> b(32 * n);
    for (size_t i = 0; i < n; ++i) sp::fe_to_be_bytes<S>(s[i], b.data() + 32 * i);
    absorb(label, b.data(), b.size());
  }
  fe_t squeeze(const char* label) {
    fe_t f;
    ck(sp_transcript_squeeze(t, (const uint8_t*)label, strlen(label), u64p(&f)), "squeeze");
    return f;
  }
  void dom_sep(const char* s) { ck(sp_transcript_dom_sep(t, (const uint8_t*)s, strlen(s)), "dom_sep"); }
};
// point -> x BE || y BE (src/provider/traits.rs:288-305)
static void point_bytes(const aff_t& a, uint8_t out[64]) {
  sp::fe_to_be_bytes<B>(a.x, out);
  sp::fe_to_be_bytes<B>(a.y, out + 32);
}
// HyraxCommitment::to_transcript_bytes (src/provider/pcs/hyrax_pc.rs:714-729)
static std::vector<uint8_t> commitment_bytes(const aff_t* rows, size_t n) {
  static const char* b = "poly_commitment_begin";
  static const char* e = "poly_commitment_end";
  std::vector<uint8_t> v(b, b + strlen(b));
  v.resize(v.size() + 64 * n);
  for (size_t i = 0; i < n; ++i) point_bytes(rows[i], v.data() + strlen(b) + 64 * i);
  v.insert(v.end(), e, e + strlen(e));
  return v;
}

// fn(i) for i in [0, n) on up to 8 short-lived threads when the job is big enough to pay for them (the transcript encodings of hundreds of
// instance commitments at BASELINE config 5: two field conversions per point, 131 K points).
template <class F>
static void parallel_for(size_t n, size_t work_per_item, F fn) {
  unsigned t = std::thread::hardware_concurrency();
  if (t > 8) t = 8;
  if (t < 2 || n < 2 || n * work_per_item < 16384) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  if (t > n) t = (unsigned)n;
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err(t);
  for (unsigned k = 0; k < t; ++k)
    th.emplace_back([&, k] {
      try {
        for (size_t i = k; i < n; i += t) fn(i);
      } catch (...) {
        err[k] = std::current_exception();
      }
    });
  for (auto& x : th) x.join();
  for (auto& e : err)
    if (e) std::rethrow_exception(e);
}

// One helper thread per prover for host work whose inputs are known long before its result is needed (hashing the 64 KiB encoding of comm_W:
// 0.3 ms that used to sit on the critical path of the PCS phase). The thread sleeps between jobs; the prover picks the result up with a short spin,
// by which time the job has normally been finished for hundreds of microseconds.
class Background {
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool posted_ = false, stop_ = false;
  std::atomic<int> pending_{0};
  std::exception_ptr err_;
  std::chrono::steady_clock::time_point posted_at_;
  // A sleeping thread's wake-up is the scheduler's to time - usually 5-50 us, now and then milliseconds (one prove in a few thousand was 4 ms long
  // for it): a waiter that finds the job still UNCLAIMED 30 us after it was posted takes it back and runs it itself.
  bool steal_and_run() {
    std::function<void()> f;
    {
      std::lock_guard<std::mutex> l(m_);
      if (!posted_ || std::chrono::steady_clock::now() - posted_at_ <= std::chrono::microseconds(30)) return false;
      f.swap(job_);
      posted_ = false;  // the helper, when it does wake, finds nothing posted and sleeps on
      posted_hint_.store(0, std::memory_order_relaxed);
    }
    try {
      f();
    } catch (...) {
      err_ = std::current_exception();
    }
    pending_.store(0, std::memory_order_release);
    return true;
  }
  // (HELPER_LOOKOUT_US: longer than a config-2 prove, so that in a stream of proves the helper whose one job sits at the START of a prove - the transcript
  // prefix - is still awake when the next prove posts it: a sleeper costs the poster a futex wake, ~10 us, and itself the wake-up)
  static constexpr int HELPER_LOOKOUT_US = 1500;
  // a helper that has just finished a job looks for the next one for a short while before it goes to sleep (while the driver allows helpers to spin:
  // few proves in flight): in a stream of proves the next job is tens of microseconds away, a sleeper's wake-up 10 - 50 us on top of that
  bool (*may_spin_)() = nullptr;
  std::atomic<int> posted_hint_{0};
  void loop() {
    for (;;) {
      std::function<void()> f;
      if (may_spin_) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; !posted_hint_.load(std::memory_order_acquire); ++spins) {
          if ((spins & 255u) == 255u && (!may_spin_() || std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(HELPER_LOOKOUT_US))) break;
          sp_relax();
        }
      }
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return posted_ || stop_; });
        posted_hint_.store(0, std::memory_order_relaxed);
        if (stop_) return;
        f.swap(job_);
        posted_ = false;
      }
      try {
        f();
      } catch (...) {
        err_ = std::current_exception();
      }
      pending_.store(0, std::memory_order_release);
    }
  }

 public:
  Background() = default;
  Background(const Background&) = delete;
  Background& operator=(const Background&) = delete;
  ~Background() {
    if (!th_.joinable()) return;
    wait_nothrow();
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_one();
    th_.join();
  }
  void submit(std::function<void()> f) {  // one job at a time
    wait();
    pending_.store(1, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> l(m_);
      job_ = std::move(f);
      posted_ = true;
      posted_at_ = std::chrono::steady_clock::now();
    }
    posted_hint_.store(1, std::memory_order_release);
    if (!th_.joinable()) th_ = std::thread([this] { loop(); });
    cv_.notify_one();
  }
  void set_spin_policy(bool (*may_spin)()) { may_spin_ = may_spin; }
  // for a waiter that polls something the job produces rather than the job's end: takes the job back if the helper has not begun it (see above)
  bool try_steal() { return pending_.load(std::memory_order_acquire) && steal_and_run(); }
  void wait_nothrow() {  // for exit paths: also drops what the job threw, so that it cannot resurface in a later submit()
    for (unsigned spins = 0; pending_.load(std::memory_order_acquire); ++spins) {
      if ((spins & 63u) == 63u && steal_and_run()) break;
      sp_relax();
    }
    err_ = nullptr;
  }
  void wait() {  // rethrows what the job threw
    for (unsigned spins = 0; pending_.load(std::memory_order_acquire); ++spins) {
      if ((spins & 63u) == 63u && steal_and_run()) break;
      sp_relax();
    }
    if (err_) {
      std::exception_ptr e = err_;
      err_ = nullptr;
      std::rethrow_exception(e);
    }
  }
};

// The reference's `par_iter` over a host vector: body(lo, hi) on contiguous ranges, spread over the library's polling threads (sp_host_parallel_for;
// ranges nobody claims are run by the caller). Field arithmetic is exact, so the partition never shows in a result.
template <class F>
static inline void par_for(size_t n, size_t min_chunk, F&& body) {
  size_t parts = min_chunk ? n / min_chunk : 1;
  const size_t w = (size_t)sp_walkers() + 1;
  if (parts > w) parts = w;
  if (parts > 32) parts = 32;
  if (parts <= 1) {
    if (n) body((size_t)0, n);
    return;
  }
  struct Ctx {
    typename std::remove_reference<F>::type* f;
    size_t n;
  } c{&body, n};
  ck(sp_host_parallel_for((unsigned)parts, [](void* a, unsigned p, unsigned np) {
       Ctx& x = *static_cast<Ctx*>(a);
       (*x.f)(x.n * p / np, x.n * (p + 1) / np);
     }, &c),
     "parallel_for");
}

struct Tape {
  const uint8_t* bytes;
  size_t blocks, pos = 0;
  fe_t next() {
    if (pos >= blocks) throw Error(SP_ERR_INTERNAL, "random tape exhausted");
    return fe_from_uniform<S>(bytes + 64 * pos++);
  }
  void skip(size_t n) {
    if (pos + n > blocks) throw Error(SP_ERR_INTERNAL, "random tape exhausted");
    pos += n;
  }
};

}  // namespace spartan2
