// Host-side plumbing shared by the translation units of libspartan_hip.so: context, device tables,
// error reporting, per-kernel-class HIP-event timing.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spartan_hip.h"
#include "field.hpp"
#include "keccak.hpp"

namespace sp {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
// one iteration of a host-side wait for the device (a result slot, a mailbox answer, a helper's flag): the calling thread's wait hook
// (sp_set_wait_hook: a driver that runs several proofs on one thread switches to another proof here) or a `pause`
int live_contexts();  // contexts of this process between sp_ctx_create and sp_ctx_destroy
void relax();
// hipStreamSynchronize / hipEventSynchronize for library code: on a thread with a wait hook they poll (hipStreamQuery / hipEventQuery) through relax()
// instead of blocking — a blocked thread could not serve the other proofs it carries, and one of those may own a kernel that sits in front of this
// stream's work in a shared hardware queue while it waits for its host's next challenge. Never call them (or relax()) with a library mutex held.
hipError_t stream_sync(hipStream_t s);
// the same for waits that are known to be short and sit in a transcript chain (the NIFS rounds): polls hipStreamQuery for up to 2 ms before it lets the
// runtime block the thread - a thread that sleeps on the completion interrupt is now and then woken 7-30 ms late (one prove in a few hundred at config 3,
// always here: `nifs lap finish 11.7 ms` under SPARTAN_HOST_LAPS)
hipError_t stream_sync_short(hipStream_t s);
hipError_t event_sync(hipEvent_t e);
// A host-side poll gave up and takes its slow path (a stream synchronise, the mirror, a sleep): with SPARTAN_SLOWPATH_LOG set, one line on stderr per
// event - these are the places a rare multi-millisecond prove comes from (a profiler serialising the streams, a result that only became visible at
// the end of its kernel, a helper's job queued behind a waiting kernel)
void slow_note(const char* site, long spins);
// capi_group.hip: called by sp_sumcheck_quad after the challenge of `round` has gone to the device, when the context holds an announced opening
void pcs_ahead_on_challenge(void* ctx, size_t round, const uint64_t r[4]);
void pcs_ahead_free(sp_ctx* c);
bool pcs_ahead_wants(const sp_ctx* c, size_t rounds);  // is an opening announced whose point a sum-check of this many rounds draws?

#define SP_HIP(expr)                                                                               \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return sp::fail(SP_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// One sleeping helper thread of a context for host work that a library call can run beside its own device work (sp_hyrax_prove: the 64 KiB
// transcript encoding of the commitment and its Keccak blocks while the MSMs run). One job at a time; wait() spins (jobs are tens of microseconds).
// A sleeping thread's wake-up is the scheduler's to time - usually 5-50 us, now and then milliseconds (one prove in a few thousand was 4 ms long for
// it): a waiter that finds the job still UNCLAIMED 30 us after it was posted takes it back and runs it itself; a job the helper has begun is waited for.
class Worker {
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool posted_ = false, stop_ = false;
  std::atomic<int> pending_{0};
  std::atomic<bool> has_job_{false};      // mirror of posted_ for the spinning phase (no lock taken while polling)
  std::atomic<long long> hot_until_{0};   // steady-clock nanoseconds: until then the thread polls for its next job instead of sleeping
  std::chrono::steady_clock::time_point posted_at_;
  static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void loop() {
    for (;;) {
      std::function<void()> f;
      while (!has_job_.load(std::memory_order_acquire) && now_ns() < hot_until_.load(std::memory_order_relaxed)) sp::relax();
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return posted_ || stop_ || now_ns() < hot_until_.load(std::memory_order_relaxed); });
        if (stop_) return;
        if (!posted_) continue;  // woken to poll (keep_hot)
        f.swap(job_);
        posted_ = false;
        has_job_.store(false, std::memory_order_relaxed);
      }
      f();
      pending_.store(0, std::memory_order_release);
    }
  }
  void start() {
    if (!th_.joinable()) th_ = std::thread([this] { loop(); });
  }

 public:
  Worker() = default;
  Worker(const Worker&) = delete;
  Worker& operator=(const Worker&) = delete;
  ~Worker() {
    if (!th_.joinable()) return;
    wait();
    hot_until_.store(0, std::memory_order_relaxed);  // a polling thread falls through to the wait below and sees stop_
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_one();
    th_.join();
  }
  // The caller expects to post short jobs within the next `us` microseconds: the thread is woken now and polls for them (a job is then claimed within a
  // fraction of a microsecond instead of a wake-up's 5-50 us). Bounded: the thread goes back to sleep when the time is up.
  void keep_hot(long us) {
    const long long until = now_ns() + 1000ll * us;
    if (until > hot_until_.load(std::memory_order_relaxed)) hot_until_.store(until, std::memory_order_relaxed);
    start();
    cv_.notify_one();
  }
  bool hot() const { return th_.joinable() && now_ns() < hot_until_.load(std::memory_order_relaxed); }
  void submit(std::function<void()> f) {  // the job must not throw
    wait();
    pending_.store(1, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> l(m_);
      job_ = std::move(f);
      posted_ = true;
      has_job_.store(true, std::memory_order_release);
      posted_at_ = std::chrono::steady_clock::now();
    }
    start();
    cv_.notify_one();
  }
  // take_back_us: how long a job may stay unclaimed before the waiter runs it itself
  void wait(long take_back_us = 30) {
    for (unsigned spins = 0; pending_.load(std::memory_order_acquire); ++spins) {
      if ((spins & 63u) == 63u || take_back_us < 5) {
        std::function<void()> f;
        {
          std::lock_guard<std::mutex> l(m_);
          if (posted_ && std::chrono::steady_clock::now() - posted_at_ > std::chrono::microseconds(take_back_us)) {
            f.swap(job_);
            posted_ = false;  // the helper, when it does wake, finds nothing posted and sleeps on
            has_job_.store(false, std::memory_order_relaxed);
          }
        }
        if (f) {
          f();
          pending_.store(0, std::memory_order_release);
          return;
        }
      }
      sp::relax();
    }
  }
};

struct KStat {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double ms = 0;
  uint64_t launches = 0, bytes = 0;
};

}  // namespace sp

struct sp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream_eq = nullptr;  // sp_eq_table_begin's pyramids (a stream of their own: the auxiliary stream may hold a 100 us MSM stage of the helper thread)
  hipEvent_t eq_ev = nullptr;
  hipEvent_t aside_ev = nullptr, aside_main_ev = nullptr;  // sp_table_assemble_aside: its end / what the main stream held when it was issued
  bool aside_pending = false;
  hipEvent_t tail_ev = nullptr;  // recorded behind the last resident quadratic tail (sp_table_assemble_aside with behind_queued = 0 waits for it)
  bool tail_ev_pending = false;
  hipEvent_t eq_read_ev = nullptr;  // recorded behind the last k_eq_outer_lastk that READS d_eq_ahead: the next pyramids wait for it before rewriting the buffer
  bool eq_read_pending = false;
  fe_t* d_cubic_eq = nullptr;     // the cubic sum-check's two eq pyramids (EqSumCheckInstance::new tables)
  size_t cubic_eq_elems = 0;
  hipEvent_t cubic_ev = nullptr;  // the cubic sum-check's eq pyramids (built on the eq stream beside the main stream's queue) -> its first evaluation
  fe_t* d_eq_ahead = nullptr;     // two pyramids of <= 2^11 entries
  size_t eq_ahead_ell = 0, eq_ahead_known = 0;  // set by sp_eq_table_begin, consumed by sp_eq_table_finish
  fe_t eq_ahead_r[32];
  hipStream_t stream2 = nullptr;  // auxiliary stream: work that does not depend on the transcript (ipa.rs:139-147 delta) overlaps the sum-checks
  fe_t* d_scratch = nullptr;  // block partials etc.
  size_t scratch_elems = 0;
  fe_t* h_pinned = nullptr;  // small result buffer, pinned + mapped
  fe_t* d_pinned = nullptr;  // device-side address of h_pinned
  // challenge mailbox (kernels_poly.hpp mail_wait): in fine-grained device memory written through the PCIe BAR when the system has a large BAR
  // (mail_dev), otherwise inside h_pinned. h_mail / d_mail = host-side / device-side address of line 0 of the ring (MAIL_RING lines of 64 bytes, line = seq & 7).
  volatile uint32_t* h_mail = nullptr;
  const unsigned* d_mail = nullptr;
  volatile uint32_t* h_mail_mirror = nullptr;  // mail_dev only: the host-memory copy of the ring (second path of mail_wait)
  const unsigned* d_mail_mirror = nullptr;
  void* mail_alloc = nullptr;
  bool mail_dev = false;
  bool hooks_host_only = false;  // sp_ctx_round_hooks_host_only: the batched sum-checks may queue a round's launch ahead of the caller's hook
  unsigned* d_fold_tickets = nullptr;  // per-slot arrival counters of a streaming launch that finishes its own second stage (kernels_poly.hpp LazyOut)
  fe_t* d_gate = nullptr;  // MAIL_RING challenge slots written by k_mail_gate (a streaming launch queued behind its gate reads its challenge here)
  void* h_pinned_fb = nullptr;  // pinned staging for asynchronous fixed-base jobs
  void* h_pinned_fbs = nullptr;  // pinned staging of the synchronous fixed-base calls (<= 1024 scalars: the per-round commitments of the ZK verifier circuit)
  // one-launch FixedBaseMul::multi_mul (sp_fbtables_multi_mul, kernels_msm.hpp k_multi_mul_coop): mapped pinned pages (result slot at byte 0, scalars
  // at byte 256), their device-side address, device scratch (ticket + block sums) and the sequence number of the result in flight
  // mapped pages of the <= 128-scalar fixed-base calls (k_fixed_base_rows_coop_mapped): [0] the synchronous calls on the main stream, [1] the job on the
  // auxiliary stream; each = 128 result slots of 128 B, then 128 scalars
  void* h_fbm[2] = {nullptr, nullptr};
  void* d_fbm[2] = {nullptr, nullptr};
  unsigned fbm_seq[2] = {0, 0};
  bool fb_async_busy = false;  // an sp_fixed_base_mul_h_begin job (mapped or copy form) has not been finished yet
  // lane 1 = the auxiliary stream (sp_fbtables_multi_mul*), lane 0 = the main stream (second walk of sp_hyrax_prove)
  void* h_mm[2] = {nullptr, nullptr};
  void* d_mm[2] = {nullptr, nullptr};
  void* d_mm_work[2] = {nullptr, nullptr};
  unsigned mm_seq[2] = {0, 0};
  unsigned mm_groups[2] = {0, 0};  // result slots of the launch in flight on the lane when its blocks join in groups (0: one slot at the head of the page)
  size_t mm_host_bytes[2] = {0, 0};  // scalars the last launch of the lane copied into its mapped page (mask / blind material): wiped when the result is collected
  hipEvent_t fb_ev = nullptr;
  hipEvent_t fb_event() {
    if (!fb_ev) hipEventCreateWithFlags(&fb_ev, hipEventDisableTiming);
    return fb_ev;
  }
  void* h_stage = nullptr;  // pinned staging of sp_table_write_async: STAGE_SLOTS x 64 KiB, reused round-robin behind an event per slot
  hipEvent_t stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned stage_next = 0;
  unsigned pending_slots = 0;  // > 0: the launch in flight delivers per-block sums in that many host slots (kernels_poly.hpp emit_partials)
  unsigned result_seq = 0;  // sequence number of the round result currently in flight (see kernels_poly.hpp publish_result)
  unsigned long long msm_jobs_issued[2] = {0, 0};
  hipEvent_t msm_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // completion event of the MSM job in each landing slot
  hipStream_t stream3 = nullptr;     // second auxiliary stream (sp_rowmat_vec_eq_begin), created on first use
  // window-table builds (capi_group.hip launch_window_tables): a lowest-priority stream of their own and its grow-only scratch; tab_mu orders the
  // enqueueing threads (the stream orders the builds, which share the scratch)
  hipStream_t stream_tab = nullptr;
  void* tab_scratch = nullptr;
  size_t tab_scratch_bytes = 0;
  std::mutex tab_mu;
  void* h_pinned_vec = nullptr;      // pinned landing buffer of sp_rowmat_vec_eq jobs, grow-only
  size_t h_pinned_vec_bytes = 0, h_pinned_vec_cols = 0;
  unsigned vec_seq = 0;  // sequence number of sp_rowmat_vec_eq_finish_scaled's arrival flags
  hipEvent_t vec_ev = nullptr;
  // sp_hyrax_prove: pinned landing buffer of LZ (main stream), its event, and the helper thread that hashes the commitment
  void* h_pcs = nullptr;
  size_t h_pcs_bytes = 0;
  hipEvent_t pcs_ev = nullptr;
  sp::Worker* pcs_worker = nullptr;
  // an opening announced ahead of PCS::prove (sp_hyrax_prove_announce, capi_group.hip): the inner sum-check's round loop reports its challenges to it
  struct sp_pcs_ahead* pcs_ahead = nullptr;
  void* h_pinned_lane[2] = {nullptr, nullptr};  // pinned landing buffers for per-window MSM sums (one per stream), 8 KiB each
  size_t pinned_elems = 0;
  bool timing = false;
  std::string timing_only;  // when non-empty, only this kernel class is instrumented (keeps event overhead out of a timed region)
  std::map<std::string, sp::KStat> stats;
  std::vector<hipEvent_t> event_pool;

  // grow-only persistent device buffers, one per slot, so hot-path calls never hipMalloc/hipFree
  enum { WS_MSM_ORDER = 0, WS_MSM_START, WS_MSM_BUCKETS, WS_MSM_WSUM, WS_SCALARS_RAW, WS_SCALARS_CANON, WS_FB_SCALARS, WS_FB_OUT, WS_ROWMAT_L,
         WS_ROWMAT_PART, WS_ROWMAT_OUT, WS_COMMIT_CANON, WS_COMMIT_FLAGS, WS_COMMIT_ROWS, WS_BASES_TMP, WS_MSM_FOLDED, WS_MSM_DIGITS, WS_NARROW_SCALARS, WS_NARROW_OUT, WS_NARROW_BLINDS, WS_POLYABC_PARTIALS, WS_POLYABC_TICKETS, WS_MSM_TASKS, WS_MSM_PARTIAL,
         WS_PER_LANE,
         WS_SLOTS = 2 * WS_PER_LANE };  // lane 1 = the auxiliary stream used by asynchronous MSM jobs
  void* ws_ptr[WS_SLOTS] = {};
  size_t ws_bytes[WS_SLOTS] = {};
  void* workspace(int slot, size_t bytes, int lane = 0);

  hipEvent_t get_event();
  int ensure_scratch(size_t elems);
  std::mutex stats_mu;  // the helper thread instruments the auxiliary streams while the owner instruments the main one
  // records (start, stop) events around `launch` on stream `st` when timing is enabled
  template <class L>
  void timed_on(hipStream_t st, const char* what, uint64_t alg_bytes, L&& launch) {
    if (!timing || (!timing_only.empty() && timing_only != what)) {
      launch();
      return;
    }
    hipEvent_t a, b;
    {
      std::lock_guard<std::mutex> l(stats_mu);
      a = get_event();
      b = get_event();
    }
    hipEventRecord(a, st);
    launch();
    hipEventRecord(b, st);
    std::lock_guard<std::mutex> l(stats_mu);
    sp::KStat& s = stats[what];
    s.pending.emplace_back(a, b);
    s.launches += 1;
    s.bytes += alg_bytes;
  }
  template <class L>
  void timed(const char* what, uint64_t alg_bytes, L&& launch) {
    timed_on(stream, what, alg_bytes, launch);
  }
  // one kernel: the events are attached to the dispatch itself (hipExtLaunchKernelGGL), so the pair brackets the kernel's execution and not the
  // launch latency of an idle stream as well — this is what the roofline kernel's duration is measured with
  template <class K, class... Args>
  void timed_kernel(const char* what, uint64_t alg_bytes, K kernel, dim3 grid, dim3 block, Args... args) {
    if (!timing || (!timing_only.empty() && timing_only != what)) {
      hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
      return;
    }
    hipEvent_t a, b;
    {
      std::lock_guard<std::mutex> l(stats_mu);
      a = get_event();
      b = get_event();
    }
    hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, a, b, 0, args...);
    std::lock_guard<std::mutex> l(stats_mu);
    sp::KStat& s = stats[what];
    s.pending.emplace_back(a, b);
    s.launches += 1;
    s.bytes += alg_bytes;
  }
  void drain_stats();
};

struct sp_table {
  sp_ctx* ctx = nullptr;
  fe_t* d = nullptr;
  size_t cap = 0;  // allocated elements
  size_t len = 0;  // logical length (power of two while used as a multilinear table)
  size_t lo_eff = (size_t)-1, hi_eff = (size_t)-1;
  bool view = false;  // non-owning window onto storage owned by another object (sp_nifs layers)
};

// A long absorb handed to the library's hashing thread (sp_transcript_set_async). The job owns its input and works on a COPY of the running hasher, so the
// worker never touches the transcript; the transcript installs the result when it next needs the sponge. If the worker has not picked the job up within
// ~20 us (a sleeping thread's wake-up can take milliseconds when the process is at its CPU quota) the waiting caller takes the job back and hashes inline.
struct sp_absorb_job {
  std::atomic<int> state{1};  // 1 posted, 2 running on the worker, 3 taken back by the caller, 4 done
  std::vector<uint8_t> data;
  size_t label = 0;
  sp::Keccak256State h;  // in: the running hasher; out: after the absorb
  void run() {
    h.update(data.data(), label);
    h.update(data.data() + label, data.size() - label);
  }
};
struct sp_transcript {
  sp::Transcript t;
  mutable std::shared_ptr<sp_absorb_job> pend;
  bool async_absorb = false;  // sp_transcript_set_async: a single-threaded caller opts in; a driver that already hashes on a thread of its own does not
  sp_transcript() = default;
  sp_transcript(const sp_transcript&) = delete;
  sp_transcript& operator=(const sp_transcript&) = delete;
  void join() const {
    if (!pend) return;
    sp_absorb_job& j = *pend;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const int st = j.state.load(std::memory_order_acquire);
      if (st == 4) break;
      if (st == 1 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(20)) {
        int expect = 1;
        if (j.state.compare_exchange_strong(expect, 3, std::memory_order_acq_rel)) {
          j.run();
          j.state.store(4, std::memory_order_release);
          break;
        }
      }
      sp::relax();
    }
    const_cast<sp_transcript*>(this)->t.h = j.h;
    pend.reset();
  }
  ~sp_transcript() { join(); }
};
struct sp_absorb_state {
  sp::Keccak256State h;
};

namespace sp {
inline size_t eff_lo(const sp_table* t) {
  size_t n = t->len / 2;
  return t->lo_eff < n ? t->lo_eff : n;
}
inline size_t eff_hi(const sp_table* t) {
  size_t n = t->len / 2;
  return t->hi_eff < n ? t->hi_eff : n;
}
inline size_t eff_pairs(const sp_table* t) {  // MultilinearPolynomial::eff_pairs (src/polys/multilinear.rs:78-84)
  size_t lo = eff_lo(t), hi = eff_hi(t);
  return lo > hi ? lo : hi;
}
inline void after_bind(sp_table* t) {  // multilinear.rs:160-163
  size_t n = t->len / 2, eff = eff_pairs(t);
  t->len = n;
  t->lo_eff = eff < n / 2 ? eff : n / 2;
  t->hi_eff = eff > n / 2 ? eff - n / 2 : 0;
}
int alloc_table(sp_ctx* ctx, size_t len, sp_table** out);
}  // namespace sp
