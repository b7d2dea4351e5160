// Host-side table walks for the commitments that sit in a transcript chain (capi_group.hip sp_hyrax_commit_split_*): a process-wide set of polling
// threads that add window-table entries into (X, Y, ZZ, ZZZ) accumulators.
//
// Why the host: a round commitment of the ZK verifier circuit (src/bellpepper/r1cs.rs:735-816 -> PCS::commit on the width-32 key,
// hyrax_pc.rs:221-260) is ~10 fixed-base multiplications whose result the transcript needs before the device may start the next sum-check round. Over
// the 16-bit-window tables that is ~130 dependent-free mixed additions. One addition is ~4.8 us on the device however many lanes are active (a chain of
// 256-bit products on a part built for throughput: kernels_msm.hpp, the cooperative addition) and ~0.35 us on a host core, so the device form is four
// tree levels + launch + PCIe = 26 us and a further 17 us of host work, while eight host cores walk 16 entries each in ~6 us. The device keeps every
// commitment that is wide (the witness rows, the MSMs of the opening); the host takes the ones that are narrow AND on the critical path.
//
// A batch is a list of table entries cut into `nparts` contiguous parts; part p is claimed by whoever gets there first (a polling walker or the owner
// itself, which never waits for a part nobody has claimed). The walkers poll one cache line of batch states while `keep_hot` says a prove is in
// flight and sleep on a condition variable otherwise.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "curve.hpp"

namespace sp {

class WalkPool {
 public:
  static constexpr int MAX_BATCHES = 8, MAX_PARTS = 32, MAX_ENTS = 16 * 40;
  typedef void (*PartFn)(void* arg, unsigned part, unsigned nparts);
  struct Batch {
    PartFn fn = nullptr;  // a parallel-for over the parts (run below); nullptr = the table walk over `ents`
    void* arg = nullptr;
    const aff_t* ents[MAX_ENTS];
    unsigned n_ents = 0, nparts = 0;
    xyzz_t part[MAX_PARTS];
    std::atomic<unsigned char> pdone[MAX_PARTS];  // part p's result is in part[p]
    std::atomic<bool> orphan{false};              // the owner has left without waiting for every claimed part: the last finisher gives the slot back
    void (*on_last)(void*) = nullptr;             // ... and calls this (the owner's reference on what the parts read and write)
    void* on_last_arg = nullptr;
    std::atomic<unsigned> done{0};
    unsigned gen = 0;
    int slot = -1;
  };

 private:
  // state word of batch slot i: generation (32) | nparts (16) | next unclaimed part (16); 0 = nothing posted
  alignas(64) std::atomic<uint64_t> states_[MAX_BATCHES];
  alignas(64) std::atomic<long long> hot_until_{0};
  Batch batches_[MAX_BATCHES];
  bool in_use_[MAX_BATCHES] = {};
  unsigned gen_ = 0;
  std::mutex mu_;  // slot allocation, thread start, sleeping walkers
  std::condition_variable cv_;
  std::vector<std::thread> th_;
  bool stop_ = false, started_ = false;
  std::atomic<unsigned long> redone_{0};
  int want_ = 0;

  // the CPUs that share the calling thread's last-level cache, minus the calling thread's own (false: not known, or fewer than three)
  static bool near_cpus(cpu_set_t* out, int* ncpus) {
    const char* e = getenv("SPARTAN_WALKERS_PIN");
    if (e && e[0] == '0') return false;
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    char path[128];
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[256] = {0};
    const bool got = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    CPU_ZERO(out);
    int count = 0;
    for (const char* p = buf; *p && *p != '\n';) {  // "0-7,128-135"
      char* end;
      const long a = strtol(p, &end, 10);
      long b = a;
      if (*end == '-') b = strtol(end + 1, &end, 10);
      for (long k = a; k <= b && k < CPU_SETSIZE; ++k)
        if (CPU_ISSET(k, &allowed) && first_sibling((int)k) == (int)k && first_sibling(cpu) != (int)k) {  // one logical CPU a core, not the caller's core
          CPU_SET(k, out);
          ++count;
        }
      p = *end == ',' ? end + 1 : end;
      if (end == p && *p) break;
    }
    if (count < 3) return false;
    *ncpus = count;
    return true;
  }
  static int first_sibling(int cpu) {  // lowest-numbered hardware thread of the core
    char path[128];
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
    FILE* f = fopen(path, "r");
    if (!f) return cpu;
    long a = cpu;
    if (fscanf(f, "%ld", &a) != 1) a = cpu;
    fclose(f);
    return (int)a;
  }
  static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  static void cpu_pause() { __builtin_ia32_pause(); }

  // claims one part of batch slot i; -1 when none is left (or nothing is posted)
  int claim(int i, unsigned* gen_out) {
    uint64_t s = states_[i].load(std::memory_order_acquire);
    for (;;) {
      const unsigned next = (unsigned)(s & 0xffffu), np = (unsigned)((s >> 16) & 0xffffu);
      if (next >= np) return -1;
      if (states_[i].compare_exchange_weak(s, s + 1, std::memory_order_acq_rel, std::memory_order_acquire)) {
        *gen_out = (unsigned)(s >> 32);
        return (int)next;
      }
    }
  }
  static xyzz_t walk_part(const Batch& b, unsigned p) {
    const unsigned lo = (unsigned)((uint64_t)b.n_ents * p / b.nparts), hi = (unsigned)((uint64_t)b.n_ents * (p + 1) / b.nparts);
    for (unsigned k = lo; k < hi; ++k) __builtin_prefetch(b.ents[k], 0, 0);  // each entry is a miss in a table of 64 MiB
    xyzz_t acc = xyzz_identity();
    for (unsigned k = lo; k < hi; ++k) acc = xyzz_add_mixed(acc, *b.ents[k]);
    return acc;
  }
  void run_part(Batch& b, unsigned p) {
    if (b.fn) b.fn(b.arg, p, b.nparts);
    else b.part[p] = walk_part(b, p);
    b.pdone[p].store(1, std::memory_order_release);
    // (a claimed part counts once, whoever else may have run it meanwhile; the slot of a batch its owner has left goes back with its last part)
    if (b.done.fetch_add(1, std::memory_order_acq_rel) + 1 == b.nparts && b.orphan.exchange(false, std::memory_order_acq_rel)) {
      void (*f)(void*) = b.on_last;
      void* a = b.on_last_arg;
      release(&b);
      if (f) f(a);
    }
  }
  void loop() {
    // SPARTAN_WALKERS_IDLE=1 puts the walkers into the lowest scheduling class (they then never take a core from a thread of the application or of the
    // library). Not the default: a walker of that class that holds a claimed part of a parallel region loses its core to ANY runnable thread for whole
    // scheduler ticks - 3.5-6.8 ms proves, 1 % of them at config 3 (tools/c3_soak.py, 3000 proves), against ~0.2 % of 7-18 ms proves (a helper thread
    // kept off its core) in the default class, which the run without walkers shows too. Idle polling offers the core every ~30 us in either class.
    {
      const char* e = getenv("SPARTAN_WALKERS_IDLE");
      struct sched_param sp0;
      sp0.sched_priority = 0;
      if (e && e[0] == '1' && sched_setscheduler(0, SCHED_IDLE, &sp0) != 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 19);
    }
    unsigned idle = 0;
    for (;;) {
      bool worked = false;
      for (int i = 0; i < MAX_BATCHES; ++i) {
        unsigned g;
        const int p = claim(i, &g);
        if (p >= 0) {
          run_part(batches_[i], (unsigned)p);  // the owner recycles a slot only after `done` has reached nparts: the batch is ours to read
          worked = true;
        }
      }
      if (worked) {
        idle = 0;
        continue;
      }
      for (int spin = 0; spin < 64; ++spin) cpu_pause();
      // a polling thread never blocks, and a pinned one never moves: a helper thread of the library that last ran on this core (the contexts' job
      // threads) would wait in its run queue for the end of a time slice - milliseconds (seen as one 18 ms prove in ~30 at config 3). Every ~30 us of
      // idle polling the core is offered to whoever is runnable on it.
      if ((++idle & 31u) == 0) sched_yield();
      if (now_ns() < hot_until_.load(std::memory_order_relaxed)) continue;
      std::unique_lock<std::mutex> l(mu_);
      cv_.wait(l, [&] { return stop_ || now_ns() < hot_until_.load(std::memory_order_relaxed); });
      if (stop_) return;
    }
  }
  WalkPool() {
    for (auto& s : states_) s.store(0, std::memory_order_relaxed);
    const char* e = getenv("SPARTAN_WALKERS");  // polling host threads for the split commitments; 0 = none (the callers keep the device form)
    int n = e ? atoi(e) : 8;
    const unsigned hw = std::thread::hardware_concurrency();
    if (!e && hw && (int)hw / 4 < n) n = (int)hw / 4;  // default: at most a quarter of the machine
    if (n < 0) n = 0;
    if (n > MAX_PARTS - 1) n = MAX_PARTS - 1;
    want_ = n;
  }
  ~WalkPool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
      hot_until_.store(0, std::memory_order_relaxed);
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }

 public:
  static WalkPool& get() {
    static WalkPool p;
    return p;
  }
  int walkers() const { return want_; }
  // a prove is in flight for the next `us` microseconds: the walkers poll
  void keep_hot(long us) {
    if (!want_) return;
    const long long until = now_ns() + 1000ll * us, cur = hot_until_.load(std::memory_order_relaxed);
    const bool cold = cur < now_ns();
    if (until > cur) hot_until_.store(until, std::memory_order_relaxed);
    if (!started_ || cold) {
      std::lock_guard<std::mutex> l(mu_);
      if (!started_) {
        started_ = true;
        // the walkers share the L3 of the core that first asks for them: a part's hand-shake is two cache-line transfers, ~25 ns inside a core complex
        // and 150-500 ns across complexes or sockets (SPARTAN_WALKERS_PIN=0: wherever the scheduler puts them)
        cpu_set_t set;
        int ncpus = 0;
        const bool pin = near_cpus(&set, &ncpus);
        if (pin && ncpus < want_) want_ = ncpus;  // a core each
        for (int i = 0; i < want_; ++i) {
          th_.emplace_back([this] { loop(); });
          if (pin) (void)pthread_setaffinity_np(th_.back().native_handle(), sizeof(set), &set);
        }
      }
      cv_.notify_all();
    }
  }
  // a free batch to fill (ents, n_ents), or nullptr when all slots are taken (the caller then walks on its own thread)
  Batch* acquire() {
    std::lock_guard<std::mutex> l(mu_);
    for (int i = 0; i < MAX_BATCHES; ++i)
      if (!in_use_[i]) {
        in_use_[i] = true;
        Batch& b = batches_[i];
        b.slot = i;
        b.fn = nullptr;
        b.arg = nullptr;
        b.n_ents = 0;
        b.nparts = 0;
        b.done.store(0, std::memory_order_relaxed);
        b.orphan.store(false, std::memory_order_relaxed);
        b.on_last = nullptr;
        b.on_last_arg = nullptr;
        for (auto& f : b.pdone) f.store(0, std::memory_order_relaxed);
        if (++gen_ == 0) ++gen_;
        b.gen = gen_;
        return &b;
      }
    return nullptr;
  }
  // cut into parts and make them claimable
  void post(Batch* b, unsigned nparts) {
    if (nparts < 1) nparts = 1;
    if (nparts > (unsigned)MAX_PARTS) nparts = MAX_PARTS;
    if (nparts > b->n_ents) nparts = b->n_ents ? b->n_ents : 1;
    b->fn = nullptr;
    b->nparts = nparts;
    states_[b->slot].store(((uint64_t)b->gen << 32) | ((uint64_t)nparts << 16), std::memory_order_release);
  }
  // The owner takes whatever is unclaimed, waits for the claimed rest, adds the parts and gives the slot back. A part whose walker does not deliver
  // within ~25 us (it normally takes 3-5: the thread has lost its core - to a helper thread of the library waking up on it, seen as one 8-18 ms prove in
  // ~500 at config 3) is walked again by the owner; the straggler's result is ignored and the slot goes back when it has finished (a walk only reads the
  // batch's own entry list and the key's tables: sp_ck_free waits for stragglers, `quiesce`).
  xyzz_t finish(Batch* b) {
    unsigned g;
    for (int p; (p = claim(b->slot, &g)) >= 0;) run_part(*b, (unsigned)p);
    xyzz_t acc = xyzz_identity();
    long long deadline = 0;
    for (unsigned p = 0; p < b->nparts; ++p) {
      bool have = b->pdone[p].load(std::memory_order_acquire) != 0;
      for (unsigned spins = 0; !have; ++spins) {
        cpu_pause();
        have = b->pdone[p].load(std::memory_order_acquire) != 0;
        if (!have && (spins & 63u) == 63u) {
          const long long t = now_ns();
          if (!deadline) deadline = t + 25000;
          else if (t > deadline) break;
        }
      }
      if (have) {
        acc = xyzz_add(acc, b->part[p]);
      } else {
        acc = xyzz_add(acc, walk_part(*b, p));
        redone_.fetch_add(1, std::memory_order_relaxed);
      }
    }
    if (b->done.load(std::memory_order_acquire) == b->nparts) {
      release(b);
    } else {
      b->orphan.store(true, std::memory_order_release);
      if (b->done.load(std::memory_order_acquire) == b->nparts && b->orphan.exchange(false, std::memory_order_acq_rel)) release(b);
    }
    return acc;
  }
  unsigned long redone() const { return redone_.load(std::memory_order_relaxed); }  // parts the owners walked again (diagnostics)
  // no walker is still inside a batch its owner has left (called before tables a walk may read are freed); bounded
  void quiesce() {
    const long long until = now_ns() + 2000000000ll;
    for (;;) {
      bool busy = false;
      {
        std::lock_guard<std::mutex> l(mu_);
        for (int i = 0; i < MAX_BATCHES; ++i) busy = busy || (in_use_[i] && batches_[i].orphan.load(std::memory_order_acquire));
      }
      if (!busy || now_ns() > until) return;
      std::this_thread::yield();
    }
  }
  // fn(arg, p, nparts) for p = 0 .. nparts - 1 on the walkers and the calling thread; returns when every part has run. Parts nobody has claimed are the
  // caller's (a sleeping walker costs nothing but its help), and with no free slot the caller runs them all. The host rounds of the sum-checks behind a
  // hand-over (capi_core.hip) are loops of a few hundred independent field products: this is their `par_iter`.
  void run(unsigned nparts, PartFn fn, void* arg) {
    if (nparts > (unsigned)MAX_PARTS) nparts = MAX_PARTS;
    Batch* b = nparts > 1 && want_ ? acquire() : nullptr;
    if (!b) {
      for (unsigned p = 0; p < nparts; ++p) fn(arg, p, nparts);
      return;
    }
    b->fn = fn;
    b->arg = arg;
    b->nparts = nparts;
    states_[b->slot].store(((uint64_t)b->gen << 32) | ((uint64_t)nparts << 16), std::memory_order_release);
    unsigned g;
    for (int p; (p = claim(b->slot, &g)) >= 0;) run_part(*b, (unsigned)p);
    while (b->done.load(std::memory_order_acquire) < b->nparts) cpu_pause();
    release(b);
  }
  // the same region posted and collected in two calls: the parts run on the walkers while the owner does something else; `wait_fn` takes the parts nobody
  // has claimed and waits for the rest (`arg` must stay alive until then). nullptr: no free slot / no walkers - the caller runs fn itself when it needs it.
  Batch* post_fn(unsigned nparts, PartFn fn, void* arg) {
    if (nparts > (unsigned)MAX_PARTS) nparts = MAX_PARTS;
    Batch* b = nparts >= 1 && want_ ? acquire() : nullptr;
    if (!b) return nullptr;
    b->fn = fn;
    b->arg = arg;
    b->nparts = nparts;
    states_[b->slot].store(((uint64_t)b->gen << 32) | ((uint64_t)nparts << 16), std::memory_order_release);
    return b;
  }
  // The same for a region whose parts are IDEMPOTENT and whose data the caller can keep alive: the owner takes what is unclaimed, gives every claimed part
  // `timeout_ns` (from its first wait) and runs the late ones again itself - a walker that has lost its core (to a helper thread of the library waking
  // on it: milliseconds) no longer holds the owner. Returns true when stragglers are still out; `end_fn` then leaves the slot - and `on_last(arg)`, the
  // caller's reference on the region's data - to the last of them, and otherwise does both at once.
  bool collect_fn(Batch* b, long long timeout_ns) {
    unsigned g;
    for (int p; (p = claim(b->slot, &g)) >= 0;) run_part(*b, (unsigned)p);
    long long deadline = 0;
    for (unsigned p = 0; p < b->nparts; ++p) {
      bool have = b->pdone[p].load(std::memory_order_acquire) != 0;
      for (unsigned spins = 0; !have; ++spins) {
        cpu_pause();
        have = b->pdone[p].load(std::memory_order_acquire) != 0;
        if (!have && (spins & 63u) == 63u) {
          const long long t = now_ns();
          if (!deadline) deadline = t + timeout_ns;
          else if (t > deadline) break;
        }
      }
      if (!have) {
        b->fn(b->arg, p, b->nparts);
        redone_.fetch_add(1, std::memory_order_relaxed);
      }
    }
    return b->done.load(std::memory_order_acquire) < b->nparts;
  }
  void end_fn(Batch* b, void (*on_last)(void*), void* arg) {
    if (b->done.load(std::memory_order_acquire) == b->nparts) {
      release(b);
      if (on_last) on_last(arg);
      return;
    }
    b->on_last = on_last;
    b->on_last_arg = arg;
    b->orphan.store(true, std::memory_order_release);
    if (b->done.load(std::memory_order_acquire) == b->nparts && b->orphan.exchange(false, std::memory_order_acq_rel)) {
      release(b);
      if (on_last) on_last(arg);
    }
  }
  void wait_fn(Batch* b) {
    unsigned g;
    for (int p; (p = claim(b->slot, &g)) >= 0;) run_part(*b, (unsigned)p);
    while (b->done.load(std::memory_order_acquire) < b->nparts) cpu_pause();
    release(b);
  }
  bool finished(const Batch* b) const { return b->done.load(std::memory_order_acquire) >= b->nparts; }
  void release(Batch* b) {
    // (a walk's entry pointers are the digits of its scalars - blinds among them - and its part sums partial commitments: nothing of either stays behind)
    if (!b->fn && b->n_ents) {
      explicit_bzero(b->ents, b->n_ents * sizeof(b->ents[0]));
      explicit_bzero(b->part, b->nparts * sizeof(b->part[0]));
    }
    states_[b->slot].store(0, std::memory_order_release);
    std::lock_guard<std::mutex> l(mu_);
    in_use_[b->slot] = false;
  }
};

}  // namespace sp
