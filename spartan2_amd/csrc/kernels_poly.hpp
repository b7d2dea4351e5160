// HIP kernels for the multilinear-table side of the sum-check (gfx950, wave64).
//   k_bind_top      K1  MultilinearPolynomial::bind_poly_var_top        src/polys/multilinear.rs:95-164
//   k_eval_cubic    K2  EqSumCheckInstance::evaluation_points_cubic...   src/sumcheck.rs:1025-1156 (+ t(-1) of :1327-1396)
//   k_eval_quad     K3  compute_eval_points_quad                         src/sumcheck.rs:128-174
//   k_eq_levels/k_eq_outer  K8  EqPolynomial::evals_from_points, EqSumCheckInstance::new tables  src/polys/eq.rs:59-117, src/sumcheck.rs:956-992
//   k_sum_partials       second stage of every block-partial reduction
// All arithmetic is 256-bit modular integer work on VALU (v_mad_u64_u32 + carry chains); there is no MFMA
// anywhere. Tables are arrays of 32-byte elements; lane l touches element base+l, so a wave reads/writes
// 2 KiB contiguous per table access (two dwordx4 per lane).
#pragma once
#include "device_utils.hpp"

namespace spk {

// Round results land in a mapped, pinned host buffer: elements [0..3) = sums, word 0 of element RESULT_FLAG_ELEM = sequence
// number, stored with system-scope release after the data so the host can poll it instead of paying a stream synchronise.
constexpr int RESULT_FLAG_ELEM = 7;
__device__ __forceinline__ void publish_result(fe_t* result, unsigned seq) {
  __threadfence_system();
  __hip_atomic_store(reinterpret_cast<unsigned*>(result + RESULT_FLAG_ELEM), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Block partials of an evaluation launch. Up to HOST_SUM_MAX_BLOCKS blocks: every block writes its NACC sums and the sequence number into its own
// 128-byte slot of the mapped pinned buffer and the HOST adds them (a few dozen 256-bit additions) — no second-stage launch on the per-round
// critical path. Larger grids store to device memory for k_sum_partials.
constexpr int HOST_SUM_MAX_BLOCKS = 64;
constexpr int SLOT_BASE_ELEM = 64;  // element index of slot 0 in the mapped buffer; slot b = 4 elements: sums[0..3), word 0 of the 4th = sequence
// A slot is self-validating: element 3 carries the sequence number TWICE (words 0 and 3) and two independent check words over the data
// (word 1 = sequence + plain sum, word 2 = sequence * K + position-weighted sum), so the host accepts a slot only when all of it has landed,
// whatever order the stores reach host memory in, and the producer needs no fence for it. A torn read would have to match both 32-bit
// checks (2^-64 for unrelated stale words) and both copies of the sequence number.
struct slot_chk {
  unsigned a, b;
};
constexpr unsigned SLOT_CHK_K = 0x9E3779B1u;
// the same arithmetic on the host side of capi_core.hip (wait_slot / reduce_partials_wait)
__host__ __device__ __forceinline__ void slot_chk_add(slot_chk& c, const fe_t& v, int k) {
  // c.b += sum_i (8k + i + 1) v_i, written as (8k + 1) sum_i v_i + sum_i i v_i (the same value mod 2^32): with a run-time k the weights are ONE per-lane
  // value instead of eight - the resident tail's compiler hoisted the eight out of its round loop and spilled them (tools/spill_report.py)
  unsigned plain = 0, ramp = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    plain += v.v[i];
    ramp += (unsigned)i * v.v[i];
  }
  c.a += plain;
  c.b += (unsigned)(8 * k + 1) * plain + ramp;
}
// (mapped host memory is uncached on the device side: plain stores go straight out, as two 16-byte writes per element and two 8-byte tag halves)
__device__ __forceinline__ void slot_store_elem(fe_t* dst, const fe_t& v) { *dst = v; }
__device__ __forceinline__ void slot_store_tag(fe_t* slot, unsigned seq, const slot_chk& c) {
  const unsigned long long hi = ((unsigned long long)seq << 32) | (seq * SLOT_CHK_K + c.b);  // words 2, 3
  const unsigned long long lo = ((unsigned long long)(seq + c.a) << 32) | seq;               // words 0, 1 (word 0 is what the host polls)
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(&slot[3].v[2]), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(&slot[3].v[0]), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the same slot format in the hand-over area (HAND_BASE_ELEM: 342 slots) for a launch of more than HOST_SUM_MAX_BLOCKS blocks whose sums the host adds itself
template <int NACC>
__device__ __forceinline__ void emit_partials_wide(const fe_t (&acc)[NACC], fe_t* __restrict__ mapped, unsigned seq);
template <int NACC>
__device__ __forceinline__ void emit_partials(const fe_t (&acc)[NACC], fe_t* __restrict__ partials, fe_t* __restrict__ mapped, unsigned seq) {
  if (gridDim.x <= HOST_SUM_MAX_BLOCKS) {
    fe_t* slot = mapped + SLOT_BASE_ELEM + 4 * blockIdx.x;
    slot_chk chk = {0u, 0u};
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
      slot_store_elem(slot + k, acc[k]);
      slot_chk_add(chk, acc[k], k);
    }
    slot_store_tag(slot, seq, chk);
  } else {
#pragma unroll
    for (int k = 0; k < NACC; ++k) partials[(size_t)blockIdx.x * NACC + k] = acc[k];
  }
}


// Where a streaming launch leaves its block partials (stream_block_partials below) and, when `tickets` is set, how the LAST block of each slot finishes them:
// the second stage (k_sum_partials_lazy) folded into its producer. nslots host result slots; group g = block >> gl belongs to slot (g / 64) % nslots, a lane
// per group, exactly the second-stage kernel's split; per_slot blocks arrive at a slot's ticket (left at zero again by the last one).
struct LazyOut {
  lazy9_t* P;
  unsigned* tickets;  // nullptr: partials only, a second-stage launch follows
  const fe_t* eq_out;
  fe_t* mapped;
  unsigned seq, nslots, per_slot;
  int gl;
  LazyOut() = default;
  __host__ __device__ LazyOut(lazy9_t* p) : P(p), tickets(nullptr), eq_out(nullptr), mapped(nullptr), seq(0), nslots(0), per_slot(0), gl(0) {}
};

// ---- challenge mailbox ------------------------------------------------------------------------------------------------------------------
// A kernel that binds with a challenge the host has not drawn yet is launched AHEAD of it and picks the challenge up from a 64-byte mailbox line:
// words 0..7 = challenge, 8 = the result sequence number it answers, 9 = check word (sequence + sum of the challenge words), 10 = second check
// word (sequence * K + position-weighted sum), 11 = the sequence number again, 12 = abort (the host gave the sum-check up: stop waiting). The line lives in
// fine-grained DEVICE memory that the host writes through the PCIe BAR (capi_core.hip post_challenge) — so a thousand blocks can poll it without
// touching the bus — or, without a large BAR, in mapped host memory (resident tail only). Lanes 0..12 of wave 0 read the thirteen words in ONE
// instruction; a poll that straddles the host's stores fails the check and is repeated. Never hangs: after MAIL_WATCHDOG_TICKS (8 s) the error word in the mapped
// result buffer is set and the kernel carries on with whatever it read.
constexpr int TAIL_ERR_ELEM = 10, TAIL_FINAL_ELEM = 16;  // element indices in the mapped result buffer
// TWO ROUNDS PER ROUND TRIP. Once the resident tail is down to one block, a trip over the bus (result out, challenge back: 8 - 12 us) carries two
// rounds: besides the sums of the round over the current table (n entries) the kernel sends the sums of the round AFTER it as polynomials in the
// challenge r it does not know yet — the bound table is A[x] + r (A[x + n/2] - A[x]), so every sum of the next round is of degree 2 in r: three
// coefficient sums each, of which the products share most factors with this round's (8 sums for the quadratic sum-check, 12 for the cubic one).
// The host finishes this round, draws r, evaluates the three coefficients at r, finishes the next round, draws its challenge and posts BOTH;
// the kernel binds twice. Same polynomials, same transcript, half the trips. The result takes TWO sequence numbers, one per challenge that answers
// it, and goes out through the result slots of blocks 0..3 (only block 0 is left in this regime): three sums and a tag per 128-byte slot, so every
// line that carries data also carries a system-scope tag store, which is what pushes a line of the mapped buffer out to host memory (sums in a
// region of their own, written with plain stores, stayed in the L2 until the kernel ended; system-scope stores for every word cost more than the trip saved).
constexpr int TAIL_WIDE_VALS = 12;
constexpr int TAIL_DOUBLE_SUMS_QUAD = 8, TAIL_DOUBLE_SUMS_CUBIC = 12;
// HAND-OVER OF THE LAST ROUNDS. Once a bind leaves tables of n <= hand_n entries, the kernel sends the TABLES instead of sums (n elements per table, three
// per 128-byte slot of the hand-over area, each slot tagged like a result slot), the host binds and evaluates the remaining rounds itself and hands the
// final claims back through the mailbox - lines answering seq, seq + 1 (, seq + 2) - which the kernel, otherwise idle, stores into element 0 of the tables
// (the ABI's "bound in place down to length 1") before it leaves. Same polynomials, same transcript.
// hand_n is the launch's (TailArgs::hand_n, chosen by the host: capi_core.hip tail_hand_n): 16 (cubic) / 32 (quadratic) entries - the rounds a single host
// thread runs in ~3 us; above that a trip carries two rounds (tail_double). Round 6 built the larger form - the hand-over at the moment ONE block is left,
// n = 256 / 512, the host's rounds spread over the process's polling threads (walk_pool.hpp) - and measured it behind this one (see tail_hand_n); the
// kernel takes either.
constexpr int TAIL_HAND_OVER_MAX_VALS = 1024;  // 3 x 256 or 2 x 512 elements: 342 slots of the hand-over area
constexpr unsigned long long TAIL_HAND_N_MAX_CUBIC = 256, TAIL_HAND_N_MAX_QUAD = 512;  // one block left (q <= TAIL_WIDE_Q[_CUBIC]): its LDS holds the tables
#ifdef SP_TAIL_NO_HAND_OVER  // A/B builds
__host__ __device__ __forceinline__ bool tail_hand_over(unsigned long long, unsigned long long) { return false; }
#else
__host__ __device__ __forceinline__ bool tail_hand_over(unsigned long long n, unsigned long long hand_n) { return n >= 2 && n <= hand_n; }
#endif
#ifdef SP_TAIL_SINGLE  // A/B builds: one round per trip everywhere
__host__ __device__ __forceinline__ bool tail_double(bool, unsigned long long, unsigned long long) { return false; }
#else
__host__ __device__ __forceinline__ bool tail_double(bool cubic, unsigned long long n, unsigned long long hand_n) {
  return !tail_hand_over(n, hand_n) && n >= 4 && n <= (cubic ? 256ull : 512ull);
}
#endif
// The mailbox is a RING of MAIL_RING 64-byte lines indexed by the sequence number a challenge answers (line = seq & 7): a waiter only ever looks at the
// line of ITS challenge, which the host does not touch again before eight more results are in. With the ring in device memory the host also keeps a
// MIRROR of it in the mapped pinned buffer (MAIL_MIRROR_ELEM: written first, by ordinary stores); a waiter whose device line has not answered within
// MAIL_MIRROR_AFTER_TICKS starts looking at the mirror as well (one PCIe read every MAIL_MIRROR_EVERY_TICKS): a second, independent path for the same
// 64 bytes. Which path answered is counted in the diagnostics words (device memory, MAIL_DIAG_WORD of the mailbox page; sp_ctx_mail_stats).
constexpr int MAIL_RING = 8, MAIL_LINE_WORDS = 16;
constexpr int MAIL_MIRROR_ELEM = SLOT_BASE_ELEM + 4 * HOST_SUM_MAX_BLOCKS;  // 16 elements = 8 lines
constexpr int HAND_BASE_ELEM = MAIL_MIRROR_ELEM + 16;  // the hand-over area: (TAIL_HAND_OVER_MAX_VALS + 2) / 3 slots of 4 elements
constexpr int MAPPED_ELEMS = HAND_BASE_ELEM + 4 * ((TAIL_HAND_OVER_MAX_VALS + 2) / 3);
constexpr int WIDE_SLOTS = (TAIL_HAND_OVER_MAX_VALS + 2) / 3;
template <int NACC>
__device__ __forceinline__ void emit_partials_wide(const fe_t (&acc)[NACC], fe_t* __restrict__ mapped, unsigned seq) {
  fe_t* slot = mapped + HAND_BASE_ELEM + 4 * blockIdx.x;
  slot_chk chk = {0u, 0u};
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    slot_store_elem(slot + k, acc[k]);
    slot_chk_add(chk, acc[k], k);
  }
  slot_store_tag(slot, seq, chk);
}
constexpr int MAIL_DIAG_WORD = 256;  // word offset in the device mailbox page: [0] answers taken from the mirror, [1] watchdog trips, [2] / [3] want / device-line seq of the last mirror answer
constexpr unsigned long long MAIL_MIRROR_AFTER_TICKS = 3000ull, MAIL_MIRROR_EVERY_TICKS = 1000ull;  // 30 us, 10 us at the 100 MHz wall clock
// 8 s at the 100 MHz wall clock. Long on purpose: an owner thread that the host's scheduler keeps away from its CPU (a throttled cgroup: the bench
// boxes run under a 16-CPU quota) is late, not gone, and a late challenge only costs time while a tripped watchdog costs the proof.
constexpr unsigned long long MAIL_WATCHDOG_TICKS = 800000000ull;
struct MailRef {
  const unsigned* mail;    // nullptr: the challenge is the kernel argument; else word address of line 0 of the ring
  const unsigned* mirror;  // device-side address of the host-memory mirror of the ring (nullptr when `mail` is in host memory itself)
  fe_t* mapped;            // mapped pinned result buffer (error word at TAIL_ERR_ELEM)
  unsigned answers;        // sequence number of the round result the awaited challenge answers
  const fe_t* gated;       // non-null: the launch sits behind k_mail_gate in its stream and finds its challenge here (no polling)
};
// wave-uniform: does the 13-word line held in lanes 0..12 of `w` carry the challenge answering `want`? (-1: the host aborted)
__device__ __forceinline__ int mail_line_state(unsigned w, unsigned want, int lane) {
  const unsigned flag = __shfl(w, 8, 64), chk = __shfl(w, 9, 64), chk2 = __shfl(w, 10, 64), flag2 = __shfl(w, 11, 64);
  if (__shfl(w, 12, 64) != 0u) return -1;
  if (flag != want || flag2 != want) return 0;
  unsigned sum = lane < 8 ? w : 0u, wsum = lane < 8 ? (unsigned)(lane + 1) * w : 0u;
#pragma unroll
  for (int m = 4; m >= 1; m >>= 1) {
    sum += __shfl_xor(sum, m, 64);
    wsum += __shfl_xor(wsum, m, 64);
  }
  sum = __shfl(sum, 0, 64) + flag;
  wsum = __shfl(wsum, 0, 64) + flag * SLOT_CHK_K;
  return (sum == chk && wsum == chk2) ? 1 : 0;
}
__device__ __forceinline__ bool mail_wait(const unsigned* mail, const unsigned* mirror, fe_t* mapped, unsigned want, fe_t* r_smem) {
  __shared__ int ok;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned* line = mail + MAIL_LINE_WORDS * (want & (MAIL_RING - 1));
    const unsigned* mline = mirror ? mirror + MAIL_LINE_WORDS * (want & (MAIL_RING - 1)) : nullptr;
    const unsigned long long t0 = wall_clock64();
    unsigned long long next_mirror = t0 + MAIL_MIRROR_AFTER_TICKS;
    int good = 0, aborted = 0, from_mirror = 0;
    unsigned w = 0, wdev = 0;
    for (unsigned it = 0;; ++it) {
      if (lane < 13) w = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // the poll that finds nothing must be short - its period is the delay between the host's store and the kernel's start: two lane reads decide
      // "not this challenge yet" (the sequence words, the abort word); the check sums are only formed for a line that carries the right sequence
      // number, and the wall clock (mirror path, watchdog) is read every 16th time
      int st = 0;
      if (__builtin_amdgcn_readlane((int)w, 12) != 0 || ((unsigned)__builtin_amdgcn_readlane((int)w, 8) == want && (unsigned)__builtin_amdgcn_readlane((int)w, 11) == want))
        st = mail_line_state(w, want, lane);
      const bool tick = (it & 15u) == 15u;
      if (st == 0 && mline && tick) {
        const unsigned long long now = wall_clock64();
        if (now >= next_mirror) {  // the second path: the same line in host memory
          wdev = w;
          unsigned wm = 0;
          if (lane < 13) wm = __hip_atomic_load(mline + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          st = mail_line_state(wm, want, lane);
          if (st != 0) {
            w = wm;
            from_mirror = 1;
          }
          next_mirror = now + MAIL_MIRROR_EVERY_TICKS;
        }
      }
      if (st < 0) {  // aborted by the host (error exit of the round loop): leave without raising the watchdog error
        aborted = 1;
        break;
      }
      if (st > 0) {
        good = 1;
        break;
      }
      if (tick && wall_clock64() - t0 > MAIL_WATCHDOG_TICKS) break;  // the host went away: never hang the device
    }
    if (lane < 8) r_smem->v[lane] = w;
    if (mline && lane == 0 && (from_mirror || (!good && !aborted))) {  // diagnostics (device memory: atomics are fine there)
      unsigned* diag = const_cast<unsigned*>(mail) + MAIL_DIAG_WORD;
      if (from_mirror) {
        atomicAdd(diag + 0, 1u);
        diag[2] = want;
      } else {
        atomicAdd(diag + 1, 1u);
      }
    }
    if (mline && from_mirror && lane == 8) const_cast<unsigned*>(mail)[MAIL_DIAG_WORD + 3] = wdev;  // what the device line showed when the mirror answered
    if (!good && !aborted) {  // watchdog: leave what this poll saw next to the error word (diagnostics for the host's error message)
      const unsigned seen = lane < 13 ? w : 0u;
      if (lane < 13) reinterpret_cast<unsigned*>(mapped + TAIL_ERR_ELEM + 1)[lane] = seen;
      if (lane == 0) {
        unsigned* e = reinterpret_cast<unsigned*>(mapped + TAIL_ERR_ELEM);
        e[1] = want;
        e[2] = blockIdx.x;
        e[3] = gridDim.x;
        __hip_atomic_store(e, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (lane == 0) ok = good;
  }
  __syncthreads();
  return ok != 0;
}
// the challenge of a fused bind+evaluate kernel: its argument, or the mailbox when it was launched ahead (all threads of the block must call)
// (read through the constant address space: a wave-uniform scalar load - the eight words arrive in scalar registers like a challenge passed as a
// kernel argument, and the scalar cache starts every kernel empty, so the gate's store is what is read)
__device__ __forceinline__ fe_t gated_challenge(const fe_t* p) {
  typedef const __attribute__((address_space(4))) uint32_t* const_words;
  const_words q = (const_words)(uintptr_t)p;
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = q[i];
  return r;
}
__device__ __forceinline__ fe_t challenge_or(const MailRef& m, const fe_t& r_arg) {
  if (m.gated) return gated_challenge(m.gated);
  if (!m.mail) return r_arg;
  __shared__ fe_t r_sh;
  mail_wait(m.mail, m.mirror, m.mapped, m.answers, &r_sh);
  return r_sh;
}

// The gate of a streaming launch (capi_core.hip gate_launch): one wave waits at the mailbox and stores the challenge for the launch queued behind it.
// On an abort or a watchdog trip the slot gets whatever was read: the launch behind it computes garbage on tables the host has given up.
__global__ void __launch_bounds__(64) k_mail_gate(MailRef m, fe_t* __restrict__ out) {
  __shared__ fe_t r_sh;
  mail_wait(m.mail, m.mirror, m.mapped, m.answers, &r_sh);
  if (threadIdx.x < 8) out->v[threadIdx.x] = r_sh.v[threadIdx.x];
}

// ---- K1: bind the top variable of up to 8 tables with the same challenge -----------------------------------
constexpr int BIND_MAX_TABLES = 8;  // (the six tables of a batched outer round go in one launch)
struct BindArgs {
  fe_t* z[BIND_MAX_TABLES];
  unsigned long long n[BIND_MAX_TABLES];   // half length of each table
  unsigned long long lo[BIND_MAX_TABLES];  // min(lo_eff, n)
  unsigned long long hi[BIND_MAX_TABLES];  // min(hi_eff, n)
  fe_t r;
  fe_t one_minus_r;
};
// grid = (blocks, ntables). In place: thread i reads Z[i], Z[n+i], writes Z[i].
__global__ void __launch_bounds__(256) k_bind_top(BindArgs a) {
  const int t = blockIdx.y;
  fe_t* __restrict__ Z = a.z[t];
  const unsigned long long n = a.n[t], lo = a.lo[t], hi = a.hi[t];
  const unsigned long long eff = lo > hi ? lo : hi;
  const unsigned long long both = lo < hi ? lo : hi;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < eff; i += (unsigned long long)gridDim.x * blockDim.x) {
    fe_t out;
    if (i < both) {  // a + r (b - a)
      fe_t lo_v = Z[i], hi_v = Z[n + i];
      out = fe_add<S>(lo_v, fe_mul<S>(a.r, fe_sub<S>(hi_v, lo_v)));
    } else if (i < lo) {  // high half known zero: a (1 - r)
      out = fe_mul<S>(Z[i], a.one_minus_r);
    } else {  // low half known zero: r b
      out = fe_mul<S>(a.r, Z[n + i]);
    }
    Z[i] = out;
  }
}

// ---- K8: eq tables -------------------------------------------------------------------------------------------
// One block builds every prefix level of the eq table over v[0..m): level k (2^k entries, at out + 2^k - 1 ... see
// level_offset) is the table over the LAST k variables, with the earliest of them on the index MSB
// (compute_eq_polynomials, src/sumcheck.rs:960-979; EqPolynomial::evals_from_points, src/polys/eq.rs:66-76).
__host__ __device__ __forceinline__ size_t eq_level_offset(int k) { return ((size_t)1 << k) - 1; }
// One level of the pyramid per barrier. The levels of up to 1024 entries are kept in LDS as well (in place: entry i of level k becomes entries i and
// 2^k + i of level k + 1), so a level costs an LDS round trip + one product instead of a store to and a load from the L2 (1.5 -> 0.7 us per level; the
// pyramids of tau and of r_x stand in front of the first evaluation of the outer and of the inner sum-check); the global stores are fire-and-forget.
constexpr int EQ_LDS_ENTRIES = 1024;
__device__ __forceinline__ void eq_levels_block(const fe_t* v_rev /* v_rev[k] = the challenge of level k */, int m, fe_t* __restrict__ out, fe_t* lv) {
  if (threadIdx.x == 0) {
    out[0] = fe_one<S>();
    lv[0] = fe_one<S>();
  }
  __syncthreads();
  for (int k = 0; k < m; ++k) {
    const fe_t r = v_rev[-k];
    fe_t* next = out + eq_level_offset(k + 1);
    const size_t size = (size_t)1 << k;
    if (2 * size <= (size_t)EQ_LDS_ENTRIES) {
      for (size_t i = threadIdx.x; i < size; i += blockDim.x) {
        const fe_t e = lv[i];
        const fe_t y = fe_mul<S>(e, r), x = fe_sub<S>(e, y);
        lv[size + i] = y;
        lv[i] = x;
        next[size + i] = y;
        next[i] = x;
      }
    } else {
      const fe_t* prev = out + eq_level_offset(k);
      for (size_t i = threadIdx.x; i < size; i += blockDim.x) {
        const fe_t e = prev[i];
        const fe_t y = fe_mul<S>(e, r);
        next[size + i] = y;
        next[i] = fe_sub<S>(e, y);
      }
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(1024) k_eq_levels(const fe_t* __restrict__ v, int m, fe_t* __restrict__ out) {
  __shared__ fe_t lv[EQ_LDS_ENTRIES];
  eq_levels_block(v + (m - 1), m, out, lv);
}
// both pyramids of EqSumCheckInstance::new (src/sumcheck.rs:956-992) in one launch: block 0 the left one, block 1 the right one, the taus by value
// (no upload in front of the first evaluation)
struct EqPairArgs {
  fe_t v[2][16];
  int m[2];
  fe_t* out[2];
};
__global__ void __launch_bounds__(1024) k_eq_levels_pair(EqPairArgs a) {
  __shared__ fe_t lv[EQ_LDS_ENTRIES];
  const int b = blockIdx.x;
  const int m = a.m[b];
  eq_levels_block(&a.v[b][m - 1], m, a.out[b], lv);
}
// The same table with the LAST K variables (2 <= K <= 4) applied here: T_lo covers only the low variables up to them, so both pyramids can be built K
// rounds before the sum-check that draws the point ends (K = 4: when its resident kernel hands the last rounds to the host).
// out[(hi << 10) | lo] = T_hi[hi] (T_lo[lo >> K] e_K[lo & (2^K - 1)]), lo_bits = 10. The 2^K weights e_K arrive BY VALUE (the host forms them: 2^(K+1) - 4
// products, ~0.5 us, where a block's first 2^K threads spent K - 1 dependent device products and a barrier in front of every other wave: 19.8 -> 16.2 us
// at 2^20). A block of EQ_LASTK_BLOCK threads = one half of the 1024 low indices, thread = lo: it forms ITS low factor once (one product), then EQ_LASTK_HPB
// high entries (uniform loads, issued before the first product), one product and one 32-byte store each - 1 + 1 / HPB products per output, every store of
// a wave 2 KiB contiguous. 512 x 8 measured best of {256, 512, 1024} x {1, 2, 4, 8, 16}: 14.6 us at 2^20, 1.3 us above the same stores without arithmetic
// (tools/eq_bench.hip, profiles/r05_eq_bench.txt; lane-contiguous 16-byte stores through the wave's LDS: no gain).
constexpr int EQ_LASTK_HPB = 8;
constexpr int EQ_LASTK_BLOCK = 512;
struct EqLastK {
  fe_t w[16];  // e_K[j], j < 2^K: first of the K variables = most significant of the K bits
};
__global__ void __launch_bounds__(EQ_LASTK_BLOCK) k_eq_outer_lastk(const fe_t* __restrict__ t_hi, const fe_t* __restrict__ t_lo, int K, size_t n_hi, EqLastK wk,
                                                                   fe_t* __restrict__ out) {
  constexpr unsigned SLICES = 1024 / EQ_LASTK_BLOCK;
  const unsigned t = threadIdx.x + (blockIdx.x % SLICES) * EQ_LASTK_BLOCK;
  const size_t hi0 = (size_t)(blockIdx.x / SLICES) * EQ_LASTK_HPB;
  fe_t th[EQ_LASTK_HPB];
#pragma unroll
  for (int h = 0; h < EQ_LASTK_HPB; ++h) th[h] = t_hi[hi0 + h < n_hi ? hi0 + h : 0];
  const fe_t lo = fe_mul<S>(t_lo[t >> K], wk.w[t & ((1u << K) - 1)]);
#pragma unroll
  for (int h = 0; h < EQ_LASTK_HPB; ++h) {
    if (hi0 + h >= n_hi) break;
    out[((hi0 + h) << 10) + t] = fe_mul<S>(th[h], lo);
  }
}
// out[(hi << lo_bits) | lo] = T_hi[hi] * T_lo[lo]
__global__ void __launch_bounds__(256) k_eq_outer(const fe_t* __restrict__ t_hi, const fe_t* __restrict__ t_lo, int lo_bits, size_t total,
                                                  fe_t* __restrict__ out) {
  const size_t mask = ((size_t)1 << lo_bits) - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    out[i] = fe_mul<S>(t_hi[i >> lo_bits], t_lo[i & mask]);
}

// ---- K2: cubic evaluation sums with split-eq weights ------------------------------------------------------------
// For pair index id in [0, half): weight E(id) = eq_out[id >> s] * eq_in[id & (2^s - 1)].
//   t0   = sum E (A0 B0 - C0)
//   tinf = sum E (A1 - A0)(B1 - B0)
//   tm1  = sum E ((2A0 - A1)(2B0 - B1) - (2C0 - C1))      [only when WITH_M1: the tau == 0 fallback]
// MODE 0: one table, E = eq_in[id]                       (second-half rounds, src/sumcheck.rs:1107-1147)
// MODE 1: factored — a block's chunk lies inside one x_out, block sum is multiplied by eq_out once
// MODE 2: direct — per-pair product eq_out * eq_in (tiny tables where a chunk spans several x_out)
constexpr int EVAL_PPT = 1;  // pairs per thread (1: the kernels are latency-bound per lane; more waves hide it better than more work per lane)
// ZC: the zero-check round-0 form (evaluation_points_zero_check_round0, src/sumcheck.rs:1163-1271): only t_inf is computed, C is not read, t0 = 0.
template <int MODE, bool WITH_M1, bool ZC = false>
__global__ void __launch_bounds__(256) k_eval_cubic(const fe_t* __restrict__ A, const fe_t* __restrict__ B, const fe_t* __restrict__ C, size_t half,
                                                    const fe_t* __restrict__ eq_in, const fe_t* __restrict__ eq_out, int s,
                                                    fe_t* __restrict__ partials, fe_t* __restrict__ single_out, unsigned seq) {
  constexpr int NACC = WITH_M1 ? 3 : 2;
  __shared__ fe_t smem[NACC * 4];
  const size_t chunk = (size_t)blockDim.x * EVAL_PPT;
  const size_t base = (size_t)blockIdx.x * chunk;
  const size_t mask = ((size_t)1 << s) - 1;
  fe_t acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = fe_zero();
#pragma unroll 1
  for (int k = 0; k < EVAL_PPT; ++k) {
    const size_t id = base + (size_t)k * blockDim.x + threadIdx.x;
    if (id < half) {
      const fe_t a0 = A[id], a1 = A[id + half], b0 = B[id], b1 = B[id + half], c0 = ZC ? fe_zero() : C[id];
      fe_t w = (MODE == 0) ? eq_in[id] : eq_in[id & mask];
      if (MODE == 2) w = fe_mul<S>(w, eq_out[id >> s]);
      const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
      if (!ZC) {
        const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
        acc[0] = fe_add<S>(acc[0], fe_mul<S>(w, t0e));
      }
      acc[1] = fe_add<S>(acc[1], fe_mul<S>(w, tie));
      if (WITH_M1) {
        const fe_t c1 = C[id + half];
        const fe_t ma = fe_sub<S>(fe_dbl<S>(a0), a1), mb = fe_sub<S>(fe_dbl<S>(b0), b1), mc = fe_sub<S>(fe_dbl<S>(c0), c1);
        acc[NACC - 1] = fe_add<S>(acc[NACC - 1], fe_mul<S>(w, fe_sub<S>(fe_mul<S>(ma, mb), mc)));
      }
    }
  }
  block_sum<NACC>(acc, smem);
  if (threadIdx.x == 0) {
    if (MODE == 1) {
      const fe_t eo = eq_out[base >> s];
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = fe_mul<S>(acc[k], eo);
    }
    emit_partials<NACC>(acc, partials, single_out, seq);
  }
}

// Round 1 of the cubic sum-check from precomputed per-pair products (k_spmv3_pairs): t0 = sum E(id) P0[id], t_inf = sum E(id) P1[id].
// Streaming form (lazy wave sums, second stage k_sum_partials_lazy applies eq_out per group): half must be a multiple of 256 and, in factored
// mode, 2^s >= 256.
template <int MODE, int PPT>
__global__ void __launch_bounds__(256) k_eval_products_stream(const fe_t* __restrict__ P0, const fe_t* __restrict__ P1, const fe_t* __restrict__ eq_in, int s,
                                                              LazyOut partials);

// ---- K1+K2 fused: bind round i with challenge r, evaluate round i+1 on the values just produced ------------------------
// Tables have length L = 4q before the bind. Thread id in [0, q) owns the new pair (Z'[id], Z'[id+q]):
//   Z'[x] = Z[x] + r (Z[x + 2q] - Z[x]),  x in {id, id + q}
// reads 4 elements and writes 2 per table (in place, disjoint across threads), i.e. exactly the 48 L bytes/table of an
// unfused bind; the next round's sums come from registers for free.
__device__ __forceinline__ fe_t bind1(const fe_t& lo, const fe_t& hi, const fe_t& r) { return fe_add<S>(lo, fe_mul<S>(r, fe_sub<S>(hi, lo))); }
// One pair per lane (EVAL_PPT == 1). Everything that does not depend on the challenge - the twelve element loads, the weight - is issued BEFORE the
// challenge is taken: a launch issued ahead of its challenge then waits at the mailbox with its operands in registers, and the 2-3 us of memory
// latency are off the round's critical path.
static_assert(EVAL_PPT == 1, "the fused bind + evaluate kernels below hold one pair per lane across the mailbox wait");
template <int MODE>
__global__ void __launch_bounds__(256) k_bind_eval_cubic(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r_arg,
                                                         const fe_t* __restrict__ eq_in, const fe_t* __restrict__ eq_out, int s,
                                                         fe_t* __restrict__ partials, fe_t* __restrict__ single_out, unsigned seq, MailRef mref) {
  __shared__ fe_t smem[2 * 4];
  const size_t base = (size_t)blockIdx.x * blockDim.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const size_t id = base + threadIdx.x;
  const bool live = id < q;
  const size_t ld = live ? id : 0;  // (dead lanes of the last block load element 0: no branch around the loads)
  const fe_t la0 = A[ld], la1 = A[ld + q], la2 = A[ld + 2 * q], la3 = A[ld + 3 * q];
  const fe_t lb0 = B[ld], lb1 = B[ld + q], lb2 = B[ld + 2 * q], lb3 = B[ld + 3 * q];
  const fe_t lc0 = C[ld], lc1 = C[ld + q], lc2 = C[ld + 2 * q], lc3 = C[ld + 3 * q];
  fe_t w = (MODE == 0) ? eq_in[ld] : eq_in[ld & mask];
  if (MODE == 2) w = fe_mul<S>(w, eq_out[ld >> s]);
  const fe_t r = challenge_or(mref, r_arg);
  fe_t acc[2] = {fe_zero(), fe_zero()};
  if (live) {
    const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
    const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
    const fe_t c0 = bind1(lc0, lc2, r), c1 = bind1(lc1, lc3, r);
    A[id] = a0;
    A[id + q] = a1;
    B[id] = b0;
    B[id + q] = b1;
    C[id] = c0;
    C[id + q] = c1;
    const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
    const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
    acc[0] = fe_mul<S>(w, t0e);
    acc[1] = fe_mul<S>(w, tie);
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    if (MODE == 1) {
      const fe_t eo = eq_out[base >> s];
      acc[0] = fe_mul<S>(acc[0], eo);
      acc[1] = fe_mul<S>(acc[1], eo);
    }
    emit_partials<2>(acc, partials, single_out, seq);
  }
}
// dense quadratic variant (both tables fully non-zero)
__global__ void __launch_bounds__(256) k_bind_eval_quad(fe_t* __restrict__ A, fe_t* __restrict__ B, size_t q, fe_t r_arg, fe_t* __restrict__ partials,
                                                        fe_t* __restrict__ single_out, unsigned seq, MailRef mref) {
  __shared__ fe_t smem[2 * 4];
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = id < q;
  const size_t ld = live ? id : 0;
  const fe_t la0 = A[ld], la1 = A[ld + q], la2 = A[ld + 2 * q], la3 = A[ld + 3 * q];
  const fe_t lb0 = B[ld], lb1 = B[ld + q], lb2 = B[ld + 2 * q], lb3 = B[ld + 3 * q];
  const fe_t r = challenge_or(mref, r_arg);
  fe_t acc[2] = {fe_zero(), fe_zero()};
  if (live) {
    const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
    const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
    A[id] = a0;
    A[id + q] = a1;
    B[id] = b0;
    B[id + q] = b1;
    acc[0] = fe_mul<S>(a0, b0);
    acc[1] = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    emit_partials<2>(acc, partials, single_out, seq);
  }
}

// ---- streaming variants (tables beyond the L2s: L*T*32 B >= 48 MiB) -----------------------------------------------------------
// Same data movement as the fused kernels above; the reduction is per WAVE and lazy (9-word sums, no modular adds, no LDS, no
// barrier), and the eq_left factor / modular reduction happen once per x_out group in k_sum_partials_lazy. On a 30 us kernel the
// block-level modular tree was a quarter of the time (tools/fused_bench.hip).
// wave sums -> LDS -> one lazy block partial pair per 256-thread block (plain multiword adds: no modular arithmetic here).
// Layout of the partials (nparts = gridDim.x blocks, two accumulators of nine words): word w of accumulator a of block b at
// P[(2 w + a) nparts + b] - word-major, so that the second stage's lane, which owns a group of consecutive blocks, fetches each word of its whole
// group with ONE vector load and a wave's loads are contiguous (block-major 72-byte records cost the old second stage one dependent, uncoalesced
// load per block and word: 8-10 us for 150 KB).
// The second stage's work for result slot `slot` of `nslots` (k_sum_partials_lazy's body): a lane per group of 2^gl consecutive blocks, the lazy sum of
// the group, ONE reduction mod p, the product with eq_out[group] when given, a modular wave sum, the slot store. AGENT: the partials were written by other
// blocks of the SAME launch on other XCDs - read past this XCD's L2 (agent-scope loads), as the resident tail reads what its neighbours bound.
template <bool AGENT>
__device__ __forceinline__ void lazy_slot_sums(const uint32_t* __restrict__ P, size_t nparts, int gl, const fe_t* __restrict__ eq_out, fe_t* __restrict__ mapped, unsigned seq,
                                               unsigned slot, unsigned nslots, int lane) {
  const size_t ngroups = nparts >> gl, per = (size_t)1 << gl;
  fe_t acc[2] = {fe_zero(), fe_zero()};
  // slot s owns the contiguous groups [s G, (s + 1) G), G = ngroups / nslots (G = 64 k: the second-stage kernel's split when nslots = ngroups / 64)
  const size_t G = ngroups / nslots;
  for (size_t g = (size_t)slot * G + lane; g < (size_t)(slot + 1) * G; g += 64) {
    fe_t f[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      lazy9_t l;
      unsigned long long carry = 0;
#pragma unroll
      for (int w = 0; w < 9; ++w) {
        const uint32_t* src = P + (size_t)(2 * w + a) * nparts + g * per;
        unsigned long long t = carry;
        if (per == 1) {
          t += AGENT ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[0];
        } else {  // (per is a power of two >= 2 and the group's words are 8-byte aligned: pairs of words per load)
          const unsigned long long* s2 = reinterpret_cast<const unsigned long long*>(src);
          for (size_t k = 0; k < per / 2; ++k) {
            const unsigned long long v = AGENT ? __hip_atomic_load(s2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s2[k];
            t += (v & 0xffffffffull) + (v >> 32);
          }
        }
        l.v[w] = (uint32_t)t;
        carry = t >> 32;
      }
      f[a] = lazy_reduce(l);
    }
    if (eq_out) {
      const fe_t eo = eq_out[g];
      f[0] = fe_mul<S>(f[0], eo);
      f[1] = fe_mul<S>(f[1], eo);
    }
    acc[0] = fe_add<S>(acc[0], f[0]);
    acc[1] = fe_add<S>(acc[1], f[1]);
  }
  acc[0] = wave_sum(acc[0]);
  acc[1] = wave_sum(acc[1]);
  if (lane == 0) {
    fe_t* sl = mapped + SLOT_BASE_ELEM + 4 * slot;
    slot_chk chk = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      slot_store_elem(sl + k, acc[k]);
      slot_chk_add(chk, acc[k], k);
    }
    slot_store_tag(sl, seq, chk);
  }
}
__device__ __forceinline__ void stream_block_partials(const lazy9_t& s0, const lazy9_t& s1, LazyOut partials) {
  __shared__ lazy9_t sm[4][2];
  __shared__ unsigned last_sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sm[wave][0] = s0;
    sm[wave][1] = s1;
  }
  __syncthreads();
  const bool fold = partials.tickets != nullptr;
  if (threadIdx.x < 2) {  // thread 0 -> accumulator 0, thread 1 -> accumulator 1
    const lazy9_t t = lazy_add(lazy_add(sm[0][threadIdx.x], sm[1][threadIdx.x]), lazy_add(sm[2][threadIdx.x], sm[3][threadIdx.x]));
    uint32_t* P = reinterpret_cast<uint32_t*>(partials.P);
    if (fold) {
      // written through to where every XCD sees them (agent scope) and ACKNOWLEDGED before this block takes its ticket: no release fence - a release
      // would write back this XCD's whole L2, i.e. wait for every block's table stores
#pragma unroll
      for (int w = 0; w < 9; ++w) __hip_atomic_store(P + (size_t)(2 * w + threadIdx.x) * gridDim.x + blockIdx.x, t.v[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int w = 0; w < 9; ++w) P[(size_t)(2 * w + threadIdx.x) * gridDim.x + blockIdx.x] = t.v[w];
    }
  }
  if (!fold) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned slot = (unsigned)(blockIdx.x / partials.per_slot);  // (contiguous: per_slot blocks = a whole number of groups)
    const unsigned old = __hip_atomic_fetch_add(partials.tickets + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_sh = old + 1 == partials.per_slot ? slot + 1 : 0u;
  }
  __syncthreads();
  if (last_sh == 0 || threadIdx.x >= 64) return;
  const unsigned slot = last_sh - 1;
  if (threadIdx.x == 0) __hip_atomic_store(partials.tickets + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  lazy_slot_sums<true>(reinterpret_cast<const uint32_t*>(partials.P), gridDim.x, partials.gl, partials.eq_out, partials.mapped, partials.seq, slot, partials.nslots, lane);
}
// AHEAD: launched before its challenge is known (waits at the mailbox). A separate instantiation so that the ordinary form keeps its register
// budget (128 VGPRs, 4 waves per SIMD): holding the twelve loaded elements across the mailbox barrier costs 20 more.
// GATED: queued behind k_mail_gate (capi_core.hip gate_launch): the challenge is in device memory when the kernel starts. Its own instantiation too,
// so that the ordinary form stays the code it was.
template <int MODE, bool AHEAD, bool GATED = false>
__global__ void __launch_bounds__(256) k_bind_eval_cubic_stream(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r_arg,
                                                                const fe_t* __restrict__ eq_in, int s, LazyOut partials, MailRef mref) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // q is a multiple of the block size here
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t la0 = A[id], la1 = A[id + q], la2 = A[id + 2 * q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb1 = B[id + q], lb2 = B[id + 2 * q], lb3 = B[id + 3 * q];
  const fe_t lc0 = C[id], lc1 = C[id + q], lc2 = C[id + 2 * q], lc3 = C[id + 3 * q];
  fe_t r = r_arg;
  if (AHEAD) r = challenge_or(mref, r_arg);  // after the loads: a kernel launched ahead fetches its tables while the host draws the challenge
  if (GATED) r = gated_challenge(mref.gated);
  const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
  const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
  const fe_t c0 = bind1(lc0, lc2, r), c1 = bind1(lc1, lc3, r);
  A[id] = a0;
  A[id + q] = a1;
  B[id] = b0;
  B[id + q] = b1;
  C[id] = c0;
  C[id + q] = c1;
  const fe_t w = (MODE == 0) ? eq_in[id] : eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}
// Round 1 of the cubic sum-check straight from the tables, streaming form (one pair per lane, all five element loads and the weight issued before
// the first product, lazy wave sums; k_sum_partials_lazy applies eq_out per group): for tables past the L2s, where k_eval_cubic's modular
// block tree and 8-pair loop leave it at 42-44 % of the HBM rate. half must be a multiple of 256 and, in factored mode, 2^s >= 256.
template <int MODE>
__global__ void __launch_bounds__(256) k_eval_cubic_stream(const fe_t* __restrict__ A, const fe_t* __restrict__ B, const fe_t* __restrict__ C, size_t half,
                                                           const fe_t* __restrict__ eq_in, int s, LazyOut partials) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t a0 = A[id], a1 = A[id + half], b0 = B[id], b1 = B[id + half], c0 = C[id];
  const fe_t w = (MODE == 0) ? eq_in[id] : eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}
// PPT consecutive 256-pair chunks per block (the block stays inside one x_out group: 256 PPT <= 2^s in factored mode), all 2 PPT element loads and the
// PPT weights of a lane issued before the first product, one lazy sum per lane and accumulator and ONE wave sum each. A wave whose P0 entries are all
// zero - every wave when the instance is satisfied: Az o Bz - Cz vanishes on the hypercube - skips that accumulator's products and its wave sum (the
// entries are still read: nothing is assumed about the table).
template <int MODE, int PPT>
__global__ void __launch_bounds__(256) k_eval_products_stream(const fe_t* __restrict__ P0, const fe_t* __restrict__ P1, const fe_t* __restrict__ eq_in, int s,
                                                              LazyOut partials) {
  const size_t id0 = (size_t)blockIdx.x * (256 * PPT) + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  fe_t p0[PPT], p1[PPT], w[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const size_t id = id0 + 256 * (size_t)k;
    p0[k] = P0[id];
    p1[k] = P1[id];
    w[k] = (MODE == 0) ? eq_in[id] : eq_in[id & mask];
  }
  bool nz = false;
#pragma unroll
  for (int k = 0; k < PPT; ++k) nz |= !fe_is_zero(p0[k]);
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
#pragma unroll
  for (int k = 0; k < PPT; ++k) l1 = lazy_add(l1, lazy_from(fe_mul<S>(w[k], p1[k])));
  l1 = lazy_wave_sum(l1);
  if (__any(nz)) {  // wave-uniform
#pragma unroll
    for (int k = 0; k < PPT; ++k) l0 = lazy_add(l0, lazy_from(fe_mul<S>(w[k], p0[k])));
    l0 = lazy_wave_sum(l0);
  }
  stream_block_partials(l0, l1, partials);
}
__global__ void __launch_bounds__(256) k_bind_eval_quad_stream(fe_t* __restrict__ A, fe_t* __restrict__ B, size_t q, fe_t r_arg, LazyOut partials,
                                                               MailRef mref) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const fe_t la0 = A[id], la1 = A[id + q], la2 = A[id + 2 * q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb1 = B[id + q], lb2 = B[id + 2 * q], lb3 = B[id + 3 * q];
  const fe_t r = challenge_or(mref, r_arg);
  const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
  const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
  A[id] = a0;
  A[id + q] = a1;
  B[id] = b0;
  B[id + q] = b1;
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(a0, b0))), lazy_wave_sum(lazy_from(fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0)))), partials);
}
// The same for tables whose high half is zero beyond the first hiA / hiB entries (hi* <= q): the inner sum-check's first bind on z and poly_ABC
// (2M long, non-zero up to M + num_extra; bind_poly_var_top's `hi <= lo` branch, src/polys/multilinear.rs:118-141). The high half is not read at
// all except for those few entries, and the evaluation of the next round still comes from registers.
__global__ void __launch_bounds__(256) k_bind_eval_quad_stream_sparse(fe_t* __restrict__ A, fe_t* __restrict__ B, size_t q, fe_t r_arg, size_t hiA, size_t hiB,
                                                                      LazyOut partials, MailRef mref) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const fe_t la0 = A[id], la1 = A[id + q], lb0 = B[id], lb1 = B[id + q];
  const fe_t r = challenge_or(mref, r_arg);
  const fe_t one_minus_r = fe_sub<S>(fe_one<S>(), r);
  const fe_t a0 = id < hiA ? bind1(la0, A[id + 2 * q], r) : fe_mul<S>(la0, one_minus_r);
  const fe_t b0 = id < hiB ? bind1(lb0, B[id + 2 * q], r) : fe_mul<S>(lb0, one_minus_r);
  const fe_t a1 = fe_mul<S>(la1, one_minus_r), b1 = fe_mul<S>(lb1, one_minus_r);  // id + q >= q >= hi*: partner is zero
  A[id] = a0;
  A[id + q] = a1;
  B[id] = b0;
  B[id + q] = b1;
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(a0, b0))), lazy_wave_sum(lazy_from(fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0)))), partials);
}
// compute_eval_points_quad (src/sumcheck.rs:128-174) in the streaming form: PPT pairs per lane accumulated lazily (9-word sums, no modular
// reduction), one lazy wave sum per wave. k_eval_quad spends more issue slots on its per-wave modular reduction tree than on the two products of
// a pair (50 us for 2^20 pairs, compute-bound); this form is bound by the 64 MB it reads. len must be a multiple of 256 * PPT.
template <int PPT>
__global__ void __launch_bounds__(256) k_eval_quad_stream(const fe_t* __restrict__ A, const fe_t* __restrict__ B, size_t half, size_t hiA, size_t hiB,
                                                          LazyOut partials) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
#pragma unroll
  for (int k = 0; k < PPT; ++k, id += stride) {
    const fe_t a0 = A[id], b0 = B[id];
    const fe_t a1 = id < hiA ? A[id + half] : fe_zero(), b1 = id < hiB ? B[id + half] : fe_zero();
    l0 = lazy_add(l0, lazy_from(fe_mul<S>(a0, b0)));
    l1 = lazy_add(l1, lazy_from(fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0))));
  }
  stream_block_partials(lazy_wave_sum(l0), lazy_wave_sum(l1), partials);
}
// The same sums when the high halves are zero beyond a short prefix (hi = max(hiA, hiB) << half: the inner sum-check's round 0 on z = [W | 1 | X | 0...]
// and poly_ABC, src/spartan.rs:323-384). A pair whose partners are both zero contributes a0 b0 to BOTH sums, so the pass is one product per pair
// - a dot product of the low halves - and only the lanes below hi pay the second product; their term enters as (a1 - a0)(b1 - b0) - a0 b0 and the
// dot is added to the second accumulator once at the end. All 2 PPT element loads of a lane are issued before the first product.
template <int PPT>
__global__ void __launch_bounds__(256) k_eval_quad_stream_lowhi(const fe_t* __restrict__ A, const fe_t* __restrict__ B, size_t half, size_t hiA, size_t hiB,
                                                                LazyOut partials) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t id0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe_t a0[PPT], b0[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    a0[k] = A[id0 + k * stride];
    b0[k] = B[id0 + k * stride];
  }
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const size_t id = id0 + k * stride;
    const fe_t p = fe_mul<S>(a0[k], b0[k]);
    l0 = lazy_add(l0, lazy_from(p));
    if (id < hiA || id < hiB) {
      const fe_t a1 = id < hiA ? A[id + half] : fe_zero(), b1 = id < hiB ? B[id + half] : fe_zero();
      l1 = lazy_add(l1, lazy_from(fe_sub<S>(fe_mul<S>(fe_sub<S>(a1, a0[k]), fe_sub<S>(b1, b0[k])), p)));
    }
  }
  l1 = lazy_add(l1, l0);
  stream_block_partials(lazy_wave_sum(l0), lazy_wave_sum(l1), partials);
}
// Second stage for the streaming kernels: per group of 2^GL consecutive blocks the lazy sum, ONE reduction mod p and the product with eq_out[group]
// (when given), then a modular wave sum. Single-wave blocks, a group per lane, at most HOST_SUM_MAX_BLOCKS of them: every block lands its two sums
// in a host result slot of its own and the host adds the 4-8 slots (no LDS, no barrier, no cross-wave stage; the loads of a lane are 18 vector
// loads issued together). GL < 0: group size 2^gl_rt from the argument (scalar loads; shapes the templates do not cover).
template <int GL>
__global__ void __launch_bounds__(64) k_sum_partials_lazy(const uint32_t* __restrict__ P, size_t nparts, int gl_rt, const fe_t* __restrict__ eq_out,
                                                          fe_t* __restrict__ mapped, unsigned seq) {
  const int gl = GL >= 0 ? GL : gl_rt;
  const size_t ngroups = nparts >> gl, per = (size_t)1 << gl;
  fe_t acc[2] = {fe_zero(), fe_zero()};
  for (size_t g = (size_t)blockIdx.x * 64 + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * 64) {
    unsigned long long col[2][9];
#pragma unroll
    for (int w = 0; w < 9; ++w)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const uint32_t* src = P + (size_t)(2 * w + a) * nparts + g * per;
        if constexpr (GL == 0) {
          col[a][w] = src[0];
        } else if constexpr (GL == 1) {
          const uint2 v = *reinterpret_cast<const uint2*>(src);
          col[a][w] = (unsigned long long)v.x + v.y;
        } else if constexpr (GL == 2) {
          const uint4 v = *reinterpret_cast<const uint4*>(src);
          col[a][w] = ((unsigned long long)v.x + v.y) + ((unsigned long long)v.z + v.w);
        } else if constexpr (GL == 3) {
          const uint4 v = *reinterpret_cast<const uint4*>(src), u = *reinterpret_cast<const uint4*>(src + 4);
          col[a][w] = (((unsigned long long)v.x + v.y) + ((unsigned long long)v.z + v.w)) + (((unsigned long long)u.x + u.y) + ((unsigned long long)u.z + u.w));
        } else {
          unsigned long long t = 0;
          for (size_t k = 0; k < per; ++k) t += src[k];
          col[a][w] = t;
        }
      }
    fe_t f[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {  // column sums -> nine words (a block's sum is below 2^264, a group's below 2^288 for up to 2^24 blocks)
      lazy9_t l;
      unsigned long long carry = 0;
#pragma unroll
      for (int w = 0; w < 9; ++w) {
        const unsigned long long t = col[a][w] + carry;
        l.v[w] = (uint32_t)t;
        carry = t >> 32;
      }
      f[a] = lazy_reduce(l);
    }
    if (eq_out) {
      const fe_t eo = eq_out[g];
      f[0] = fe_mul<S>(f[0], eo);
      f[1] = fe_mul<S>(f[1], eo);
    }
    acc[0] = fe_add<S>(acc[0], f[0]);
    acc[1] = fe_add<S>(acc[1], f[1]);
  }
  acc[0] = wave_sum(acc[0]);
  acc[1] = wave_sum(acc[1]);
  if (threadIdx.x == 0) emit_partials<2>(acc, nullptr, mapped, seq);  // gridDim.x <= HOST_SUM_MAX_BLOCKS: the slot path
}

// ---- K3: quadratic evaluation sums ---------------------------------------------------------------------------------
//   eval0 = sum_{i < len} A0 B0 ; tinf = sum_{i < len} (A1 - A0)(B1 - B0), len = min(eff_pairs(A), eff_pairs(B), half)
// hiA / hiB = eff_hi of the tables: the high half is zero from there on and is not read (the inner sum-check's round 0 runs on 2M-long tables whose
// high halves hold num_extra entries: half the traffic)
__global__ void __launch_bounds__(256) k_eval_quad(const fe_t* __restrict__ A, const fe_t* __restrict__ B, size_t half, size_t len, size_t hiA, size_t hiB,
                                                   fe_t* __restrict__ partials, fe_t* __restrict__ single_out, unsigned seq) {
  __shared__ fe_t smem[2 * 4];
  const size_t chunk = (size_t)blockDim.x * EVAL_PPT;
  const size_t base = (size_t)blockIdx.x * chunk;
  fe_t acc[2] = {fe_zero(), fe_zero()};
#pragma unroll 1
  for (int k = 0; k < EVAL_PPT; ++k) {
    const size_t id = base + (size_t)k * blockDim.x + threadIdx.x;
    if (id < len) {
      const fe_t a0 = A[id], b0 = B[id];
      const fe_t a1 = id < hiA ? A[id + half] : fe_zero(), b1 = id < hiB ? B[id + half] : fe_zero();
      acc[0] = fe_add<S>(acc[0], fe_mul<S>(a0, b0));
      acc[1] = fe_add<S>(acc[1], fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0)));
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    emit_partials<2>(acc, partials, single_out, seq);
  }
}

// ---- K4: NeutronNova outer rounds: sums at 0, 2, 3 of pow_tau * (A B - C) with pow_tau = left (x) right ---------------------------
// compute_eval_points_cubic_with_additive_term_with_outer_pow (src/sumcheck.rs:366-498): pair low = i + j*left, weight
// left[i] * right[j] (low end) / left[i] * right[j + right] (high end); FALLBACK (len < left, :262-342): the pow table itself is
// a fourth bound table, weight ends left[low] / left[low + len]. Per-pair weights (two extra products per pair vs the
// reference's per-i factoring; same values).
template <bool FALLBACK>
__global__ void __launch_bounds__(256) k_eval_cubic_outer_pow(const fe_t* __restrict__ pleft, size_t left, const fe_t* __restrict__ pright, size_t right,
                                                              const fe_t* __restrict__ A, const fe_t* __restrict__ B, const fe_t* __restrict__ C, size_t len,
                                                              fe_t* __restrict__ partials, fe_t* __restrict__ single_out, unsigned seq) {
  __shared__ fe_t smem[3 * 4];
  fe_t acc[3] = {fe_zero(), fe_zero(), fe_zero()};
  const size_t low = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (low < len) {
    fe_t tl, th;
    if (FALLBACK) {
      tl = pleft[low];
      th = pleft[low + len];
    } else {
      const size_t i = low % left, j = low / left;
      const fe_t pl = pleft[i];
      tl = fe_mul<S>(pl, pright[j]);
      th = fe_mul<S>(pl, pright[j + right]);
    }
    const fe_t al = A[low], ah = A[low + len], bl = B[low], bh = B[low + len], cl = C[low], ch = C[low + len];
    acc[0] = fe_mul<S>(tl, fe_sub<S>(fe_mul<S>(al, bl), cl));
    fe_t tb = fe_sub<S>(fe_dbl<S>(th), tl), ab = fe_sub<S>(fe_dbl<S>(ah), al), bb = fe_sub<S>(fe_dbl<S>(bh), bl), cb = fe_sub<S>(fe_dbl<S>(ch), cl);
    acc[1] = fe_mul<S>(tb, fe_sub<S>(fe_mul<S>(ab, bb), cb));
    tb = fe_sub<S>(fe_add<S>(tb, th), tl);
    ab = fe_sub<S>(fe_add<S>(ab, ah), al);
    bb = fe_sub<S>(fe_add<S>(bb, bh), bl);
    cb = fe_sub<S>(fe_add<S>(cb, ch), cl);
    acc[2] = fe_mul<S>(tb, fe_sub<S>(fe_mul<S>(ab, bb), cb));
  }
  block_sum<3>(acc, smem);
  if (threadIdx.x == 0) {
    emit_partials<3>(acc, partials, single_out, seq);
  }
}

// The two instances (step, core) of a batched NeutronNova round in ONE launch: blocks [0, nb) evaluate instance 0, blocks [nb, 2 nb) instance 1, `pp`
// pairs per thread so that 2 nb <= HOST_SUM_MAX_BLOCKS and every block lands its sums in a host slot of its own (the host adds the two groups
// separately). One launch and one wait per round instead of two of each: the rounds of the batched sum-checks are launch / hand-off latency.
struct CubicPairArgs {
  const fe_t *A[2], *B[2], *C[2];
};
template <bool FALLBACK>
__global__ void __launch_bounds__(256) k_eval_cubic_outer_pow_pair(const fe_t* __restrict__ pleft, size_t left, const fe_t* __restrict__ pright, size_t right,
                                                                   CubicPairArgs t, size_t len, unsigned nb, unsigned pp, fe_t* __restrict__ mapped, unsigned seq) {
  __shared__ fe_t smem[3 * 4];
  const unsigned inst = blockIdx.x / nb, bx = blockIdx.x % nb;
  const fe_t* __restrict__ A = t.A[inst];
  const fe_t* __restrict__ B = t.B[inst];
  const fe_t* __restrict__ C = t.C[inst];
  fe_t acc[3] = {fe_zero(), fe_zero(), fe_zero()};
  for (unsigned k = 0; k < pp; ++k) {
    const size_t low = ((size_t)bx * pp + k) * blockDim.x + threadIdx.x;
    if (low >= len) break;
    fe_t tl, th;
    if (FALLBACK) {
      tl = pleft[low];
      th = pleft[low + len];
    } else {
      const size_t i = low % left, j = low / left;
      const fe_t pl = pleft[i];
      tl = fe_mul<S>(pl, pright[j]);
      th = fe_mul<S>(pl, pright[j + right]);
    }
    const fe_t al = A[low], ah = A[low + len], bl = B[low], bh = B[low + len], cl = C[low], ch = C[low + len];
    acc[0] = fe_add<S>(acc[0], fe_mul<S>(tl, fe_sub<S>(fe_mul<S>(al, bl), cl)));
    fe_t tb = fe_sub<S>(fe_dbl<S>(th), tl), ab = fe_sub<S>(fe_dbl<S>(ah), al), bb = fe_sub<S>(fe_dbl<S>(bh), bl), cb = fe_sub<S>(fe_dbl<S>(ch), cl);
    acc[1] = fe_add<S>(acc[1], fe_mul<S>(tb, fe_sub<S>(fe_mul<S>(ab, bb), cb)));
    tb = fe_sub<S>(fe_add<S>(tb, th), tl);
    ab = fe_sub<S>(fe_add<S>(ab, ah), al);
    bb = fe_sub<S>(fe_add<S>(bb, bh), bl);
    cb = fe_sub<S>(fe_add<S>(cb, ch), cl);
    acc[2] = fe_add<S>(acc[2], fe_mul<S>(tb, fe_sub<S>(fe_mul<S>(ab, bb), cb)));
  }
  block_sum<3>(acc, smem);
  if (threadIdx.x == 0) emit_partials<3>(acc, nullptr, mapped, seq);  // gridDim.x <= HOST_SUM_MAX_BLOCKS by construction: the slot path
}
struct QuadPairArgs {
  const fe_t *A[2], *B[2];
  size_t len[2], hiA[2], hiB[2];
};
__global__ void __launch_bounds__(256) k_eval_quad_pair(QuadPairArgs t, size_t half, unsigned nb, unsigned pp, fe_t* __restrict__ mapped, unsigned seq) {
  __shared__ fe_t smem[2 * 4];
  const unsigned inst = blockIdx.x / nb, bx = blockIdx.x % nb;
  const fe_t* __restrict__ A = t.A[inst];
  const fe_t* __restrict__ B = t.B[inst];
  const size_t len = t.len[inst], hiA = t.hiA[inst], hiB = t.hiB[inst];
  fe_t acc[2] = {fe_zero(), fe_zero()};
  for (unsigned k = 0; k < pp; ++k) {
    const size_t id = ((size_t)bx * pp + k) * blockDim.x + threadIdx.x;
    if (id >= len) break;
    const fe_t a0 = A[id], b0 = B[id];
    const fe_t a1 = id < hiA ? A[id + half] : fe_zero(), b1 = id < hiB ? B[id + half] : fe_zero();
    acc[0] = fe_add<S>(acc[0], fe_mul<S>(a0, b0));
    acc[1] = fe_add<S>(acc[1], fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0)));
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) emit_partials<2>(acc, nullptr, mapped, seq);
}

// ---- batched NeutronNova rounds on SMALL tables: the previous round's bind and this round's evaluation in one launch, one product per lane ----------
// From 2^13 elements down a batched round is pure latency: bind launch (6.7 us) -> evaluation launch (17 us: every lane runs the two weight products,
// three evaluation points of two products each and the modular block tree one after the other) -> wait, ~25 us whatever the size. Here a block owns
// <= 64 pairs of one instance: phase A gives every lane ONE bind (or one weight product) - 8 kinds x pairs items, the bound elements go to memory and
// to LDS -, phase B gives every lane ONE evaluation point of one pair (two dependent products; wave w = point w, so a wave sum finishes it), and
// each block's three sums land in a host slot of their own. Dependent chain: 1 + 2 products instead of ~15.
// q = pairs of the round being evaluated = a quarter of the tables' length before the bind; new element x comes from old x and x + 2q.
constexpr int SMALL_PAIR_PPB = 64;  // pairs per block
struct CubicPairBindArgs {
  fe_t *A[2], *B[2], *C[2];
};
// `chunks` (round 6): a block takes that many consecutive groups of 64 pairs one after the other, its lanes carrying their point's running sum - so the
// launch stays within the 64 result slots up to q = 2^14 pairs (tables of 2^16 elements: the first rounds of config 3's batched sum-checks, which used to
// be a bind launch, an evaluation launch and a second stage each).
template <bool FALLBACK>
__global__ void __launch_bounds__(256) k_bind_eval_cubic_pow_pair_small(const fe_t* __restrict__ pleft, size_t left, const fe_t* __restrict__ pright, size_t right,
                                                                        CubicPairBindArgs t, unsigned q, unsigned nb, unsigned chunks, fe_t r_arg, MailRef mref,
                                                                        fe_t* __restrict__ mapped, unsigned seq) {
  __shared__ fe_t sh[8][SMALL_PAIR_PPB];  // bound A lo/hi, B lo/hi, C lo/hi, weight lo/hi
  __shared__ fe_t sums[3];
  const fe_t r = challenge_or(mref, r_arg);  // (queued ahead of the caller's round hook: the challenge comes through the mailbox)
  const unsigned inst = blockIdx.x / nb, bx = blockIdx.x % nb;
  const unsigned pt = threadIdx.x >> 6, p = threadIdx.x & 63u;
  fe_t vsum = fe_zero();
  for (unsigned ch = 0; ch < chunks; ++ch) {
    const unsigned base = (bx * chunks + ch) * SMALL_PAIR_PPB;
    if (base >= q) break;  // (block-uniform)
    const unsigned np = q - base < (unsigned)SMALL_PAIR_PPB ? q - base : (unsigned)SMALL_PAIR_PPB;
    if (ch) __syncthreads();  // the previous group's points have been read
    for (unsigned i = threadIdx.x; i < 8 * (unsigned)SMALL_PAIR_PPB; i += blockDim.x) {
      const unsigned kind = i / SMALL_PAIR_PPB, pp = i % SMALL_PAIR_PPB;  // kind is wave-uniform
      if (pp >= np) continue;
      const unsigned low = base + pp;
      fe_t v;
      if (kind < 6) {
        fe_t* __restrict__ T = kind < 2 ? t.A[inst] : kind < 4 ? t.B[inst] : t.C[inst];
        const size_t idx = (size_t)low + (kind & 1u) * (size_t)q;
        v = bind1(T[idx], T[idx + 2 * (size_t)q], r);
        T[idx] = v;
      } else if (FALLBACK) {
        v = pleft[low + (kind & 1u) * q];
      } else {
        v = fe_mul<S>(pleft[low % left], pright[low / left + (kind & 1u) * right]);
      }
      sh[kind][pp] = v;
    }
    __syncthreads();
    if (pt < 3 && p < np) {
      fe_t a = sh[0][p], b = sh[2][p], c = sh[4][p], w = sh[6][p];
      if (pt) {  // the point 2 (3): 2 hi - lo (3 hi - 2 lo)
        const fe_t ah = sh[1][p], bh = sh[3][p], chh = sh[5][p], wh = sh[7][p];
        fe_t a2 = fe_sub<S>(fe_dbl<S>(ah), a), b2 = fe_sub<S>(fe_dbl<S>(bh), b), c2 = fe_sub<S>(fe_dbl<S>(chh), c), w2 = fe_sub<S>(fe_dbl<S>(wh), w);
        if (pt == 2) {
          a2 = fe_sub<S>(fe_add<S>(a2, ah), a);
          b2 = fe_sub<S>(fe_add<S>(b2, bh), b);
          c2 = fe_sub<S>(fe_add<S>(c2, chh), c);
          w2 = fe_sub<S>(fe_add<S>(w2, wh), w);
        }
        a = a2;
        b = b2;
        c = c2;
        w = w2;
      }
      vsum = fe_add<S>(vsum, fe_mul<S>(w, fe_sub<S>(fe_mul<S>(a, b), c)));
    }
  }
  if (pt < 3) {
    const fe_t v = wave_sum(vsum);
    if (p == 0) sums[pt] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const fe_t acc[3] = {sums[0], sums[1], sums[2]};
    if (gridDim.x <= HOST_SUM_MAX_BLOCKS) emit_partials<3>(acc, nullptr, mapped, seq);  // the slot path
    else emit_partials_wide<3>(acc, mapped, seq);  // q = 2^12, 2^13 pairs: 128 / 256 blocks, the wide slot area (gridDim.x <= WIDE_SLOTS by construction)
  }
}
struct QuadPairBindArgs {
  fe_t *A[2], *B[2];
};
__global__ void __launch_bounds__(256) k_bind_eval_quad_pair_small(QuadPairBindArgs t, unsigned q, unsigned nb, unsigned chunks, fe_t r_arg, MailRef mref,
                                                                   fe_t* __restrict__ mapped, unsigned seq) {
  __shared__ fe_t sh[4][SMALL_PAIR_PPB];  // bound A lo/hi, B lo/hi
  __shared__ fe_t sums[2];
  const fe_t r = challenge_or(mref, r_arg);
  const unsigned inst = blockIdx.x / nb, bx = blockIdx.x % nb;
  const unsigned pt = threadIdx.x >> 6, p = threadIdx.x & 63u;
  fe_t vsum = fe_zero();
  for (unsigned ch = 0; ch < chunks; ++ch) {  // (see k_bind_eval_cubic_pow_pair_small)
    const unsigned base = (bx * chunks + ch) * SMALL_PAIR_PPB;
    if (base >= q) break;
    const unsigned np = q - base < (unsigned)SMALL_PAIR_PPB ? q - base : (unsigned)SMALL_PAIR_PPB;
    if (ch) __syncthreads();
    {
      const unsigned kind = threadIdx.x / SMALL_PAIR_PPB, pp = threadIdx.x % SMALL_PAIR_PPB;  // 4 kinds x 64 pairs = the block
      if (pp < np) {
        fe_t* __restrict__ T = kind < 2 ? t.A[inst] : t.B[inst];
        const size_t idx = (size_t)(base + pp) + (kind & 1u) * (size_t)q;
        const fe_t v = bind1(T[idx], T[idx + 2 * (size_t)q], r);
        T[idx] = v;
        sh[kind][pp] = v;
      }
    }
    __syncthreads();
    if (pt < 2 && p < np) {
      const fe_t a0 = sh[0][p], b0 = sh[2][p];
      vsum = fe_add<S>(vsum, pt == 0 ? fe_mul<S>(a0, b0) : fe_mul<S>(fe_sub<S>(sh[1][p], a0), fe_sub<S>(sh[3][p], b0)));
    }
  }
  if (pt < 2) {
    const fe_t v = wave_sum(vsum);
    if (p == 0) sums[pt] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const fe_t acc[2] = {sums[0], sums[1]};
    if (gridDim.x <= HOST_SUM_MAX_BLOCKS) emit_partials<2>(acc, nullptr, mapped, seq);
    else emit_partials_wide<2>(acc, mapped, seq);
  }
}

// dot product of the first n elements (value of DelayedReduction::reduce(sum a_i b_i))
__global__ void __launch_bounds__(256) k_dot(const fe_t* __restrict__ A, const fe_t* __restrict__ B, size_t n, fe_t* __restrict__ partials,
                                             fe_t* __restrict__ single_out, unsigned seq) {
  __shared__ fe_t smem[4];
  fe_t acc[1] = {fe_zero()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc[0] = fe_add<S>(acc[0], fe_mul<S>(A[i], B[i]));
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) {
    emit_partials<1>(acc, partials, single_out, seq);
  }
}

// Second stage of the mid-size rounds: sum_b partials[b * nacc + k], nacc <= 3, over more than HOST_SUM_MAX_BLOCKS producer blocks. Single-wave
// blocks (at most HOST_SUM_MAX_BLOCKS), lazy multiword adds through the wave shuffles, one modular reduction per sum, then the block's result
// slot: the host adds the 2-8 slots. This stage sits on the critical path of every mid-size round; its one-block form with an LDS stage and a
// serial tail took 5-6 us.
__global__ void __launch_bounds__(64) k_sum_partials(const fe_t* __restrict__ partials, size_t nblocks, int nacc, fe_t* __restrict__ mapped, unsigned seq) {
  lazy9_t t[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = lazy_from(fe_zero());
  for (size_t b = (size_t)blockIdx.x * 64 + threadIdx.x; b < nblocks; b += (size_t)gridDim.x * 64) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < nacc) t[k] = lazy_add(t[k], lazy_from(partials[b * nacc + k]));
  }
  fe_t acc[3] = {fe_zero(), fe_zero(), fe_zero()};
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < nacc) acc[k] = lazy_reduce(lazy_wave_sum(t[k]));
  if (threadIdx.x == 0) emit_partials<3>(acc, nullptr, mapped, seq);  // gridDim.x <= HOST_SUM_MAX_BLOCKS: the slot path (the host reads nacc of the three)
}

// ---- persistent tail of a sum-check: the last log2(len) rounds in ONE single-block launch ---------------------------------------------
// Below ~2^11 elements a round is pure latency: kernel launch + second-stage launch + host wake-up cost more than the arithmetic. This kernel
// stays resident for all remaining rounds: per round it binds with the challenge, evaluates the next round from registers, publishes the
// sums to mapped pinned host memory (publish_result) and then POLLS a mapped host slot for the next challenge, which the host writes after
// its transcript step. Cubic mode also delivers the third sum t(-1) (fallback_three_inputs, src/sumcheck.rs:1327-1396): the host uses it only
// when tau * p is not invertible, otherwise it derives the evaluations from the claim exactly as the reference does (:1276-1324).
// Slots in the mapped buffer: word 0 of TAIL_ERR_ELEM = error, TAIL_FINAL_ELEM = the final claims; the challenges arrive through the mailbox ring.
constexpr int TAIL_THREADS = 1024;
constexpr unsigned long long TAIL_WIDE_Q_CUBIC = 128;  // the cubic rounds bind three tables and weight every product: half the pairs per block keep its bind phase to one pass
constexpr unsigned long long TAIL_WIDE_Q = 256;  // pairs per resident block and round: every product gets its own lane (3 * 256 <= TAIL_THREADS)
// wave sum when only the first `active` lanes of every wave hold a term (the others hold zero): the levels above `active` add nothing
__device__ __forceinline__ fe_t wave_sum_low(fe_t a, unsigned active) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
    if ((unsigned)m < active) a = fe_add<S>(a, shfl_xor_fe(a, m));
  return a;
}
// -DSP_TAIL_TRACE: thread 0 of block 0 stamps the 100 MHz wall clock at the stations of a one-round step into the mapped buffer (elements 32..47,
// eight stamps per round, ring of 8 rounds); capi_core.hip prints them under SPARTAN_ROUND_TRACE. Diagnostics only.
#ifdef SP_TAIL_TRACE
#define SP_TT(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(a.mapped + 32)[8 * (rnd & 7) + (i)] = wall_clock64(); } while (0)
#else
#define SP_TT(i) do { } while (0)
#endif
struct TailArgs {
  fe_t *A, *B, *C;          // C unused in quadratic mode
  unsigned long long len;   // table length at entry, power of two, 2 <= len <= 4 * TAIL_WIDE_Q * gridDim.x
  fe_t r0;                  // challenge of the round whose sums were produced before the launch
  // cubic: the split-eq tables of EqSumCheckInstance (src/sumcheck.rs:956-1016): pyramids over taus[1..first_half) and taus[first_half..ell);
  // rnd0 = 1-based index of the first round this kernel EVALUATES
  const fe_t *eq_pl, *eq_pr;
  int ell, first_half, rnd0;
  const unsigned* mail;     // the challenge mailbox ring (see mail_wait)
  const unsigned* mirror;   // its host-memory mirror, or nullptr
  int r0_from_mail;         // launched ahead of r0: the first round waits for the mailbox too
  fe_t* mapped;             // device address of the mapped pinned buffer (per-block result slots at SLOT_BASE_ELEM, error word)
  unsigned seq0;            // sequence number of the first result this kernel publishes
  unsigned hand_n;          // tables of <= hand_n entries go to the host (tail_hand_over)
};
__device__ __forceinline__ fe_t load_agent(const fe_t* p) {
  fe_t v;
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned long long t = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.v[2 * i] = (unsigned)t;
    v.v[2 * i + 1] = (unsigned)(t >> 32);
  }
  return v;
}
// One resident block per TAIL_WIDE_Q pairs. A block owns the pairs [b * TAIL_WIDE_Q, ...) of every round (its range only shrinks, so a block whose
// range is empty leaves for good); pairs are block-local, so there is no barrier between blocks — the ordering between rounds comes from the
// host, which posts the next challenge only after every active block has published its slot (each behind an agent-scope release of its table
// writes), and from reading other blocks' elements with agent-scope loads.
template <bool CUBIC>
__global__ void __launch_bounds__(TAIL_THREADS) k_sumcheck_tail(TailArgs a) {
  constexpr int NACC = CUBIC ? 3 : 2;
  constexpr unsigned long long WQ = CUBIC ? TAIL_WIDE_Q_CUBIC : TAIL_WIDE_Q;
  __shared__ fe_t smem[16];
  __shared__ fe_t r_sh;
  __shared__ slot_chk chk_sh[TAIL_HAND_OVER_MAX_VALS];
  // the elements a step has just bound, for its own evaluation (a block evaluates exactly the pairs it bound): table t at [t * 2 qb, ..), the low
  // elements x = base + e first, then their partners x = q + base + e. They go to memory as well - the next step's bind reads them from there.
  // (cubic: room for a first step of 2 WQ pairs per block - a launch over twice the pairs its 64 blocks take at WQ each, see tail_blocks in capi_core.hip)
  __shared__ fe_t bound[(CUBIC ? 3 * 2 : 2) * 2 * (unsigned)WQ];
  const unsigned long long base = (unsigned long long)blockIdx.x * WQ;
  // LOCAL REGIME (while more than one block is active, q > WQ): every block of the launch stays active and owns the pairs whose index y has
  // (y / 4) mod gridDim.x == blockIdx.x - local pair e <-> y = ((e / 4) gridDim.x + blockIdx.x) 4 + e % 4 (128-byte groups). The round's pairings are at
  // offsets q and 2q, multiples of 4 gridDim.x as long as a block holds m = q / gridDim.x >= 4 pairs (it does down to q = 2 WQ: m = 2 WQ^2 / q0 >= 4), so a
  // block's elements form a sub-table of their own that is bound and evaluated like the whole - from LDS: after the first step no block reads an element
  // that another one wrote, there is no table traffic, no release and no wait for stores between rounds. The last local step (q = 2 WQ) writes the bound
  // tables back in their natural layout behind a release; block 0 then owns everything, as before. (Contiguous ranges whose owners halved every round made
  // half of every bind's inputs another XCD's: 2.2 us of agent-scope loads at the head of each step, 1.2 us of store acknowledgement + release at its end.)
  const unsigned nb0 = gridDim.x;
  auto local_y = [&](unsigned e) -> unsigned long long { return ((unsigned long long)((e >> 2) * nb0 + blockIdx.x) << 2) | (e & 3u); };
  unsigned long long len = a.len;
  unsigned seq = a.seq0;
  int rnd = a.rnd0;
  // the challenge in hand lives in r_sh (LDS), not in eight registers carried around the loop: a.r0 at first, then whatever the last mail_wait delivered;
  // the binds read it where they use it (the 128-register budget of a 1024-thread block spilled it, tools/spill_report.py)
  if (threadIdx.x == 0) r_sh = a.r0;
  __syncthreads();
  bool first = true, pending2 = false;
  fe_t* slot = a.mapped + SLOT_BASE_ELEM + 4 * blockIdx.x;
  // E(id) of round `rd` (select_eq in capi_core.hip; src/sumcheck.rs:1041-1147)
  auto weight = [&](int rd, unsigned long long id) -> fe_t {
    if (rd < a.first_half) {
      const int s2 = a.ell - a.first_half;
      return fe_mul<S>(a.eq_pr[eq_level_offset(s2) + (id & ((1ull << s2) - 1))], a.eq_pl[eq_level_offset(a.first_half - rd) + (id >> s2)]);
    }
    return a.eq_pr[eq_level_offset(a.ell - rd) + id];
  };
  while (true) {
    bool have_r = false;
    if (pending2) {  // the previous result carried two rounds: two challenges answer it (seq - 2, seq - 1) and the table is bound twice
      if (blockIdx.x != 0) return;
      if (!mail_wait(a.mail, a.mirror, a.mapped, seq - 2, &r_sh)) return;
      const fe_t ra = r_sh;
      __syncthreads();
      if (!mail_wait(a.mail, a.mirror, a.mapped, seq - 1, &r_sh)) return;
      if (len == 4) {  // both rounds were the last two: the final claims
        if (threadIdx.x == 0) {
          const fe_t r = r_sh;
          fe_t* fin = a.mapped + TAIL_FINAL_ELEM;
          const fe_t fa = bind1(bind1(a.A[0], a.A[2], ra), bind1(a.A[1], a.A[3], ra), r);
          const fe_t fb = bind1(bind1(a.B[0], a.B[2], ra), bind1(a.B[1], a.B[3], ra), r);
          a.A[0] = fa;
          a.B[0] = fb;
          slot_store_elem(fin, fa);
          slot_store_elem(fin + 1, fb);
          slot_chk chk = {0u, 0u};
          slot_chk_add(chk, fa, 0);
          slot_chk_add(chk, fb, 1);
          if (CUBIC) {
            const fe_t fc = bind1(bind1(a.C[0], a.C[2], ra), bind1(a.C[1], a.C[3], ra), r);
            a.C[0] = fc;
            slot_store_elem(fin + 2, fc);
            slot_chk_add(chk, fc, 2);
          }
          slot_store_tag(fin, seq - 1, chk);
        }
        return;
      }
      const unsigned h = (unsigned)(len / 2), ntab = CUBIC ? 3u : 2u;  // first of the two binds (one block owns everything here)
      for (unsigned idx = threadIdx.x; idx < ntab * h; idx += TAIL_THREADS) {
        fe_t* Z = idx / h == 0 ? a.A : (idx / h == 1 ? a.B : a.C);
        const unsigned x = idx % h;
        Z[x] = bind1(Z[x], Z[x + h], ra);
      }
      __syncthreads();
      len /= 2;
      pending2 = false;
      have_r = true;  // the ordinary step below binds with the second challenge
    }
    const unsigned long long q = len / 4;
    const bool local = nb0 > 1 && q > WQ;  // (q > WQ holds exactly while the launch is in its multi-block rounds: gridDim.x = q0 / WQ)
    if (!local && (len > 2 ? base >= q : blockIdx.x != 0)) return;
    // the weight of this lane's product does not depend on the challenge: fetched (and, in the first-half rounds, multiplied together) before the wait
    fe_t w_pre = fe_zero();
    if (CUBIC && len > 2) {
      const unsigned qb_p = local ? (unsigned)(q / nb0) : (unsigned)(q - base < WQ ? q - base : WQ);
      if (tail_double(CUBIC, len / 2, a.hand_n)) {
        const unsigned qd_p = (unsigned)(len / 8), seg_p = qd_p < 64 ? 64u : qd_p, g_p = threadIdx.x / seg_p, i_p = threadIdx.x % seg_p;
        if (g_p < 9 && i_p < qd_p) w_pre = weight(rnd + 1, i_p);
        else if (g_p < 15 && i_p < qd_p) w_pre = weight(rnd, i_p + ((g_p - 9) & 1u) * qd_p);
      } else {
        const unsigned seg_p = qb_p < 64 ? 64u : qb_p, wh_p = threadIdx.x / seg_p, i_p = threadIdx.x % seg_p;
        if (wh_p < (unsigned)NACC && i_p < qb_p) w_pre = weight(rnd, local ? local_y(i_p) : base + i_p);
      }
    }
    SP_TT(0);
    if (!have_r && (!first || a.r0_from_mail)) {
      if (!mail_wait(a.mail, a.mirror, a.mapped, seq - 1, &r_sh)) return;
    }
    SP_TT(1);
    const fe_t r = r_sh;
    // the previous round had 2q pairs: while that is more than one block's worth, the elements bound below were written by other blocks
    // (other XCDs, other L2s). They are read with agent-scope loads, which go past this XCD's L2, instead of an acquire fence, which would
    // invalidate it (measured: 4 us per round).
    const bool foreign = !local && 2 * q > WQ;
    const bool from_lds = local && !first;  // a local step after the first: its inputs are the elements this block bound in the step before, in `bound`
    first = false;
    if (len == 2) {  // last round: bind only; the final claims also go to the host in a slot of their own (saves three synchronous reads)
      if (threadIdx.x == 0) {
        fe_t* fin = a.mapped + TAIL_FINAL_ELEM;
        const fe_t fa = bind1(a.A[0], a.A[1], r), fb = bind1(a.B[0], a.B[1], r);
        a.A[0] = fa;
        a.B[0] = fb;
        slot_store_elem(fin, fa);
        slot_store_elem(fin + 1, fb);
        slot_chk chk = {0u, 0u};
        slot_chk_add(chk, fa, 0);
        slot_chk_add(chk, fb, 1);
        if (CUBIC) {
          const fe_t fc = bind1(a.C[0], a.C[1], r);
          a.C[0] = fc;
          slot_store_elem(fin + 2, fc);
          slot_chk_add(chk, fc, 2);
        }
        slot_store_tag(fin, seq - 1, chk);  // tagged with the result number the last challenge answered
      }
      return;
    }
    const unsigned qb = local ? (unsigned)(q / nb0) : (unsigned)(q - base < WQ ? q - base : WQ);  // pairs of this block (a power of two)
    const bool last_local = local && q == 2 * WQ;  // the next step is block 0's alone: the bound tables go back to memory
    // phase A: one bind per lane. New element x takes old x and x + 2q; this block owns x in [base, base + qb) and [q + base, q + base + qb) - in the
    // local regime the x of its sub-table (local_y), whose element e of the step before sits at bound[t * 4 qb + e].
    const unsigned nt = CUBIC ? 3 : 2;
    if (local && !from_lds) {  // the launch's first step, from the tables in memory (a share of 2 WQ pairs takes two passes over the lanes)
      for (unsigned idx = threadIdx.x; idx < nt * 2 * qb; idx += TAIL_THREADS) {
        const unsigned t = idx / (2 * qb), e = idx % (2 * qb);
        fe_t* Z = t == 0 ? a.A : (t == 1 ? a.B : a.C);
        const unsigned long long x = e < qb ? local_y(e) : q + local_y(e - qb);
        const fe_t v = bind1(Z[x], Z[x + 2 * q], r);
        bound[idx] = v;
        if (last_local) Z[x] = v;
      }
    } else if (local) {  // (nt * 2 qb <= TAIL_THREADS: one element per lane, so the in-place LDS update needs one barrier between its reads and its writes)
      const unsigned idx = threadIdx.x, t = idx / (2 * qb), e = idx % (2 * qb);
      const bool has = idx < nt * 2 * qb;
      fe_t v = fe_zero();
      fe_t* Z = t == 0 ? a.A : (t == 1 ? a.B : a.C);
      const unsigned long long x = e < qb ? local_y(e) : q + local_y(e - qb);
      if (has) v = bind1(bound[t * 4 * qb + e], bound[t * 4 * qb + 2 * qb + e], r);
      __syncthreads();
      if (has) {
        bound[idx] = v;
        if (last_local) Z[x] = v;
      }
    } else {
      for (unsigned idx = threadIdx.x; idx < nt * 2 * qb; idx += TAIL_THREADS) {
        const unsigned t = idx / (2 * qb), e = idx % (2 * qb);
        fe_t* Z = t == 0 ? a.A : (t == 1 ? a.B : a.C);
        const unsigned long long x = e < qb ? base + e : q + base + (e - qb);
        const fe_t v = foreign ? bind1(load_agent(Z + x), load_agent(Z + x + 2 * q), r) : bind1(Z[x], Z[x + 2 * q], r);
        Z[x] = v;
        bound[idx] = v;  // = bound[t * 2 qb + e]
      }
    }
    __syncthreads();
    SP_TT(2);
    if (tail_hand_over(len / 2, a.hand_n)) {
      // the host takes the remaining rounds (see tail_hand_over). One block is left here (q <= WQ: base = 0, qb = q), so bound[t * n + x] is element x
      // of table t, n = len / 2: value k = t * n + x goes to slot k / 3, element k % 3, tagged with this result's sequence number.
      const unsigned n = (unsigned)(len / 2), nv = nt * n;
      fe_t* wide = a.mapped + HAND_BASE_ELEM;
      if (threadIdx.x < nv) {
        const unsigned k = threadIdx.x;
        const fe_t t = bound[k];
        slot_store_elem(wide + 4 * (k / 3) + k % 3, t);
        slot_chk ck = {0u, 0u};
        slot_chk_add(ck, t, (int)(k % 3));
        chk_sh[k] = ck;
      }
      __syncthreads();
      if (threadIdx.x < (nv + 2) / 3) {
        const unsigned sidx = threadIdx.x;
        slot_chk chk = {0u, 0u};
        for (unsigned k = 3 * sidx; k < 3 * sidx + 3 && k < nv; ++k) {
          chk.a += chk_sh[k].a;
          chk.b += chk_sh[k].b;
        }
        slot_store_tag(wide + 4 * sidx, seq, chk);
      }
      // the final claims come back as the "challenges" answering seq, seq + 1 (, seq + 2): element 0 of every table, as a kernel that had run all
      // rounds itself would have left it
      for (unsigned t = 0; t < nt; ++t) {
        __syncthreads();
        if (!mail_wait(a.mail, a.mirror, a.mapped, seq + t, &r_sh)) return;
        if (threadIdx.x == 0) (t == 0 ? a.A : (t == 1 ? a.B : a.C))[0] = r_sh;
      }
      return;
    }
    if (tail_double(CUBIC, len / 2, a.hand_n)) {
      // the round over the n = len / 2 entries just bound AND the coefficient sums of the round after it (see TAIL_WIDE_VALS). y < qd = n / 4;
      // a0..a3 = A[y + k qd]. One product per lane, `which` wave-uniform:
      //   quadratic (8 sums): a0 b0 | a1 b1 | (a2-a0)(b2-b0) | (a3-a1)(b3-b1) | a2 b2 | (a1-a0)(b1-b0) | (a3-a1-a2+a0)(..) | (a3-a2)(b3-b2)
      //   cubic (12 sums, weights E' = E(rnd + 1, y) for 0..8, E(rnd, x) for 9..11 over x < 2 qd as two lane groups each):
      //     0: a0 b0 - c0 | 1: a2 b2 - c2 | 2: (a2-a0)(b2-b0) | 3: (a1-a0)(b1-b0) | 4: dU dV | 5: (a3-a2)(b3-b2)
      //     6: (2a0-a1)(2b0-b1) - (2c0-c1) | 7: dM dN, dM = 2(a2-a0) - (a3-a1) | 8: (2a2-a3)(2b2-b3) - (2c2-c3)
      //     9: t(0) | 10: t_inf | 11: t(-1) of this round
      const unsigned qd = (unsigned)(len / 8);
      const unsigned segd = qd < 64 ? 64u : qd;
      const unsigned grp = threadIdx.x / segd, il = threadIdx.x % segd;
      constexpr unsigned NGRP = CUBIC ? 15u : 8u;
      fe_t v = fe_zero();
      if (grp < NGRP && il < qd) {
        if (!CUBIC || grp < 9) {
          // (every case loads only the quarters it uses: all twelve at once would not fit the 128 registers of a 1024-thread block)
          // (one block here: base = 0, q = qb = 2 qd, so element x of table t is bound[t * 2 qb + x])
          auto qa = [&](unsigned k) { return bound[il + k * qd]; };
          auto qb2 = [&](unsigned k) { return bound[2 * qb + il + k * qd]; };
          auto qc = [&](unsigned k) { return bound[4 * qb + il + k * qd]; };
          auto prod_diff = [&](unsigned hi, unsigned lo) { return fe_mul<S>(fe_sub<S>(qa(hi), qa(lo)), fe_sub<S>(qb2(hi), qb2(lo))); };  // (a_hi - a_lo)(b_hi - b_lo)
          auto prod_2m = [&](unsigned k0, unsigned k1) {  // (2 a_k0 - a_k1)(2 b_k0 - b_k1)
            return fe_mul<S>(fe_sub<S>(fe_dbl<S>(qa(k0)), qa(k1)), fe_sub<S>(fe_dbl<S>(qb2(k0)), qb2(k1)));
          };
          if (!CUBIC) {
            switch (grp) {
              case 0: v = fe_mul<S>(qa(0), qb2(0)); break;
              case 1: v = fe_mul<S>(qa(1), qb2(1)); break;
              case 2: v = prod_diff(2, 0); break;
              case 3: v = prod_diff(3, 1); break;
              case 4: v = fe_mul<S>(qa(2), qb2(2)); break;
              case 5: v = prod_diff(1, 0); break;
              case 6: v = fe_mul<S>(fe_sub<S>(fe_sub<S>(qa(3), qa(1)), fe_sub<S>(qa(2), qa(0))), fe_sub<S>(fe_sub<S>(qb2(3), qb2(1)), fe_sub<S>(qb2(2), qb2(0)))); break;
              default: v = prod_diff(3, 2); break;
            }
          } else {
            switch (grp) {
              case 0: v = fe_sub<S>(fe_mul<S>(qa(0), qb2(0)), qc(0)); break;
              case 1: v = fe_sub<S>(fe_mul<S>(qa(2), qb2(2)), qc(2)); break;
              case 2: v = prod_diff(2, 0); break;
              case 3: v = prod_diff(1, 0); break;
              case 4: v = fe_mul<S>(fe_sub<S>(fe_sub<S>(qa(3), qa(1)), fe_sub<S>(qa(2), qa(0))), fe_sub<S>(fe_sub<S>(qb2(3), qb2(1)), fe_sub<S>(qb2(2), qb2(0)))); break;
              case 5: v = prod_diff(3, 2); break;
              case 6: v = fe_sub<S>(prod_2m(0, 1), fe_sub<S>(fe_dbl<S>(qc(0)), qc(1))); break;
              case 7:
                v = fe_mul<S>(fe_sub<S>(fe_dbl<S>(fe_sub<S>(qa(2), qa(0))), fe_sub<S>(qa(3), qa(1))), fe_sub<S>(fe_dbl<S>(fe_sub<S>(qb2(2), qb2(0))), fe_sub<S>(qb2(3), qb2(1))));
                break;
              default: v = fe_sub<S>(prod_2m(2, 3), fe_sub<S>(fe_dbl<S>(qc(2)), qc(3))); break;
            }
            v = fe_mul<S>(w_pre, v);
          }
        } else {  // cubic, this round's own three sums over the 2 qd pairs (x, x + n / 2): groups 9.. in pairs (low x, high x)
          const unsigned k = (grp - 9) / 2, x = il + ((grp - 9) & 1u) * qd, hn = 2 * qd;
          const fe_t a0 = bound[x], a1 = bound[x + hn], b0 = bound[2 * qb + x], b1 = bound[2 * qb + x + hn], c0 = bound[4 * qb + x], c1 = bound[4 * qb + x + hn];
          if (k == 0) v = fe_sub<S>(fe_mul<S>(a0, b0), c0);
          else if (k == 1) v = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
          else v = fe_sub<S>(fe_mul<S>(fe_sub<S>(fe_dbl<S>(a0), a1), fe_sub<S>(fe_dbl<S>(b0), b1)), fe_sub<S>(fe_dbl<S>(c0), c1));
          v = fe_mul<S>(w_pre, v);
        }
      }
      v = wave_sum_low(v, qd);  // (qd is a power of two; lanes il >= qd hold zero)
      const int lane2 = threadIdx.x & 63, wave2 = threadIdx.x >> 6;
      if (lane2 == 0) smem[wave2] = v;
      __syncthreads();
      constexpr unsigned NS = CUBIC ? (unsigned)TAIL_DOUBLE_SUMS_CUBIC : (unsigned)TAIL_DOUBLE_SUMS_QUAD;
      fe_t* wide = a.mapped + SLOT_BASE_ELEM;  // sum k -> slot k / 3, element k % 3
      if (threadIdx.x < NS) {  // thread k adds the waves of the lane group(s) of sum k
        const unsigned wps = segd / 64, k = threadIdx.x;
        const unsigned g0 = (CUBIC && k >= 9) ? 9 + 2 * (k - 9) : k, ng = (CUBIC && k >= 9) ? 2u : 1u;
        fe_t t = fe_zero();
        for (unsigned w = 0; w < ng * wps; ++w) t = fe_add<S>(t, smem[g0 * wps + w]);
        slot_store_elem(wide + 4 * (k / 3) + k % 3, t);
        slot_chk ck = {0u, 0u};
        slot_chk_add(ck, t, (int)(k % 3));
        chk_sh[k] = ck;
      }
      __syncthreads();
      if (threadIdx.x < (NS + 2) / 3) {  // thread s tags slot s
        const unsigned sidx = threadIdx.x;
        slot_chk chk = {0u, 0u};
        for (unsigned k = 3 * sidx; k < 3 * sidx + 3 && k < NS; ++k) {
          chk.a += chk_sh[k].a;
          chk.b += chk_sh[k].b;
        }
        slot_store_tag(wide + 4 * sidx, seq, chk);
      }
      seq += 2;
      rnd += 2;
      len /= 2;
      pending2 = true;
      __syncthreads();  // smem reuse
      continue;
    }
    // phase B: one (sum, pair) product per lane; `which` is wave-uniform
    const unsigned seg = qb < 64 ? 64u : qb;
    const unsigned which = threadIdx.x / seg, il = threadIdx.x % seg;
    fe_t v = fe_zero();
    if (which < (unsigned)NACC && il < qb) {
      const fe_t a0 = bound[il], a1 = bound[qb + il], b0 = bound[2 * qb + il], b1 = bound[3 * qb + il];
      if (which == 0) v = fe_mul<S>(a0, b0);
      else if (which == 1) v = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
      else v = fe_mul<S>(fe_sub<S>(fe_dbl<S>(a0), a1), fe_sub<S>(fe_dbl<S>(b0), b1));
      if (CUBIC) {
        const fe_t c0 = bound[4 * qb + il], c1 = bound[5 * qb + il];
        if (which == 0) v = fe_sub<S>(v, c0);
        else if (which == 2) v = fe_sub<S>(v, fe_sub<S>(fe_dbl<S>(c0), c1));
        v = fe_mul<S>(w_pre, v);
      }
    }
    SP_TT(3);
    v = wave_sum_low(v, qb);
    SP_TT(4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    SP_TT(5);
    if (threadIdx.x < (unsigned)NACC) {  // thread k adds the waves of sum k into this block's result slot
      const unsigned wps = seg / 64;
      fe_t t = smem[threadIdx.x * wps];
      for (unsigned w = 1; w < wps; ++w) t = fe_add<S>(t, smem[threadIdx.x * wps + w]);
      slot_store_elem(slot + threadIdx.x, t);
      slot_chk ck = {0u, 0u};
      slot_chk_add(ck, t, (int)threadIdx.x);
      chk_sh[threadIdx.x] = ck;
    }
    // while several blocks are active the next round reads other blocks' elements (other XCDs, other L2s): every wave first waits until its own
    // table stores of phase A have been acknowledged by the L2 (outside tgsplit mode the workgroup barrier alone does not wait for other
    // waves' stores), then — behind the barrier — thread 0 issues ONE agent-scope release (L2 write-back) for the whole block before the host
    // can see this block's slot. The single-block rounds need no fence at all.
    const bool release = local ? last_local : q > WQ;  // (a contiguous multi-block step only exists when the launch has one block per WQ pairs: never with `local`)
    if (release) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (release) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
      slot_chk chk = {0u, 0u};
      for (int k = 0; k < NACC; ++k) {
        chk.a += chk_sh[k].a;
        chk.b += chk_sh[k].b;
      }
      slot_store_tag(slot, seq, chk);
    }
    SP_TT(6);
    ++seq;
    ++rnd;
    len /= 2;
    __syncthreads();  // smem reuse
  }
}

}  // namespace spk
