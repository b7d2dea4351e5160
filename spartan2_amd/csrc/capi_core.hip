// libspartan_hip.so — context, device tables, transcript and the sum-check entry points of include/spartan_hip.h.
// The round loop lives below the C ABI: per round the device computes the evaluation sums (K2/K3), the host
// finishes the O(1) part (claim-derived evaluations, UniPoly, Keccak transcript), and the challenge goes back
// as a kernel argument of the bind (K1).
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "core.hpp"
#include "walk_pool.hpp"
#include "kernels_poly.hpp"

namespace sp {
static thread_local std::string g_err;
static thread_local sp_wait_hook g_wait_hook = nullptr;
static thread_local void* g_wait_user = nullptr;
void set_error(const std::string& m) { g_err = m; }
static std::atomic<int> g_live_contexts{0};
int live_contexts() { return g_live_contexts.load(std::memory_order_relaxed); }
void relax() {
  if (g_wait_hook) g_wait_hook(g_wait_user);
  else __builtin_ia32_pause();
}
void slow_note(const char* site, long spins) {
  static const bool on = getenv("SPARTAN_SLOWPATH_LOG") != nullptr;
  if (on) fprintf(stderr, "[slow path] %s after %ld polls\n", site, spins);
}
// Without a wait hook the runtime's own wait. SPARTAN_SYNC_SPIN_US = n polls hipStreamQuery / hipEventQuery for n microseconds first (round 6, an experiment
// on the "late host thread" of the NeutronNova proves - one step in a few hundred 7-30 ms long, always in a phase that synchronises a stream every round:
// hipStreamSynchronize sleeps on the completion signal, and a sleeping thread is woken when the scheduler gets to it). Measured at config 3, 2 x 2500 proves each
// way: the same handful of long steps with polling and without, medians 3.16-3.24 against 3.07-3.15 ms (a query is slower than the runtime's wait) - off.
static long sync_spin_us() {
  static const long v = [] {
    const char* e = getenv("SPARTAN_SYNC_SPIN_US");
    const long x = e ? atol(e) : 0;
    return x < 0 ? 0 : x;
  }();
  return v;
}
hipError_t stream_sync(hipStream_t s) {
  hipError_t e;
  if (!g_wait_hook) {
    const long spin = sync_spin_us();
    if (spin) {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned k = 0;; ++k) {
        if ((e = hipStreamQuery(s)) != hipErrorNotReady) return e;
        if ((k & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin)) break;
        __builtin_ia32_pause();
      }
    }
    return hipStreamSynchronize(s);
  }
  while ((e = hipStreamQuery(s)) == hipErrorNotReady) g_wait_hook(g_wait_user);
  return e;
}
hipError_t stream_sync_short(hipStream_t s) {
  hipError_t e;
  if (g_wait_hook) {
    while ((e = hipStreamQuery(s)) == hipErrorNotReady) g_wait_hook(g_wait_user);
    return e;
  }
  static const bool off = [] {
    const char* v = getenv("SPARTAN_SYNC_SHORT");  // "0": the runtime's blocking wait (A/B runs)
    return v && v[0] == '0';
  }();
  if (!off) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned k = 0;; ++k) {
      if ((e = hipStreamQuery(s)) != hipErrorNotReady) return e;
      if ((k & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) break;
      __builtin_ia32_pause();
    }
  }
  return hipStreamSynchronize(s);
}
hipError_t event_sync(hipEvent_t ev) {
  hipError_t e;
  if (!g_wait_hook) {
    const long spin = sync_spin_us();
    if (spin) {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned k = 0;; ++k) {
        if ((e = hipEventQuery(ev)) != hipErrorNotReady) return e;
        if ((k & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin)) break;
        __builtin_ia32_pause();
      }
    }
    return hipEventSynchronize(ev);
  }
  while ((e = hipEventQuery(ev)) == hipErrorNotReady) g_wait_hook(g_wait_user);
  return e;
}
int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}
int alloc_table(sp_ctx* ctx, size_t len, sp_table** out) {
  sp_table* t = new sp_table();
  t->ctx = ctx;
  t->cap = len ? len : 1;
  t->len = len;
  hipError_t e = hipMalloc((void**)&t->d, t->cap * sizeof(fe_t));
  if (e != hipSuccess) {
    delete t;
    return fail(SP_ERR_NO_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  *out = t;
  return SP_OK;
}
}  // namespace sp

using sp::fail;
typedef FqP S;

hipEvent_t sp_ctx::get_event() {
  if (!event_pool.empty()) {
    hipEvent_t e = event_pool.back();
    event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
int sp_ctx::ensure_scratch(size_t elems) {
  if (elems <= scratch_elems) return SP_OK;
  if (d_scratch) hipFree(d_scratch);
  d_scratch = nullptr;
  scratch_elems = 0;
  SP_HIP(hipMalloc((void**)&d_scratch, elems * sizeof(fe_t)));
  scratch_elems = elems;
  return SP_OK;
}
void* sp_ctx::workspace(int slot, size_t bytes, int lane) {
  slot += lane * WS_PER_LANE;
  if (bytes == 0) bytes = 16;
  if (bytes <= ws_bytes[slot]) return ws_ptr[slot];
  if (ws_ptr[slot]) {
    sp::stream_sync(stream);
    sp::stream_sync(stream2);
    hipFree(ws_ptr[slot]);
    ws_ptr[slot] = nullptr;
    ws_bytes[slot] = 0;
  }
  size_t want = bytes + bytes / 4;
  if (hipMalloc(&ws_ptr[slot], want) != hipSuccess) {
    sp::set_error("hipMalloc failed for a workspace buffer");
    return nullptr;
  }
  ws_bytes[slot] = want;
  return ws_ptr[slot];
}
void sp_ctx::drain_stats() {
  std::lock_guard<std::mutex> l(stats_mu);
  for (auto& kv : stats) {
    for (auto& pr : kv.second.pending) {
      float ms = 0;
      sp::event_sync(pr.second);
      hipEventElapsedTime(&ms, pr.first, pr.second);
      kv.second.ms += ms;
      event_pool.push_back(pr.first);
      event_pool.push_back(pr.second);
    }
    kv.second.pending.clear();
  }
}

static inline fe_t load_fe(const uint64_t* p) {
  fe_t r;
  memcpy(&r, p, 32);
  return r;
}
static inline void store_fe(uint64_t* p, const fe_t& a) { memcpy(p, &a, 32); }

extern "C" {

int sp_set_wait_hook(sp_wait_hook hook, void* user) {
  sp::g_wait_hook = hook;
  sp::g_wait_user = user;
  return SP_OK;
}
void sp_relax(void) { sp::relax(); }
const char* sp_last_error(void) { return sp::g_err.c_str(); }

int sp_ctx_create(int device, sp_ctx** out) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(SP_ERR_NO_DEVICE, std::string("no HIP device visible: libspartan_hip has no CPU fallback (hipGetDeviceCount: ") + hipGetErrorString(e) + ", " + std::to_string(count) + " devices)");
  if (device < 0 || device >= count) return fail(SP_ERR_NO_DEVICE, "device ordinal out of range");
  SP_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SP_HIP(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(SP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  sp_ctx* c = new sp_ctx();
  c->device = device;
  SP_HIP(hipStreamCreate(&c->stream));
  SP_HIP(hipStreamCreate(&c->stream2));
  SP_HIP(hipStreamCreateWithFlags(&c->stream_eq, hipStreamNonBlocking));
  SP_HIP(hipEventCreateWithFlags(&c->eq_ev, hipEventDisableTiming));
  SP_HIP(hipEventCreateWithFlags(&c->eq_read_ev, hipEventDisableTiming));
  SP_HIP(hipMalloc((void**)&c->d_eq_ahead, 2 * ((size_t)1 << 11) * sizeof(fe_t)));
  c->pinned_elems = spk::MAPPED_ELEMS;
  SP_HIP(hipHostMalloc((void**)&c->h_pinned, c->pinned_elems * sizeof(fe_t), hipHostMallocMapped));
  memset(c->h_pinned, 0, c->pinned_elems * sizeof(fe_t));
  SP_HIP(hipHostGetDevicePointer((void**)&c->d_pinned, c->h_pinned, 0));
  SP_HIP(hipHostMalloc(&c->h_pinned_lane[0], 8192));
  SP_HIP(hipHostMalloc(&c->h_pinned_lane[1], 8192));
  // the mailbox ring: in host memory (the mirror region of the mapped buffer) unless the device-memory form below is available
  c->h_mail = reinterpret_cast<volatile uint32_t*>(c->h_pinned + spk::MAIL_MIRROR_ELEM);
  c->d_mail = reinterpret_cast<const unsigned*>(c->d_pinned + spk::MAIL_MIRROR_ELEM);
  {  // SPARTAN_MAIL_DEV=0 keeps the mailbox in host memory (and with it the launch-after-challenge round loop)
    const char* e = getenv("SPARTAN_MAIL_DEV");
    const unsigned flags = hipDeviceMallocFinegrained;
    int large_bar = 0;
    if (!(e && e[0] == '0') && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) == hipSuccess && large_bar &&
        hipExtMallocWithFlags(&c->mail_alloc, 4096, flags) == hipSuccess) {
      SP_HIP(hipMemset(c->mail_alloc, 0, 4096));
      SP_HIP(hipDeviceSynchronize());
      c->h_mail_mirror = c->h_mail;  // the host-memory ring becomes the mirror of the device one
      c->d_mail_mirror = c->d_mail;
      c->h_mail = reinterpret_cast<volatile uint32_t*>(c->mail_alloc);  // the host stores straight into device memory over the BAR
      c->d_mail = reinterpret_cast<const unsigned*>(c->mail_alloc);
      c->mail_dev = true;
    }
  }
  SP_HIP(hipMalloc((void**)&c->d_gate, spk::MAIL_RING * sizeof(fe_t)));
  SP_HIP(hipMalloc((void**)&c->d_fold_tickets, spk::HOST_SUM_MAX_BLOCKS * sizeof(unsigned)));  // arrival counters of the folded second stage: zero between launches
  SP_HIP(hipMemset(c->d_fold_tickets, 0, spk::HOST_SUM_MAX_BLOCKS * sizeof(unsigned)));
  int rc = c->ensure_scratch(1 << 16);
  if (rc) return rc;
  sp::g_live_contexts.fetch_add(1, std::memory_order_relaxed);
  *out = c;
  return SP_OK;
}
int sp_ctx_bind_thread(sp_ctx* c) {
  if (!c) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_ctx_bind_thread: null context");
  SP_HIP(hipSetDevice(c->device));
  return SP_OK;
}
int sp_ctx_device(const sp_ctx* c) { return c ? c->device : -1; }
int sp_ctx_round_hooks_host_only(sp_ctx* c, int on) {
  if (!c) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_ctx_round_hooks_host_only: null context");
  c->hooks_host_only = on != 0;
  return SP_OK;
}
void sp_ctx_destroy(sp_ctx* c) {
  if (!c) return;
  sp::g_live_contexts.fetch_sub(1, std::memory_order_relaxed);
  sp::pcs_ahead_free(c);
  hipSetDevice(c->device);
  c->drain_stats();
  for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
  if (c->d_scratch) hipFree(c->d_scratch);
  for (int i = 0; i < sp_ctx::WS_SLOTS; ++i)
    if (c->ws_ptr[i]) hipFree(c->ws_ptr[i]);
  if (c->h_pinned) hipHostFree(c->h_pinned);
  if (c->mail_alloc) hipFree(c->mail_alloc);
  if (c->d_gate) hipFree(c->d_gate);
  if (c->d_fold_tickets) hipFree(c->d_fold_tickets);
  if (c->h_pinned_vec) hipHostFree(c->h_pinned_vec);
  if (c->vec_ev) hipEventDestroy(c->vec_ev);
  if (c->stream3) hipStreamDestroy(c->stream3);
  for (auto& lane : c->msm_ev)
    for (hipEvent_t& e : lane)
      if (e) hipEventDestroy(e);
  for (int i = 0; i < 2; ++i)
    if (c->h_pinned_lane[i]) hipHostFree(c->h_pinned_lane[i]);
  if (c->h_pinned_fb) hipHostFree(c->h_pinned_fb);
  if (c->h_pinned_fbs) hipHostFree(c->h_pinned_fbs);
  for (void* p_ : c->h_mm)
    if (p_) hipHostFree(p_);
  delete c->pcs_worker;
  if (c->h_pcs) hipHostFree(c->h_pcs);
  if (c->pcs_ev) hipEventDestroy(c->pcs_ev);
  for (void* p_ : c->h_fbm)
    if (p_) hipHostFree(p_);
  for (void* p_ : c->d_mm_work)
    if (p_) hipFree(p_);
  if (c->h_stage) hipHostFree(c->h_stage);
  for (hipEvent_t e : c->stage_ev)
    if (e) hipEventDestroy(e);
  if (c->fb_ev) hipEventDestroy(c->fb_ev);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->stream_tab) {
    hipStreamSynchronize(c->stream_tab);
    hipStreamDestroy(c->stream_tab);
  }
  if (c->tab_scratch) hipFree(c->tab_scratch);
  if (c->aside_ev) hipEventDestroy(c->aside_ev);
  if (c->tail_ev) hipEventDestroy(c->tail_ev);
  if (c->aside_main_ev) hipEventDestroy(c->aside_main_ev);
  if (c->stream_eq) hipStreamDestroy(c->stream_eq);
  if (c->eq_ev) hipEventDestroy(c->eq_ev);
  if (c->eq_read_ev) hipEventDestroy(c->eq_read_ev);
  if (c->cubic_ev) hipEventDestroy(c->cubic_ev);
  if (c->d_cubic_eq) hipFree(c->d_cubic_eq);
  if (c->d_eq_ahead) hipFree(c->d_eq_ahead);
  delete c;
}
int sp_ctx_synchronize(sp_ctx* c) {
  SP_HIP(sp::stream_sync(c->stream));
  return SP_OK;
}
int sp_ctx_reset_stats(sp_ctx* c, int enable) {
  c->drain_stats();
  c->stats.clear();
  c->timing = enable != 0;
  return SP_OK;
}
int sp_ctx_stats_filter(sp_ctx* c, const char* only) {
  c->timing_only = only ? only : "";
  return SP_OK;
}
int sp_ctx_mail_stats(sp_ctx* c, uint64_t out[5]) {
  for (int i = 0; i < 5; ++i) out[i] = 0;
  if (!c->mail_dev) return SP_OK;
  out[4] = 1;
  uint32_t w[4];
  SP_HIP(hipSetDevice(c->device));
  SP_HIP(sp::stream_sync(c->stream));
  SP_HIP(hipMemcpy(w, reinterpret_cast<const uint32_t*>(c->mail_alloc) + spk::MAIL_DIAG_WORD, sizeof w, hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; ++i) out[i] = w[i];
  SP_HIP(hipMemset(reinterpret_cast<uint32_t*>(c->mail_alloc) + spk::MAIL_DIAG_WORD, 0, sizeof w));
  return SP_OK;
}
int sp_ctx_kernel_stats(sp_ctx* c, const char* what, double* ms, uint64_t* launches, uint64_t* alg_bytes) {
  SP_HIP(sp::stream_sync(c->stream));
  SP_HIP(sp::stream_sync(c->stream2));
  if (c->stream3) SP_HIP(sp::stream_sync(c->stream3));
  c->drain_stats();
  auto it = c->stats.find(what);
  if (it == c->stats.end()) {
    *ms = 0;
    *launches = 0;
    *alg_bytes = 0;
    return SP_OK;
  }
  *ms = it->second.ms;
  *launches = it->second.launches;
  *alg_bytes = it->second.bytes;
  return SP_OK;
}

// ---- tables ---------------------------------------------------------------------------------------------------
int sp_table_from_host(sp_ctx* c, const uint64_t* z, size_t len, size_t lo_eff, size_t hi_eff, sp_table** out) {
  sp_table* t;
  int rc = sp::alloc_table(c, len, &t);
  if (rc) return rc;
  t->lo_eff = lo_eff;
  t->hi_eff = hi_eff;
  if (len) SP_HIP(hipMemcpyAsync(t->d, z, len * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  *out = t;
  return SP_OK;
}
int sp_table_zeros(sp_ctx* c, size_t len, size_t lo_eff, size_t hi_eff, sp_table** out) {
  sp_table* t;
  int rc = sp::alloc_table(c, len, &t);
  if (rc) return rc;
  t->lo_eff = lo_eff;
  t->hi_eff = hi_eff;
  if (len) SP_HIP(hipMemsetAsync(t->d, 0, len * sizeof(fe_t), c->stream));
  *out = t;
  return SP_OK;
}
int sp_table_write(sp_ctx* c, sp_table* t, size_t off, const uint64_t* z, size_t cnt) {
  if (off + cnt > t->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_write: range exceeds the table");
  if (cnt) SP_HIP(hipMemcpyAsync(t->d + off, z, cnt * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  SP_HIP(sp::stream_sync(c->stream));  // the host buffer is only borrowed for the duration of the call
  return SP_OK;
}
int sp_table_write_async(sp_ctx* c, sp_table* t, size_t off, const uint64_t* z, size_t cnt) {
  if (off + cnt > t->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_write_async: range exceeds the table");
  if (cnt > 2048) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_write_async: at most 2048 elements per call");
  if (cnt == 0) return SP_OK;
  if (!c->h_stage) {
    SP_HIP(hipHostMalloc(&c->h_stage, 4 * 65536));
    for (hipEvent_t& e : c->stage_ev) SP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const unsigned slot = c->stage_next++ & 3u;
  SP_HIP(sp::event_sync(c->stage_ev[slot]));  // the copy that used this slot four writes ago (long finished)
  void* stage = (char*)c->h_stage + (size_t)slot * 65536;
  memcpy(stage, z, cnt * sizeof(fe_t));
  SP_HIP(hipMemcpyAsync(t->d + off, stage, cnt * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  SP_HIP(hipEventRecord(c->stage_ev[slot], c->stream));
  return SP_OK;
}
// ---- witness upload as machine words / bits (sp_table_write_u64, sp_table_write_bits) ----------------------------------------------------------------
namespace {
// out[i] = vals[i] as a Montgomery-form element: 0 and 1 (almost every entry of a booleanised witness) are constants, anything else one product by R^2
__global__ void __launch_bounds__(256) k_expand_u64(const uint64_t* __restrict__ vals, size_t n, fe_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t v = vals[i];
  out[i] = v == 0 ? fe_zero() : (v == 1 ? fe_one<S>() : fe_from_u64<S>(v));
}
// out[i] = bit (i & 7) of bits[i >> 3]: a thread expands one byte into eight elements (a wave writes 16 KiB contiguous)
__global__ void __launch_bounds__(256) k_expand_bits(const uint8_t* __restrict__ bits, size_t n, fe_t* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // element index: lanes write consecutive elements, eight lanes share a byte
  if (e >= n) return;
  out[e] = ((bits[e >> 3] >> (e & 7)) & 1u) ? fe_one<S>() : fe_zero();
}
}  // namespace
// device staging of the raw words (grow-only workspace) behind a host -> device copy on the context's stream
static int upload_raw(sp_ctx* c, const void* src, size_t bytes, void** d_raw) {
  void* d = c->workspace(sp_ctx::WS_SCALARS_RAW, bytes);
  if (!d) return SP_ERR_NO_DEVICE;
  SP_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, c->stream));
  *d_raw = d;
  return SP_OK;
}
int sp_table_write_u64(sp_ctx* c, sp_table* t, size_t off, const uint64_t* vals, size_t cnt) {
  if (!t || off > t->cap || cnt > t->cap - off) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_write_u64: range exceeds the table");
  if (cnt == 0) return SP_OK;
  if (!vals) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_write_u64: null values");
  void* d_raw = nullptr;
  int rc = upload_raw(c, vals, cnt * sizeof(uint64_t), &d_raw);
  if (rc) return rc;
  c->timed("expand_witness", 40ull * cnt, [&] {
    hipLaunchKernelGGL(k_expand_u64, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, (const uint64_t*)d_raw, cnt, t->d + off);
  });
  SP_HIP(sp::stream_sync(c->stream));  // the caller's buffer is only borrowed, and the staging workspace is shared with the MSM entry points
  return SP_OK;
}
int sp_table_write_bits(sp_ctx* c, sp_table* t, size_t off, const uint8_t* bits, size_t cnt) {
  if (!t || off > t->cap || cnt > t->cap - off) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_write_bits: range exceeds the table");
  if (cnt == 0) return SP_OK;
  if (!bits) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_write_bits: null values");
  void* d_raw = nullptr;
  int rc = upload_raw(c, bits, (cnt + 7) / 8, &d_raw);
  if (rc) return rc;
  c->timed("expand_witness", 32ull * cnt, [&] {
    hipLaunchKernelGGL(k_expand_bits, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t*)d_raw, cnt, t->d + off);
  });
  SP_HIP(sp::stream_sync(c->stream));
  return SP_OK;
}
int sp_table_zero(sp_ctx* c, sp_table* t, size_t off, size_t cnt) {
  if (off + cnt > t->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_zero: range exceeds the table");
  if (cnt) SP_HIP(hipMemsetAsync(t->d + off, 0, cnt * sizeof(fe_t), c->stream));
  return SP_OK;
}
int sp_table_copy(sp_ctx* c, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t cnt) {
  if (dst_off + cnt > dst->cap || src_off + cnt > src->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_copy: range exceeds a table");
  if (cnt) SP_HIP(hipMemcpyAsync(dst->d + dst_off, src->d + src_off, cnt * sizeof(fe_t), hipMemcpyDeviceToDevice, c->stream));
  return SP_OK;
}
// dst[dst_off, dst_off + cnt) = src[src_off, ..) and dst[zero_off, zero_off + zero_cnt) = 0 in ONE launch on the eq stream, beside the main stream: the
// bulk of z = [W | 1 | public | challenges | 0 ...] (src/spartan.rs:246-253) has no reader before the inner sum-check, so its 96 MB at config 2 need not
// sit in front of the matrix-vector product on the main stream (three blit launches with gaps: 34 us). behind_queued != 0 orders it behind whatever the main
// stream held at the call; with 0 the caller vouches that nothing queued touches the two ranges (the headline driver: the previous prove on the state has
// delivered all its results, and the product queued in front reads other columns). Nothing on the main stream is ordered behind it until sp_ctx_aside_join.
namespace {
__global__ void __launch_bounds__(256) k_copy_and_zero(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t ncopy, uint4* __restrict__ zdst, size_t nzero) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t0; i < ncopy; i += stride) dst[i] = src[i];
  const uint4 z = {0u, 0u, 0u, 0u};
  for (size_t i = t0; i < nzero; i += stride) zdst[i] = z;
}
}  // namespace
int sp_table_assemble_aside(sp_ctx* c, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t cnt, size_t zero_off, size_t zero_cnt,
                            int behind_queued) {
  if (dst_off + cnt > dst->cap || src_off + cnt > src->cap || zero_off + zero_cnt > dst->cap)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_assemble_aside: range exceeds a table");
  if (!c->aside_ev) SP_HIP(hipEventCreateWithFlags(&c->aside_ev, hipEventDisableTiming));
  if (behind_queued) {
    if (!c->aside_main_ev) SP_HIP(hipEventCreateWithFlags(&c->aside_main_ev, hipEventDisableTiming));
    SP_HIP(hipEventRecord(c->aside_main_ev, c->stream));
    SP_HIP(hipStreamWaitEvent(c->stream_eq, c->aside_main_ev, 0));
  } else if (c->tail_ev_pending) {
    // not behind the main stream's queue, but behind the last resident quadratic tail: its final store into element 0 of its tables (z among them, when
    // the caller is a prove that reuses the state of the one before) may come after that sum-check returned. The event is long complete in practice.
    SP_HIP(hipStreamWaitEvent(c->stream_eq, c->tail_ev, 0));
    c->tail_ev_pending = false;
  }
  if (cnt || zero_cnt) {
    const size_t work = 2 * (cnt > zero_cnt ? cnt : zero_cnt);  // uint4 per element: 2
    size_t blocks = (work + 4 * 256 - 1) / (4 * 256);           // ~4 vectors per thread and range
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_copy_and_zero, dim3((unsigned)blocks), dim3(256), 0, c->stream_eq, reinterpret_cast<uint4*>(dst->d + dst_off),
                       reinterpret_cast<const uint4*>(src->d + src_off), 2 * cnt, reinterpret_cast<uint4*>(dst->d + zero_off), 2 * zero_cnt);
  }
  SP_HIP(hipEventRecord(c->aside_ev, c->stream_eq));
  c->aside_pending = true;
  return SP_OK;
}
int sp_ctx_aside_join(sp_ctx* c) {
  if (c->aside_pending) SP_HIP(hipStreamWaitEvent(c->stream, c->aside_ev, 0));
  c->aside_pending = false;
  return SP_OK;
}
int sp_table_gather_strided(sp_ctx* c, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t stride, size_t cnt) {
  if (stride == 0 || dst_off + cnt > dst->cap || (cnt && src_off + (cnt - 1) * stride >= src->cap))
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_gather_strided: range exceeds a table");
  if (cnt)
    SP_HIP(hipMemcpy2DAsync(dst->d + dst_off, sizeof(fe_t), src->d + src_off, stride * sizeof(fe_t), sizeof(fe_t), cnt, hipMemcpyDeviceToDevice, c->stream));
  return SP_OK;
}
int sp_table_read(sp_ctx* c, const sp_table* t, size_t off, size_t cnt, uint64_t* out) {
  if (off + cnt > t->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_read: range exceeds the table");
  if (cnt) SP_HIP(hipMemcpyAsync(out, t->d + off, cnt * sizeof(fe_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  return SP_OK;
}
int sp_table_info(const sp_table* t, size_t* len, size_t* lo_eff, size_t* hi_eff) {
  if (len) *len = t->len;
  if (lo_eff) *lo_eff = t->lo_eff;
  if (hi_eff) *hi_eff = t->hi_eff;
  return SP_OK;
}
int sp_table_set_len(sp_table* t, size_t len, size_t lo_eff, size_t hi_eff) {
  if (len > t->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_set_len: longer than the allocation");
  t->len = len;
  t->lo_eff = lo_eff;
  t->hi_eff = hi_eff;
  return SP_OK;
}
int sp_table_scatter_strided(sp_ctx* c, sp_table* dst, size_t dst_off, size_t stride, const sp_table* src, size_t src_off, size_t cnt) {
  if (cnt == 0) return SP_OK;
  if (stride == 0 || src_off + cnt > src->cap || dst_off + (cnt - 1) * stride >= dst->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_scatter_strided: range outside the tables");
  SP_HIP(hipMemcpy2DAsync(dst->d + dst_off, stride * sizeof(fe_t), src->d + src_off, sizeof(fe_t), sizeof(fe_t), cnt, hipMemcpyDeviceToDevice, c->stream));
  return SP_OK;
}
int sp_table_view(const sp_table* t, size_t off, size_t len, sp_table** out) {
  if (!t || off > t->cap || len > t->cap - off) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_view: window outside the allocation");
  sp_table* v = new sp_table();
  v->ctx = t->ctx;
  v->d = t->d + off;
  v->cap = v->len = len;
  v->view = true;
  *out = v;
  return SP_OK;
}
int sp_table_device_ptr(const sp_table* t, void** out, size_t* cap_bytes) {
  if (!t || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_device_ptr: null argument");
  *out = t->d;
  if (cap_bytes) *cap_bytes = t->cap * sizeof(fe_t);
  return SP_OK;
}
void sp_table_free(sp_table* t) {
  if (!t) return;
  if (t->d && !t->view) hipFree(t->d);
  delete t;
}

}  // extern "C"

// bind up to 4 tables with one launch
static int launch_bind(sp_ctx* c, sp_table** tabs, int nt, const fe_t& r) {
  if (nt > spk::BIND_MAX_TABLES) return fail(SP_ERR_INTERNAL, "launch_bind: too many tables");
  spk::BindArgs a;
  size_t max_eff = 0;
  uint64_t bytes = 0;
  for (int i = 0; i < nt; ++i) {
    sp_table* t = tabs[i];
    if (t->len < 2) return fail(SP_ERR_INVALID_INPUT_LENGTH, "bind_poly_var_top: table must have at least two elements");
    a.z[i] = t->d;
    a.n[i] = t->len / 2;
    a.lo[i] = sp::eff_lo(t);
    a.hi[i] = sp::eff_hi(t);
    size_t eff = sp::eff_pairs(t);
    if (eff > max_eff) max_eff = eff;
    bytes += 48ull * t->len;  // read 32*len, write 16*len per table per round
  }
  a.r = r;
  a.one_minus_r = fe_sub<S>(fe_one<S>(), r);
  if (max_eff > 0) {
    size_t blocks = (max_eff + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks, (unsigned)nt);
    c->timed("bind", bytes, [&] { hipLaunchKernelGGL(spk::k_bind_top, grid, dim3(256), 0, c->stream, a); });
  }
  for (int i = 0; i < nt; ++i) sp::after_bind(tabs[i]);
  return SP_OK;
}

// second-stage reduction of `nblocks` x nacc block partials in d_scratch, written straight into mapped pinned host memory.
// Every evaluation launch is given c->result_seq (bumped by next_seq() just before the launch); whichever kernel produces the
// final sums publishes that number after the data, and the host polls for it.
static unsigned next_seq(sp_ctx* c) { return ++c->result_seq; }
static void reduce_partials_launch(sp_ctx* c, size_t nblocks, int nacc) {
  if (nblocks <= (size_t)spk::HOST_SUM_MAX_BLOCKS) {  // the blocks write their sums into host slots; reduce_partials_wait adds them
    c->pending_slots = (unsigned)nblocks;
    return;
  }
  // more producer blocks than slots: single-wave second-stage blocks, one slot each (kernels_poly.hpp k_sum_partials)
  size_t b2 = (nblocks + 63) / 64;
  if (b2 > (size_t)spk::HOST_SUM_MAX_BLOCKS) b2 = spk::HOST_SUM_MAX_BLOCKS;
  c->pending_slots = (unsigned)b2;
  hipLaunchKernelGGL(spk::k_sum_partials, dim3((unsigned)b2), dim3(64), 0, c->stream, c->d_scratch, nblocks, nacc, c->d_pinned, c->result_seq);
}
// Second stage of a streaming launch: groups of 2^gl consecutive producer blocks, eq_out[group] applied when given: k_sum_partials_lazy behind the producer,
// or - SPARTAN_FOLD_STAGE2=1, measured and not kept as the default - FOLDED into the producer (kernels_poly.hpp stream_block_partials: the last block to
// arrive at a slot's ticket does the slot's sums). lazy_out() builds the kernel's argument before the launch, sum_lazy_launch() finishes after it.
static bool fold_stage2_enabled() {
  static const bool on = [] {
    // "1": fold. OFF by default - measured behind the separate launch on the same box (tools/ab/run_env.sh, 3 x 200 proves each; profiles/r06_fold_stage2.txt):
    // 0.906 (64 slots) / 0.934 (16) / 0.931 (4) against 0.873 ms - every block ends on a write-through store's acknowledgement and an agent-scope
    // atomic (1-2 us each, and a streaming launch is ONE generation of blocks, so its end IS the kernel's end), and the last block reads the partials past
    // its L2: together more than the 5-7 us second-stage launch they replace.
    const char* e = getenv("SPARTAN_FOLD_STAGE2");
    return e && e[0] == '1';
  }();
  return on;
}
static spk::LazyOut lazy_out(sp_ctx* c, spk::lazy9_t* lp, size_t nparts, int gl, const fe_t* eq_out, unsigned seq) {
  spk::LazyOut lo(lp);
  const size_t ngroups = nparts >> gl;
  size_t b2 = (ngroups + 63) / 64;
  if (b2 < 1) b2 = 1;
  if (b2 > (size_t)spk::HOST_SUM_MAX_BLOCKS) b2 = spk::HOST_SUM_MAX_BLOCKS;
  lo.eq_out = eq_out;
  lo.mapped = c->d_pinned;
  lo.seq = seq;
  lo.nslots = (unsigned)b2;
  lo.gl = gl;
  // folded: as many slots as the host adds without a second stage (HOST_SUM_MAX_BLOCKS), each the contiguous run of per_slot blocks = a whole number of
  // groups - a ticket is then contended by nparts / 64 blocks (16 at 1024) instead of 256; a group's words must be readable as 8-byte pairs
  static const unsigned fold_slots = [] {
    const char* e = getenv("SPARTAN_FOLD_SLOTS");
    const int v = e ? atoi(e) : spk::HOST_SUM_MAX_BLOCKS;
    return (unsigned)(v < 1 ? 1 : (v > spk::HOST_SUM_MAX_BLOCKS ? spk::HOST_SUM_MAX_BLOCKS : v));
  }();
  size_t fs = fold_slots;
  while (fs > 1 && (fs > ngroups || ngroups % fs)) fs >>= 1;
  const bool even = ngroups > 0 && ((size_t)ngroups << gl) == nparts && ngroups % fs == 0 && (gl == 0 || nparts % 2 == 0);
  if (fold_stage2_enabled() && even && c->d_fold_tickets) {
    lo.tickets = c->d_fold_tickets;
    lo.nslots = (unsigned)fs;
    lo.per_slot = (unsigned)(nparts / fs);
  }
  return lo;
}
static void sum_lazy_launch(sp_ctx* c, const spk::LazyOut& lo, size_t nparts) {
  const int gl = lo.gl;
  const unsigned seq = lo.seq;
  const fe_t* eq_out = lo.eq_out;
  c->pending_slots = lo.nslots;
  if (lo.tickets) return;  // the producer's last blocks deliver the slots
  const uint32_t* P = reinterpret_cast<const uint32_t*>(lo.P);
  const dim3 g(lo.nslots), b(64);
  const bool vec_ok = nparts % 8 == 0;  // the vector loads of the templated forms need their 16-byte alignment
  if (vec_ok && gl == 0) hipLaunchKernelGGL((spk::k_sum_partials_lazy<0>), g, b, 0, c->stream, P, nparts, gl, eq_out, c->d_pinned, seq);
  else if (vec_ok && gl == 1) hipLaunchKernelGGL((spk::k_sum_partials_lazy<1>), g, b, 0, c->stream, P, nparts, gl, eq_out, c->d_pinned, seq);
  else if (vec_ok && gl == 2) hipLaunchKernelGGL((spk::k_sum_partials_lazy<2>), g, b, 0, c->stream, P, nparts, gl, eq_out, c->d_pinned, seq);
  else if (vec_ok && gl == 3) hipLaunchKernelGGL((spk::k_sum_partials_lazy<3>), g, b, 0, c->stream, P, nparts, gl, eq_out, c->d_pinned, seq);
  else hipLaunchKernelGGL((spk::k_sum_partials_lazy<-1>), g, b, 0, c->stream, P, nparts, gl, eq_out, c->d_pinned, seq);
}
// `resident`: a kernel on the stream is itself waiting for the host's next challenge (the resident tail that produces the result, or a launch issued
// ahead of its challenge queued behind the producer) — a stream synchronise would not return before that kernel's watchdog, so the host keeps
// polling (bounded by wall-clock; the kernel gives up after 8 s as well).
// one self-validating slot (kernels_poly.hpp slot_store_tag): wait for its sequence word, then re-read until the check word matches the data
static int wait_slot(sp_ctx* c, const fe_t* slot, unsigned want, int nvals, fe_t* v, bool resident, long* spins) {
  volatile const unsigned* fl = reinterpret_cast<volatile const unsigned*>(slot + 3);
  while (*fl != want) {
    if (++*spins > 400000) {  // a few ms
      sp::slow_note("wait_slot", *spins);
      if (resident) {        // the tail kernel is itself waiting for the host: keep polling, bounded by wall-clock
        const auto t0 = std::chrono::steady_clock::now();
        while (*fl != want) {
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(12)) return fail(SP_ERR_INTERNAL, "sum-check tail kernel did not deliver a round result");
          sp::relax();
        }
        break;
      }
      SP_HIP(sp::stream_sync(c->stream));  // e.g. under a profiler
      if (*fl != want) return fail(SP_ERR_INTERNAL, "evaluation kernel did not deliver its block sums");
    }
    sp::relax();
  }
  for (long tries = 0;; ++tries) {
    std::atomic_thread_fence(std::memory_order_acquire);
    spk::slot_chk chk = {want, want * spk::SLOT_CHK_K};
    for (int k = 0; k < nvals; ++k) {
      for (int i = 0; i < 8; ++i) v[k].v[i] = reinterpret_cast<volatile const uint32_t*>(slot + k)[i];
      spk::slot_chk_add(chk, v[k], k);
    }
    if (fl[0] == want && fl[1] == chk.a && fl[2] == chk.b && fl[3] == want) return SP_OK;
    if (tries > 4000000) return fail(SP_ERR_INTERNAL, "evaluation kernel delivered an inconsistent result slot");
    sp::relax();
  }
}
// the result of a two-round trip of the resident tail (kernels_poly.hpp TAIL_WIDE_VALS): nvals sums, three per result slot from slot 0 on
static int wait_wide(sp_ctx* c, unsigned want, int nvals, fe_t* v) {
  long spins = 0;
  for (int s0 = 0; s0 < nvals; s0 += 3) {
    int rc = wait_slot(c, c->h_pinned + spk::SLOT_BASE_ELEM + 4 * (s0 / 3), want, nvals - s0 < 3 ? nvals - s0 : 3, v + s0, true, &spins);
    if (rc) return rc;
  }
  return SP_OK;
}
static unsigned host_parts(size_t items, size_t min_per_part) {
  const size_t w = (size_t)sp::WalkPool::get().walkers() + 1;
  size_t p = items / min_per_part;
  if (p > w) p = w;
  if (p > (size_t)sp::WalkPool::MAX_PARTS) p = sp::WalkPool::MAX_PARTS;
  return p ? (unsigned)p : 1u;
}
// The tables of a hand-over (kernels_poly.hpp tail_hand_over): nvals elements, three per slot of the hand-over area from slot 0 on. Slot 0 is polled;
// once it has landed the other lines are on their way (one burst of posted writes): their cache misses are started together before the slots are
// validated one by one (a slot whose tag or check words are not there yet is polled like any result slot).
static int wait_hand_over(sp_ctx* c, unsigned want, int nvals, fe_t* v) {
  long spins = 0;
  const fe_t* area = c->h_pinned + spk::HAND_BASE_ELEM;
  int rc = wait_slot(c, area, want, nvals < 3 ? nvals : 3, v, true, &spins);
  if (rc) return rc;
  for (int s0 = 3; s0 < nvals; s0 += 3) {
    const char* line = reinterpret_cast<const char*>(area + 4 * (s0 / 3));
    __builtin_prefetch(line);
    __builtin_prefetch(line + 64);
  }
  const int nslots = (nvals + 2) / 3;
  if (nslots > 48 && sp::WalkPool::get().walkers() > 0) {  // a large hand-over: the slots are checked (and copied out) by the walkers, 16+ slots a part
    struct Check {
      sp_ctx* c;
      const fe_t* area;
      unsigned want;
      int nvals, nslots;
      fe_t* v;
      std::atomic<int> rc{SP_OK};
    } ck{c, area, want, nvals, nslots, v};
    auto part = [](void* arg, unsigned p, unsigned np) {
      Check& k = *static_cast<Check*>(arg);
      const int lo = 1 + (int)((long)(k.nslots - 1) * p / np), hi = 1 + (int)((long)(k.nslots - 1) * (p + 1) / np);
      long sp_ = 0;
      for (int sl = lo; sl < hi; ++sl) {
        const int s0 = 3 * sl;
        const int r = wait_slot(k.c, k.area + 4 * sl, k.want, k.nvals - s0 < 3 ? k.nvals - s0 : 3, k.v + s0, true, &sp_);
        if (r) k.rc.store(r, std::memory_order_relaxed);
      }
    };
    sp::WalkPool::get().run(host_parts((size_t)nslots, 16), part, &ck);
    return ck.rc.load(std::memory_order_relaxed);
  }
  for (int s0 = 3; s0 < nvals; s0 += 3)
    if ((rc = wait_slot(c, area + 4 * (s0 / 3), want, nvals - s0 < 3 ? nvals - s0 : 3, v + s0, true, &spins))) return rc;
  return SP_OK;
}
// ---- the host's rounds behind a hand-over, spread over the process's polling threads (walk_pool.hpp) ---------------------------------------------------
// Entries a table may still have when the resident tail hands it to the host (TailArgs::hand_n): 16 (cubic) / 32 (quadratic), the rounds one thread runs
// in ~3 us. SPARTAN_HAND_N_CUBIC / SPARTAN_HAND_N_QUAD move it up to the tail's first one-block step (256 / 512: 24 / 32 KiB of tables in one burst, the
// host's rounds spread over the walkers) - built and measured in round 6, NOT the default: the burst and its validation make that trip 15-19 us instead
// of 9-10, a parallel region costs ~2 us of hand-shakes even with the walkers on the caller's L3, and the rounds it replaces were already two to a trip -
// outer sum-check -3 us, inner +2 us at config 2 (profiles/r06_hand_over.txt).
static unsigned tail_hand_n(bool cubic) {
  static const unsigned v[2] = {
      [] {
        const char* e = getenv("SPARTAN_HAND_N_QUAD");
        unsigned n = e ? (unsigned)atoi(e) : 32u;
        return n > spk::TAIL_HAND_N_MAX_QUAD ? (unsigned)spk::TAIL_HAND_N_MAX_QUAD : n;
      }(),
      [] {
        const char* e = getenv("SPARTAN_HAND_N_CUBIC");
        unsigned n = e ? (unsigned)atoi(e) : 16u;
        return n > spk::TAIL_HAND_N_MAX_CUBIC ? (unsigned)spk::TAIL_HAND_N_MAX_CUBIC : n;
      }()};
  return v[cubic ? 1 : 0];
}
// bind_poly_var_top (src/polys/multilinear.rs:95-164) on `ntab` host tables of n entries, `stride` apart
struct HostBind {
  fe_t* base;
  size_t stride, n, ntab;
  fe_t r;
};
static void host_bind_part(void* arg, unsigned p, unsigned np) {
  const HostBind& b = *static_cast<const HostBind*>(arg);
  const size_t h = b.n / 2, total = b.ntab * h, lo = total * p / np, hi = total * (p + 1) / np;
  for (size_t i = lo; i < hi; ++i) {
    fe_t* Z = b.base + (i / h) * b.stride;
    const size_t x = i % h;
    Z[x] = fe_add<S>(Z[x], fe_mul<S>(b.r, fe_sub<S>(Z[x + h], Z[x])));
  }
}
static void host_bind_tables(fe_t* base, size_t stride, size_t ntab, size_t n, const fe_t& r) {
  HostBind b{base, stride, n, ntab, r};
  const unsigned np = host_parts(ntab * (n / 2), 24);
  if (np > 1) sp::WalkPool::get().run(np, host_bind_part, &b);
  else host_bind_part(&b, 0, 1);
}
// compute_eval_points_quad (src/sumcheck.rs:128-174) on host tables
struct HostQuadEval {
  const fe_t *a, *b;
  size_t half;
  fe_t out[sp::WalkPool::MAX_PARTS][2];
};
static void host_quad_part(void* arg, unsigned p, unsigned np) {
  HostQuadEval& q = *static_cast<HostQuadEval*>(arg);
  const size_t lo = q.half * p / np, hi = q.half * (p + 1) / np;
  fe_t s0 = fe_zero(), s1 = fe_zero();
  for (size_t x = lo; x < hi; ++x) {
    s0 = fe_add<S>(s0, fe_mul<S>(q.a[x], q.b[x]));
    s1 = fe_add<S>(s1, fe_mul<S>(fe_sub<S>(q.a[x + q.half], q.a[x]), fe_sub<S>(q.b[x + q.half], q.b[x])));
  }
  q.out[p][0] = s0;
  q.out[p][1] = s1;
}
static void host_quad_eval(const fe_t* a, const fe_t* b, size_t half, fe_t sums[2]) {
  HostQuadEval q;
  q.a = a;
  q.b = b;
  q.half = half;
  const unsigned np = host_parts(half, 24);
  if (np > 1) sp::WalkPool::get().run(np, host_quad_part, &q);
  else host_quad_part(&q, 0, 1);
  sums[0] = sums[1] = fe_zero();
  for (unsigned p = 0; p < np; ++p) {
    sums[0] = fe_add<S>(sums[0], q.out[p][0]);
    sums[1] = fe_add<S>(sums[1], q.out[p][1]);
  }
}
// evaluation_points_cubic_with_three_inputs + t(-1) (src/sumcheck.rs:1025-1156, :1327-1396) on host tables: pairs (x, x + n / 2) weighted with E[x]
struct HostCubicEval {
  const fe_t *a, *b, *c, *E;
  size_t hn;
  fe_t out[sp::WalkPool::MAX_PARTS][3];
};
static void host_cubic_part(void* arg, unsigned p, unsigned np) {
  HostCubicEval& q = *static_cast<HostCubicEval*>(arg);
  const size_t hn = q.hn, lo = hn * p / np, hi = hn * (p + 1) / np;
  fe_t s[3] = {fe_zero(), fe_zero(), fe_zero()};
  for (size_t x = lo; x < hi; ++x) {
    const fe_t a0 = q.a[x], a1 = q.a[x + hn], b0 = q.b[x], b1 = q.b[x + hn], c0 = q.c[x], c1 = q.c[x + hn];
    const fe_t v0 = fe_sub<S>(fe_mul<S>(a0, b0), c0);
    const fe_t v1 = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
    const fe_t v2 = fe_sub<S>(fe_mul<S>(fe_sub<S>(fe_dbl<S>(a0), a1), fe_sub<S>(fe_dbl<S>(b0), b1)), fe_sub<S>(fe_dbl<S>(c0), c1));
    s[0] = fe_add<S>(s[0], fe_mul<S>(q.E[x], v0));
    s[1] = fe_add<S>(s[1], fe_mul<S>(q.E[x], v1));
    s[2] = fe_add<S>(s[2], fe_mul<S>(q.E[x], v2));
  }
  for (int k = 0; k < 3; ++k) q.out[p][k] = s[k];
}
static void host_cubic_eval(const fe_t* a, const fe_t* b, const fe_t* cc, const fe_t* E, size_t hn, fe_t sums[3]) {
  HostCubicEval q;
  q.a = a;
  q.b = b;
  q.c = cc;
  q.E = E;
  q.hn = hn;
  const unsigned np = host_parts(hn, 12);
  if (np > 1) sp::WalkPool::get().run(np, host_cubic_part, &q);
  else host_cubic_part(&q, 0, 1);
  for (int k = 0; k < 3; ++k) sums[k] = fe_zero();
  for (unsigned p = 0; p < np; ++p)
    for (int k = 0; k < 3; ++k) sums[k] = fe_add<S>(sums[k], q.out[p][k]);
}
// E(rnd, .) = eq(taus[rnd ..), .) of every round the host may run after a hand-over (tables of <= hand_n entries: rounds rnd >= ell - log2(hand_n) + 1),
// first variable = most significant bit: level k (covering taus[k ..)) is level k + 1 with tau_k in front. Level k sits at offset 2^(ell - k) - 1 ... laid
// out as a pyramid: size-1 level first. Built once, under the device's first rounds (the cubic loop's first wait).
struct HostEqLevels {
  std::vector<fe_t> v;  // level for taus[k ..): m = 2^(ell - k) entries at offset m - 1
  size_t ell = 0, max_m = 0;
  bool built = false;
  void build(const fe_t* taus, size_t ell_, size_t max_entries) {
    ell = ell_;
    max_m = 1;
    while (max_m < max_entries && max_m < ((size_t)1 << ell)) max_m <<= 1;
    v.assign(2 * max_m, fe_zero());
    v[0] = fe_one<S>();
    size_t m = 1;
    for (size_t k = ell; k-- > 0 && m < max_m;) {  // level k from level k + 1
      const fe_t* prev = v.data() + (m - 1);
      fe_t* cur = v.data() + (2 * m - 1);
      for (size_t j = 0; j < m; ++j) {
        const fe_t hi = fe_mul<S>(prev[j], taus[k]);
        cur[j] = fe_sub<S>(prev[j], hi);
        cur[m + j] = hi;
      }
      m *= 2;
    }
    built = true;
  }
  const fe_t* level(size_t k) const {  // taus[k ..): 2^(ell - k) entries, or nullptr when it was not built
    const size_t m = (size_t)1 << (ell - k);
    return built && m <= max_m ? v.data() + (m - 1) : nullptr;
  }
};
// groups > 1 (slot path only): the first nb / groups slots are summed into out_host[0 .. nacc), the next into out_host[nacc .. 2 nacc), ... (the two
// instances of a batched round evaluated by one launch)
static int reduce_partials_wait(sp_ctx* c, int nacc, fe_t* out_host, bool resident = false, unsigned groups = 1, bool wide = false) {
  const int slot_base = wide ? spk::HAND_BASE_ELEM : spk::SLOT_BASE_ELEM;  // wide: up to WIDE_SLOTS slots in the hand-over area (emit_partials_wide)
  const unsigned want = c->result_seq;
  if (c->pending_slots) {  // per-block slots: add them on the host as they become valid
    const unsigned nb = c->pending_slots;
    c->pending_slots = 0;
    const unsigned per_group = nb / groups;
    for (int k = 0; k < nacc * (int)groups; ++k) out_host[k] = fe_zero();
    if (nb == 1) {
      fe_t v[3];
      long spins = 0;
      int rc = wait_slot(c, c->h_pinned + slot_base, want, nacc, v, resident, &spins);
      if (rc) return rc;
      for (int k = 0; k < nacc; ++k) out_host[k] = v[k];
      return SP_OK;
    }
    // Several slots: every pass first reads the tags of all outstanding slots (independent loads: their cache misses — the lines were just written
    // by the device — overlap instead of costing 0.2 us each in turn), then takes the ones that are complete.
    // A slot is two 64-byte lines the device has just written (elements 0, 1 | element 2, tag): both are touched for every outstanding slot before any is
    // read, so the misses of a pass overlap (with the data line fetched only behind its tag a slot cost ~90 ns, 64 of them 6 us behind the last arrival -
    // the resident tail's local regime delivers all its slots at the same moment). The sums are accumulated limb-wise (64 slots: < 2^38 per limb) and
    // reduced once.
    bool done[spk::WIDE_SLOTS] = {};
    uint64_t tags[spk::WIDE_SLOTS], tags2[spk::WIDE_SLOTS];
    uint64_t lazy[2][3][8] = {};  // [group][sum][limb] (groups <= 2)
    const bool lazy_ok = groups <= 2;
    unsigned remaining = nb;
    long passes = 0;
    std::chrono::steady_clock::time_point t0;
    while (remaining) {
      for (unsigned b = 0; b < nb; ++b)
        if (!done[b]) {
          const volatile uint64_t* tg = reinterpret_cast<volatile const uint64_t*>(c->h_pinned + slot_base + 4 * b + 3);
          __builtin_prefetch(reinterpret_cast<const void*>(c->h_pinned + slot_base + 4 * b), 0, 3);
          tags[b] = tg[0];
          tags2[b] = tg[1];
        }
      std::atomic_thread_fence(std::memory_order_acquire);
      for (unsigned b = 0; b < nb; ++b) {
        if (done[b] || (uint32_t)tags[b] != want || (uint32_t)(tags2[b] >> 32) != want) continue;
        const fe_t* slot = c->h_pinned + slot_base + 4 * b;
        fe_t v[3];
        spk::slot_chk chk = {want, want * spk::SLOT_CHK_K};
        for (int k = 0; k < nacc; ++k) {
          for (int i = 0; i < 4; ++i) {
            const uint64_t w = reinterpret_cast<volatile const uint64_t*>(slot + k)[i];
            v[k].v[2 * i] = (uint32_t)w;
            v[k].v[2 * i + 1] = (uint32_t)(w >> 32);
          }
          spk::slot_chk_add(chk, v[k], k);
        }
        if ((uint32_t)(tags[b] >> 32) != chk.a || (uint32_t)tags2[b] != chk.b) continue;  // data still landing: next pass
        if (lazy_ok) {
          uint64_t(*dst)[8] = lazy[b / per_group];
          for (int k = 0; k < nacc; ++k)
            for (int i = 0; i < 8; ++i) dst[k][i] += v[k].v[i];
        } else {
          fe_t* dst = out_host + (size_t)(b / per_group) * nacc;
          for (int k = 0; k < nacc; ++k) dst[k] = fe_add<S>(dst[k], v[k]);
        }
        done[b] = true;
        --remaining;
      }
      if (!remaining) {
        if (lazy_ok)
          for (unsigned g = 0; g < groups; ++g)
            for (int k = 0; k < nacc; ++k) out_host[(size_t)g * nacc + k] = fe_from_limb_sums<S>(lazy[g][k]);
        break;
      }
      if (++passes == 200000) {  // a few ms without completion
        sp::slow_note("reduce_partials_wait (slots)", passes);
        if (!resident) {
          SP_HIP(sp::stream_sync(c->stream));  // e.g. under a profiler
        }
        t0 = std::chrono::steady_clock::now();
      } else if (passes > 200000) {
        if (!resident && passes > 200002) return fail(SP_ERR_INTERNAL, "evaluation kernel did not deliver its block sums");
        if (resident && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(12))
          return fail(SP_ERR_INTERNAL, "sum-check tail kernel did not deliver a round result");
      }
      sp::relax();
    }
    return SP_OK;
  }
  c->pending_slots = 0;
  volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(c->h_pinned + spk::RESULT_FLAG_ELEM);
  bool seen = false;
  for (int spin = 0; spin < 200000; ++spin) {  // ~ a few ms worst case, then fall back to a real synchronise
    if (*flag == want) {
      seen = true;
      break;
    }
    sp::relax();
  }
  if (!seen) sp::slow_note("reduce_partials_wait (flag)", 200000);
  if (!seen && resident) {
    const auto t0 = std::chrono::steady_clock::now();
    while (*flag != want) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(12)) return fail(SP_ERR_INTERNAL, "sum-check tail kernel did not deliver a round result");
      sp::relax();
    }
    seen = true;
  }
  if (!seen) SP_HIP(sp::stream_sync(c->stream));
  std::atomic_thread_fence(std::memory_order_acquire);
  for (int k = 0; k < nacc; ++k) out_host[k] = c->h_pinned[k];
  return SP_OK;
}
static int reduce_partials(sp_ctx* c, size_t nblocks, int nacc, fe_t* out_host) {
  reduce_partials_launch(c, nblocks, nacc);
  return reduce_partials_wait(c, nacc, out_host);
}

// SPARTAN_ROUND_TRACE=1: per-round host timeline of the sum-check loops on stderr (wait = result not yet visible, host = finish + transcript)
static bool round_trace() {
  static const bool on = [] {
    const char* e = getenv("SPARTAN_ROUND_TRACE");
    return e && e[0] == '1';
  }();
  return on;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- persistent tail (kernels_poly.hpp k_sumcheck_tail): host half of the mailbox ------------------------------------------------------
// the resident tail takes a sum-check over from tables of 2^16 elements down (64 blocks: 256 pairs a block in its first step, HOST_SUM_MAX_BLOCKS
// result slots; sweeps of 2^13 .. 2^16 measured within 5 us of each other, tools/tail_sweep.sh)
static const size_t TAIL_MAX_LEN = [] {
  const char* e = getenv("SPARTAN_TAIL_LOG2");  // A/B: table length from which the resident tail takes over (13 .. 16)
  const int k = e ? atoi(e) : 16;  // 2^16 since the local regime (round 5): 0.8675 -> 0.8641 ms against 2^15 on one box, outer sum-check -5.6 us
  return (size_t)1 << (k < 10 ? 10 : (k > 16 ? 16 : k));
}();
static bool tail_enabled() { return true; }
// mailbox line (64-byte aligned, one PCIe read for the device): words 0..7 = challenge, 8 = sequence number it answers, 9 / 10 = two independent
// check words (sequence + plain sum, sequence * K + position-weighted sum), 11 = the sequence number again, so a poll that straddles the host's
// stores is recognised and retried (mail_wait in kernels_poly.hpp).
static void mail_write_line(volatile uint32_t* dst, const fe_t& r, unsigned answers_seq) {
  uint32_t chk = answers_seq, chk2 = answers_seq * spk::SLOT_CHK_K;
  for (int i = 0; i < 8; ++i) {
    dst[i] = r.v[i];
    chk += r.v[i];
    chk2 += (uint32_t)(i + 1) * r.v[i];
  }
  dst[9] = chk;
  dst[10] = chk2;
  dst[11] = answers_seq;
  std::atomic_thread_fence(std::memory_order_release);
  dst[8] = answers_seq;
}
static void tail_post_challenge(sp_ctx* c, const fe_t& r, unsigned answers_seq) {
  const size_t off = (size_t)spk::MAIL_LINE_WORDS * (answers_seq & (spk::MAIL_RING - 1));
  if (c->h_mail_mirror) mail_write_line(c->h_mail_mirror + off, r, answers_seq);  // host memory first: it is the fallback of the line below
  mail_write_line(c->h_mail + off, r, answers_seq);
  if (c->mail_dev) __builtin_ia32_sfence();  // BAR memory is write-combining: push the line out now
}
// Error exits of a round loop: kernels issued ahead of their challenge (and the resident tail) are still waiting at the mailbox and would hold
// their CUs for the 8 s watchdog, then trip the sticky error word under the next, unrelated sum-check. The abort word of every mailbox line makes
// them leave at once; the stream is drained and the words are cleared so that the next sum-check on this context starts clean.
static void tail_abort(sp_ctx* c) {
  auto set_all = [&](uint32_t v) {
    for (int l = 0; l < spk::MAIL_RING; ++l) {
      if (c->h_mail_mirror) c->h_mail_mirror[spk::MAIL_LINE_WORDS * l + 12] = v;
      c->h_mail[spk::MAIL_LINE_WORDS * l + 12] = v;
    }
    if (c->mail_dev) __builtin_ia32_sfence();
  };
  set_all(1);
  sp::stream_sync(c->stream);
  set_all(0);
  *reinterpret_cast<volatile uint32_t*>(c->h_pinned + spk::TAIL_ERR_ELEM) = 0;
}
// armed while a launch issued ahead of its challenge (or the resident tail) waits at the mailbox; an early return aborts it
struct AheadGuard {
  sp_ctx* c;
  bool armed = false;
  explicit AheadGuard(sp_ctx* ctx) : c(ctx) {}
  ~AheadGuard() {
    if (armed) tail_abort(c);
  }
};
static spk::MailRef mail_ref(sp_ctx* c, bool ahead, unsigned answers) {
  spk::MailRef m;
  m.mail = ahead ? c->d_mail : nullptr;
  m.mirror = c->d_mail_mirror;
  m.mapped = c->d_pinned;
  m.answers = answers;
  m.gated = nullptr;
  return m;
}
// A launch too large to wait at the mailbox itself (launch_ahead_ok: a grid that fills the chip while it polls starves every other stream) is queued
// AHEAD of its challenge all the same, behind a one-wave gate: k_mail_gate waits at the mailbox and leaves the challenge in device memory, the launch
// behind it in the stream starts when the gate ends and reads it there. The challenge then reaches the kernel through the command processor's
// in-queue dependency (2-3 us) instead of a host launch behind the challenge (two launch calls of 3 us each on the host's critical path + the
// dispatch latency), and nothing but one wave waits. Returns the reference the gated launch takes (mail = nullptr: it does not poll).
static spk::MailRef gate_launch(sp_ctx* c, unsigned answers) {
  fe_t* slot = c->d_gate + (answers & (spk::MAIL_RING - 1));
  hipLaunchKernelGGL(spk::k_mail_gate, dim3(1), dim3(64), 0, c->stream, mail_ref(c, true, answers), slot);
  spk::MailRef m = mail_ref(c, false, answers);
  m.gated = slot;
  return m;
}
static bool gate_ok(sp_ctx* c) {
  static const bool off = [] {
    const char* e = getenv("SPARTAN_GATE");  // "0": streaming launches behind their challenge, as before (A/B)
    return e && e[0] == '0';
  }();
  return c->mail_dev && !off;
}
// a fused bind+evaluate launch may be issued AHEAD of its challenge (the kernel waits at the mailbox) when the mailbox is in device memory.
// The largest tables are left alone: their kernels are the bandwidth-bound ones whose durations the roofline is measured on.
static bool launch_ahead_ok(sp_ctx* c, size_t table_len) {
  return c->mail_dev && table_len <= ((size_t)1 << 19);
}
// result slots (= resident blocks still active) of the evaluation over a table of `len` elements
// (cubic: at most HOST_SUM_MAX_BLOCKS blocks - a table of twice the pairs they take at TAIL_WIDE_Q_CUBIC each gives every block a first step of twice
// the share, two passes of its bind phase; the kernel's LDS is sized for that)
static unsigned tail_blocks(size_t len, bool cubic = false) {
  const size_t q = len / 2, wq = cubic ? spk::TAIL_WIDE_Q_CUBIC : spk::TAIL_WIDE_Q;
  if (q <= wq) return 1u;
  const size_t nb = q / wq;
  return (unsigned)(cubic && nb == 2 * (size_t)spk::HOST_SUM_MAX_BLOCKS ? nb / 2 : nb);
}
// Resident blocks occupy their CU (1024 threads) until the host has driven every round. If the tails of several contexts together asked for
// more blocks than the chip holds, each could sit on CUs the other needs for blocks its host is waiting for: a multi-block tail therefore
// leases its blocks from a process-wide budget and, when the budget is short, the sum-check simply keeps launching ordinary rounds until
// the tail it needs is small enough. Single-block tails (one per context) are not counted.
// The budget is a QUARTER of the chip - one full-size tail at a time - (SPARTAN_TAIL_BUDGET overrides it): a resident block needs the whole register file of its CU, so it can only be placed
// on a CU that has drained completely, and with eight contexts in flight - each with thousand-block kernels queued whose blocks fill any slot that
// frees up - the not-yet-placed blocks of a tail starved behind them while its placed blocks held their CUs (stress: one 8 - 12 s stall per ~3000
// proofs with a budget of 256, rarer but not gone with 128; a lone proof's 64-block tails are unaffected).
static std::atomic<int> g_tail_resident{0};
static int tail_block_budget() {
  static const int v = [] {
    const char* e = getenv("SPARTAN_TAIL_BUDGET");
    const int b = e ? atoi(e) : 64;
    return b < 0 ? 0 : b;
  }();
  return v;
}
struct TailLease {
  int n = 0;
  bool take(unsigned blocks) {
    if (blocks <= 1) return true;
    if (g_tail_resident.fetch_add((int)blocks) + (int)blocks > tail_block_budget()) {
      g_tail_resident.fetch_sub((int)blocks);
      return false;
    }
    n = (int)blocks;
    return true;
  }
  ~TailLease() {
    if (n) g_tail_resident.fetch_sub(n);
  }
};
static int tail_check(sp_ctx* c) {
  volatile uint32_t* err = reinterpret_cast<volatile uint32_t*>(c->h_pinned + spk::TAIL_ERR_ELEM);
  if (*err) {
    char buf[320];
    const volatile uint32_t* seen = reinterpret_cast<volatile uint32_t*>(c->h_pinned + spk::TAIL_ERR_ELEM + 1);
    snprintf(buf, sizeof buf,
             "a sum-check kernel timed out waiting for its challenge (wanted seq %u, block %u of %u; mailbox as last polled: seq %u / %u, checks %08x %08x, abort %u; "
             "host result_seq %u)",
             err[1], err[2], err[3], seen[8], seen[11], seen[9], seen[10], seen[12], c->result_seq);
    *err = 0;
    return fail(SP_ERR_INTERNAL, buf);
  }
  return SP_OK;
}

// ---- host-side O(1) glue: UniPoly (src/polys/univariate.rs) ----------------------------------------------------------
namespace {
struct UniPoly {
  fe_t c[4];
  int n;
};
fe_t two_inv() {
  static fe_t v = fe_inv_vartime<S>(fe_from_u64<S>(2));
  return v;
}
fe_t six_inv() {
  static fe_t v = fe_inv_vartime<S>(fe_from_u64<S>(6));
  return v;
}
UniPoly from_evals_deg2(const fe_t e[3]) {  // univariate.rs:84-93
  UniPoly p;
  p.n = 3;
  fe_t c0 = e[0];
  fe_t a = fe_mul<S>(fe_add<S>(fe_sub<S>(e[0], fe_dbl<S>(e[1])), e[2]), two_inv());
  fe_t b = fe_sub<S>(fe_sub<S>(e[1], c0), a);
  p.c[0] = c0;
  p.c[1] = b;
  p.c[2] = a;
  return p;
}
UniPoly from_evals_deg3(const fe_t e[4]) {  // univariate.rs:102-118
  UniPoly p;
  p.n = 4;
  fe_t d = e[0];
  fe_t e1_3 = fe_add<S>(fe_dbl<S>(e[1]), e[1]), e2_3 = fe_add<S>(fe_dbl<S>(e[2]), e[2]);
  fe_t delta3 = fe_sub<S>(fe_add<S>(fe_sub<S>(e[3], e2_3), e1_3), e[0]);
  fe_t a = fe_mul<S>(delta3, six_inv());
  fe_t delta2 = fe_add<S>(fe_sub<S>(e[2], fe_dbl<S>(e[1])), e[0]);
  fe_t b = fe_sub<S>(fe_mul<S>(delta2, two_inv()), fe_add<S>(fe_dbl<S>(a), a));
  fe_t c1 = fe_sub<S>(fe_sub<S>(fe_sub<S>(e[1], d), b), a);
  p.c[0] = d;
  p.c[1] = c1;
  p.c[2] = b;
  p.c[3] = a;
  return p;
}
fe_t poly_eval(const UniPoly& p, const fe_t& r) {  // univariate.rs:136-144
  fe_t ev = p.c[0], pw = r;
  for (int i = 1; i < p.n; ++i) {
    ev = fe_add<S>(ev, fe_mul<S>(pw, p.c[i]));
    pw = fe_mul<S>(pw, r);
  }
  return ev;
}
// absorb(b"p", &poly): compressed coefficients, to_repr LE each (univariate.rs:182-190)
void absorb_poly(sp::Transcript& t, const UniPoly& p) {
  uint8_t buf[32 * 3];
  int k = 0;
  sp::fe_to_le_bytes<S>(p.c[0], buf);
  k = 1;
  for (int i = 2; i < p.n; ++i) sp::fe_to_le_bytes<S>(p.c[i], buf + 32 * k++);
  const uint8_t lbl[1] = {'p'};
  t.absorb(lbl, 1, buf, 32 * k);
}
}  // namespace

extern "C" {

int sp_table_bind_top(sp_ctx* c, sp_table* t, const uint64_t r[4]) {
  sp_table* tabs[1] = {t};
  return launch_bind(c, tabs, 1, load_fe(r));
}

int sp_eq_table(sp_ctx* c, const uint64_t* r, size_t ell, sp_table** out) {
  if (ell > 30) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_eq_table: ell too large");
  sp_table* t;
  int rc = sp::alloc_table(c, (size_t)1 << ell, &t);
  if (rc) return rc;
  rc = sp_eq_table_into(c, r, ell, t);
  if (rc) {
    sp_table_free(t);
    return rc;
  }
  *out = t;
  return SP_OK;
}

int sp_eq_table_into(sp_ctx* c, const uint64_t* r, size_t ell, sp_table* t) {
  if (ell > 30) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_eq_table: ell too large");
  size_t total = (size_t)1 << ell;
  if (t->cap < total) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_eq_table_into: table too short");
  if (c->eq_ahead_ell == ell && ell != 0 && memcmp(c->eq_ahead_r, r, c->eq_ahead_known * sizeof(fe_t)) == 0)
    return sp_eq_table_finish(c, r, ell, t);  // the pyramids of this point's first ell - 2 coordinates were started under a sum-check
  t->len = total;
  t->lo_eff = t->hi_eff = (size_t)-1;
  int rc;
  // split the variables: low part <= 10 bits built by one block, high part likewise (ell <= 20), else recursive outer products
  int lo_bits = (int)(ell > 10 ? 10 : ell), hi_bits = (int)ell - lo_bits;
  if (hi_bits > 10) {  // > 2^20 entries: build the low 2^20 by outer product first, then widen once more
    lo_bits = (int)ell - 10;
    hi_bits = 10;
  }
  // scratch: r (ell), level pyramids
  size_t need = ell + 2 * (((size_t)1 << 11)) + 64 + (lo_bits > 10 ? ((size_t)1 << lo_bits) : 0);
  rc = c->ensure_scratch(need + 8);
  if (rc) return rc;
  fe_t* d_r = c->d_scratch;
  fe_t* d_hi = d_r + ell;
  fe_t* d_lo = d_hi + ((size_t)1 << 11);
  fe_t* d_lowbig = d_lo + ((size_t)1 << 11);
  if (ell > 10 && lo_bits <= 10) {
    // 2^11 .. 2^20 entries (evals_rx of the prover): both half pyramids in one launch with the challenges by value, then the outer product —
    // no upload of r, so nothing to wait for: the caller's next launches queue right behind
    spk::EqPairArgs ea;
    for (int i = 0; i < hi_bits; ++i) ea.v[0][i] = load_fe(r + 4 * i);
    for (int i = 0; i < lo_bits; ++i) ea.v[1][i] = load_fe(r + 4 * (hi_bits + i));
    ea.m[0] = hi_bits;
    ea.m[1] = lo_bits;
    ea.out[0] = d_hi;
    ea.out[1] = d_lo;
    hipLaunchKernelGGL(spk::k_eq_levels_pair, dim3(2), dim3(1024), 0, c->stream, ea);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    c->timed("eq_table", 32ull * total, [&] {
      hipLaunchKernelGGL(spk::k_eq_outer, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_hi + spk::eq_level_offset(hi_bits),
                         d_lo + spk::eq_level_offset(lo_bits), lo_bits, total, t->d);
    });
    return SP_OK;
  }
  if (ell) SP_HIP(hipMemcpyAsync(d_r, r, ell * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  if (ell <= 10) {
    hipLaunchKernelGGL(spk::k_eq_levels, dim3(1), dim3(1024), 0, c->stream, d_r, (int)ell, d_lo);
    SP_HIP(hipMemcpyAsync(t->d, d_lo + spk::eq_level_offset((int)ell), total * sizeof(fe_t), hipMemcpyDeviceToDevice, c->stream));
  } else if (lo_bits <= 10) {
    hipLaunchKernelGGL(spk::k_eq_levels, dim3(1), dim3(1024), 0, c->stream, d_r, hi_bits, d_hi);
    hipLaunchKernelGGL(spk::k_eq_levels, dim3(1), dim3(1024), 0, c->stream, d_r + hi_bits, lo_bits, d_lo);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    c->timed("eq_table", 32ull * total, [&] {
      hipLaunchKernelGGL(spk::k_eq_outer, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_hi + spk::eq_level_offset(hi_bits),
                         d_lo + spk::eq_level_offset(lo_bits), lo_bits, total, t->d);
    });
  } else {
    // ell in (20, 30]: low (ell-10) bits = outer product of two <=10-bit tables, then one more outer product
    int l2 = lo_bits - 10;  // in (0, 10]
    hipLaunchKernelGGL(spk::k_eq_levels, dim3(1), dim3(1024), 0, c->stream, d_r + hi_bits, l2, d_hi);
    hipLaunchKernelGGL(spk::k_eq_levels, dim3(1), dim3(1024), 0, c->stream, d_r + hi_bits + l2, 10, d_lo);
    size_t lowtotal = (size_t)1 << lo_bits;
    hipLaunchKernelGGL(spk::k_eq_outer, dim3(4096), dim3(256), 0, c->stream, d_hi + spk::eq_level_offset(l2), d_lo + spk::eq_level_offset(10), 10,
                       lowtotal, d_lowbig);
    hipLaunchKernelGGL(spk::k_eq_levels, dim3(1), dim3(1024), 0, c->stream, d_r, hi_bits, d_hi);
    hipLaunchKernelGGL(spk::k_eq_outer, dim3(8192), dim3(256), 0, c->stream, d_hi + spk::eq_level_offset(hi_bits), d_lowbig, lo_bits, total, t->d);
  }
  SP_HIP(sp::stream_sync(c->stream));  // r is a borrowed host buffer
  return SP_OK;
}

// EqPolynomial::evals_from_points for a point that a sum-check is still drawing (evals_rx, src/spartan.rs:316): `_begin`, called when all but the last
// K coordinates exist (2 <= K <= 4), builds the two pyramids on a stream of its own under the last K rounds; `_finish` (all coordinates known) is then
// ONE launch, the outer product with the last K variables applied in it - instead of pyramids (12-14 us behind a launch gap) + outer product behind the
// last challenge. K = 4 is the moment the sum-check's resident kernel hands its last rounds to the host (kernels_poly.hpp tail_hand_over): the
// pyramids then have those rounds' whole host time. 12 <= ell <= 20. `_finish` checks that the known prefix has not changed and falls back to
// sp_eq_table_into when there was no `_begin`.
int sp_eq_table_begin(sp_ctx* c, const uint64_t* r_known, size_t n_known, size_t ell) {
  if (ell < 12 || ell > 20 || n_known + 2 > ell || n_known + 4 < ell)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_eq_table_begin: 12 <= ell <= 20 and all but the last two to four coordinates known");
  const int K = (int)(ell - n_known), lo_bits = 10, hi_bits = (int)ell - lo_bits;
  spk::EqPairArgs ea;
  for (int i = 0; i < hi_bits; ++i) ea.v[0][i] = load_fe(r_known + 4 * i);
  for (int i = 0; i < lo_bits - K; ++i) ea.v[1][i] = load_fe(r_known + 4 * (hi_bits + i));
  ea.m[0] = hi_bits;
  ea.m[1] = lo_bits - K;
  ea.out[0] = c->d_eq_ahead;
  ea.out[1] = c->d_eq_ahead + ((size_t)1 << 11);
  if (c->eq_read_pending) {  // an outer product of an earlier _finish may still be reading the buffer on the main stream: the streams are not otherwise ordered
    SP_HIP(hipStreamWaitEvent(c->stream_eq, c->eq_read_ev, 0));
    c->eq_read_pending = false;
  }
  hipLaunchKernelGGL(spk::k_eq_levels_pair, dim3(2), dim3(1024), 0, c->stream_eq, ea);
  SP_HIP(hipEventRecord(c->eq_ev, c->stream_eq));
  memcpy(c->eq_ahead_r, r_known, n_known * sizeof(fe_t));
  c->eq_ahead_ell = ell;
  c->eq_ahead_known = n_known;
  return SP_OK;
}
int sp_eq_table_finish(sp_ctx* c, const uint64_t* r, size_t ell, sp_table* t) {
  const bool begun = c->eq_ahead_ell == ell && ell != 0 && memcmp(c->eq_ahead_r, r, c->eq_ahead_known * sizeof(fe_t)) == 0;
  c->eq_ahead_ell = 0;
  if (!begun) return sp_eq_table_into(c, r, ell, t);
  const size_t total = (size_t)1 << ell;
  if (t->cap < total) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_eq_table_finish: table too short");
  t->len = total;
  t->lo_eff = t->hi_eff = (size_t)-1;
  const int lo_bits = 10, hi_bits = (int)ell - lo_bits, K = (int)(ell - c->eq_ahead_known);
  SP_HIP(hipStreamWaitEvent(c->stream, c->eq_ev, 0));
  const size_t n_hi = (size_t)1 << hi_bits;
  spk::EqLastK wk;  // the 2^K weights of the last K coordinates, formed here (host products) level by level
  {
    const fe_t one = fe_one<S>();
    wk.w[0] = one;
    for (int i = 0; i < K; ++i) {
      const fe_t r_i = load_fe(r + 4 * (ell - K + i));
      for (int j = (1 << i) - 1; j >= 0; --j) {  // e[2 j + 1] = e[j] r_i, e[2 j] = e[j] - e[2 j + 1]
        const fe_t y = i == 0 ? r_i : fe_mul<S>(wk.w[j], r_i);
        wk.w[2 * j] = fe_sub<S>(wk.w[j], y);
        wk.w[2 * j + 1] = y;
      }
    }
    for (int j = 1 << K; j < 16; ++j) wk.w[j] = fe_zero();
  }
  const unsigned blocks = (unsigned)((n_hi + spk::EQ_LASTK_HPB - 1) / spk::EQ_LASTK_HPB) * (1024 / spk::EQ_LASTK_BLOCK);
  c->timed("eq_table", 32ull * total, [&] {
    hipLaunchKernelGGL(spk::k_eq_outer_lastk, dim3(blocks), dim3(spk::EQ_LASTK_BLOCK), 0, c->stream, c->d_eq_ahead + spk::eq_level_offset(hi_bits),
                       c->d_eq_ahead + ((size_t)1 << 11) + spk::eq_level_offset(lo_bits - K), K, n_hi, wk, t->d);
  });
  SP_HIP(hipEventRecord(c->eq_read_ev, c->stream));
  c->eq_read_pending = true;
  return SP_OK;
}

// ---- transcript ---------------------------------------------------------------------------------------------------
int sp_transcript_new(sp_ctx*, const uint8_t* label, size_t n, sp_transcript** out) {
  sp_transcript* t = new sp_transcript();
  t->t.init(label, n);
  *out = t;
  return SP_OK;
}
// one hashing thread per process for the long absorbs (see sp_transcript): taken by whichever transcript asks first, everyone else hashes inline
static sp::Worker g_hash_worker;
static std::atomic<bool> g_hash_taken{false};
static size_t async_absorb_min() { return 4096; }  // bytes
int sp_transcript_absorb(sp_transcript* t, const uint8_t* label, size_t ln, const uint8_t* bytes, size_t n) {
  t->join();
  const size_t amin = async_absorb_min();
  if (t->async_absorb && amin && n >= amin && !g_hash_taken.exchange(true, std::memory_order_acq_rel)) {
    auto job = std::make_shared<sp_absorb_job>();
    job->data.resize(ln + n);
    memcpy(job->data.data(), label, ln);
    memcpy(job->data.data() + ln, bytes, n);
    job->label = ln;
    job->h = t->t.h;
    t->pend = job;
    // a caller that hands over one long absorb usually has more work of the kind (the per-instance prefix of a prove, then the commitment, then the next
    // prove's): the thread polls for 1.5 ms before it sleeps again, so the next job is claimed at once instead of after a wake-up (5-50 us)
    g_hash_worker.keep_hot(1500);
    g_hash_worker.submit([job] {
      int expect = 1;
      if (job->state.compare_exchange_strong(expect, 2, std::memory_order_acq_rel)) {
        job->run();
        job->state.store(4, std::memory_order_release);
      }  // (else the caller took the job back)
      g_hash_taken.store(false, std::memory_order_release);
    });
    return SP_OK;
  }
  t->t.absorb(label, ln, bytes, n);
  return SP_OK;
}
int sp_transcript_set_async(sp_transcript* t, int on) {
  if (!t) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_transcript_set_async: null transcript");
  t->join();
  t->async_absorb = on != 0;
  return SP_OK;
}
int sp_transcript_preabsorb(const uint8_t* label, size_t ln, const uint8_t* bytes, size_t n, sp_absorb_state** out) {
  if (!out || (!label && ln) || (!bytes && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_transcript_preabsorb: null argument");
  sp_absorb_state* s = new sp_absorb_state;
  s->h.init();
  s->h.update(label, ln);
  s->h.update(bytes, n);
  *out = s;
  return SP_OK;
}
int sp_transcript_absorb_prepared(sp_transcript* t, const sp_absorb_state* s) {
  if (!t || !s) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_transcript_absorb_prepared: null argument");
  t->join();
  bool fresh = t->t.h.fill == 0;
  for (int i = 0; i < 25 && fresh; ++i) fresh = t->t.h.a[i] == 0;
  if (!fresh) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "sp_transcript_absorb_prepared: the transcript has absorbed input since its last squeeze");
  t->t.h = s->h;
  return SP_OK;
}
void sp_absorb_state_free(sp_absorb_state* s) { delete s; }
int sp_transcript_dom_sep(sp_transcript* t, const uint8_t* bytes, size_t n) {
  t->join();
  t->t.dom_sep(bytes, n);
  return SP_OK;
}
int sp_transcript_squeeze(sp_transcript* t, const uint8_t* label, size_t ln, uint64_t out[4]) {
  t->join();
  fe_t f;
  if (!t->t.squeeze<S>(label, ln, &f)) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
  store_fe(out, f);
  return SP_OK;
}
int sp_transcript_clone(const sp_transcript* t, sp_transcript** out) {
  t->join();
  sp_transcript* n = new sp_transcript();
  n->t = t->t;
  n->async_absorb = t->async_absorb;
  *out = n;
  return SP_OK;
}
void sp_transcript_free(sp_transcript* t) { delete t; }

// ---- sum-check ------------------------------------------------------------------------------------------------------
int sp_table_dot(sp_ctx* c, const sp_table* a, const sp_table* b, size_t n, uint64_t out[4]) {
  if (n > a->cap || n > b->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_table_dot: n exceeds a table");
  size_t blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  if (blocks == 0) blocks = 1;
  int rc = c->ensure_scratch(blocks + 16);
  if (rc) return rc;
  c->timed("dot", 64ull * n, [&] { hipLaunchKernelGGL(spk::k_dot, dim3((unsigned)blocks), dim3(256), 0, c->stream, a->d, b->d, n, c->d_scratch, c->d_pinned, next_seq(c)); });
  fe_t r;
  rc = reduce_partials(c, blocks, 1, &r);
  if (rc) return rc;
  store_fe(out, r);
  return SP_OK;
}

// Fused launches with at least this many pairs per table quarter take the streaming kernels (>= 2^20-entry tables: 64 MiB+ per launch,
// beyond the aggregate L2; at config 2 that is the first fused launch of each sum-check).
static const size_t STREAM_MIN_Q = (size_t)1 << 18;

int sp_eval_cubic_outer_pow(sp_ctx* c, const sp_table* pl, const sp_table* pr, const sp_table* A, const sp_table* B, const sp_table* C, uint64_t out[12]) {
  if (A->len != B->len || A->len != C->len || A->len < 2) return fail(SP_ERR_INVALID_INPUT_LENGTH, "eval_cubic_outer_pow: tables must have equal even length");
  const size_t len = A->len / 2, left = pl->len;
  const bool fallback = len < left;
  size_t right = 0;
  if (fallback) {
    if (left < 2 * len) return fail(SP_ERR_INVALID_INPUT_LENGTH, "eval_cubic_outer_pow: fallback needs a pow table at least as long as A");
  } else {
    if (left == 0 || len % left) return fail(SP_ERR_INVALID_INPUT_LENGTH, "eval_cubic_outer_pow: len must be a multiple of the left table");
    right = len / left;
    if (pr->len < 2 * right) return fail(SP_ERR_INVALID_INPUT_LENGTH, "eval_cubic_outer_pow: right table must have at least 2 * len / left entries");
  }
  size_t blocks = (len + 255) / 256;
  int rc = c->ensure_scratch(blocks * 3 + 32);
  if (rc) return rc;
  const unsigned seq = next_seq(c);
  c->timed("eval_cubic_pow", 224ull * len, [&] {
    if (fallback)
      hipLaunchKernelGGL((spk::k_eval_cubic_outer_pow<true>), dim3((unsigned)blocks), dim3(256), 0, c->stream, pl->d, left, pr ? pr->d : nullptr, right, A->d, B->d,
                         C->d, len, c->d_scratch, c->d_pinned, seq);
    else
      hipLaunchKernelGGL((spk::k_eval_cubic_outer_pow<false>), dim3((unsigned)blocks), dim3(256), 0, c->stream, pl->d, left, pr->d, right, A->d, B->d, C->d, len,
                         c->d_scratch, c->d_pinned, seq);
  });
  fe_t sums[3];
  rc = reduce_partials(c, blocks, 3, sums);
  if (rc) return rc;
  memcpy(out, sums, 96);
  return SP_OK;
}

// evaluation_points_zero_check_round0 (src/sumcheck.rs:1163-1271)
int sp_eval_cubic_zero_check_round0(sp_ctx* c, const uint64_t* taus_, size_t ell, const sp_table* A, const sp_table* B, uint64_t out[12]) {
  const size_t N = (size_t)1 << ell;
  if (ell == 0 || A->len != N || B->len != N) return fail(SP_ERR_INVALID_INPUT_LENGTH, "zero_check_round0: tables must have 2^ell elements");
  const size_t first_half = ell / 2, second_half = ell - first_half;
  const size_t nleft = first_half > 0 ? first_half - 1 : 0;
  if (nleft > 16 || second_half > 16) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sum-check over more than 2^32 rows");
  const size_t pyr_left = (size_t)2 << nleft, pyr_right = (size_t)2 << second_half;
  const size_t chunk = 256 * spk::EVAL_PPT, half = N / 2, blocks = (half + chunk - 1) / chunk;
  int rc = c->ensure_scratch(blocks * 3 + pyr_left + pyr_right + 64);
  if (rc) return rc;
  fe_t* d_part = c->d_scratch;
  fe_t* d_pl = d_part + blocks * 3;
  fe_t* d_pr = d_pl + pyr_left;
  std::vector<fe_t> taus(ell);
  for (size_t i = 0; i < ell; ++i) taus[i] = load_fe(taus_ + 4 * i);
  {
    spk::EqPairArgs ea;
    for (size_t i = 0; i < nleft; ++i) ea.v[0][i] = taus[1 + i];
    for (size_t i = 0; i < second_half; ++i) ea.v[1][i] = taus[first_half + i];
    ea.m[0] = (int)nleft;
    ea.m[1] = (int)second_half;
    ea.out[0] = d_pl;
    ea.out[1] = d_pr;
    hipLaunchKernelGGL(spk::k_eq_levels_pair, dim3(2), dim3(1024), 0, c->stream, ea);
  }
  // round 1 tables (poly_eqs_first_half / poly_eq_right_last_half, :1407-1428)
  const fe_t *eq_in, *eq_out = nullptr;
  int sbits, mode;
  if (1 < first_half) {
    eq_out = d_pl + spk::eq_level_offset((int)(first_half - 1));
    eq_in = d_pr + spk::eq_level_offset((int)second_half);
    sbits = (int)second_half;
    mode = (((size_t)1 << sbits) >= chunk) ? 1 : 2;
  } else {
    eq_in = d_pr + spk::eq_level_offset((int)(ell - 1));
    sbits = 63;
    mode = 0;
  }
  const unsigned seq = next_seq(c);
  const dim3 g((unsigned)blocks), b(256);
  c->timed("eval_cubic", 128ull * half, [&] {
    if (mode == 0) hipLaunchKernelGGL((spk::k_eval_cubic<0, false, true>), g, b, 0, c->stream, A->d, B->d, (const fe_t*)nullptr, half, eq_in, eq_out, sbits, d_part, c->d_pinned, seq);
    else if (mode == 1) hipLaunchKernelGGL((spk::k_eval_cubic<1, false, true>), g, b, 0, c->stream, A->d, B->d, (const fe_t*)nullptr, half, eq_in, eq_out, sbits, d_part, c->d_pinned, seq);
    else hipLaunchKernelGGL((spk::k_eval_cubic<2, false, true>), g, b, 0, c->stream, A->d, B->d, (const fe_t*)nullptr, half, eq_in, eq_out, sbits, d_part, c->d_pinned, seq);
  });
  fe_t sums[2];
  if ((rc = reduce_partials(c, blocks, 2, sums))) return rc;
  const fe_t tinf = sums[1], one = fe_one<S>(), tau = taus[0];
  const fe_t eq0 = fe_sub<S>(one, tau), slope = fe_sub<S>(tau, eq0), eqm1 = fe_sub<S>(eq0, slope);
  // t(0) = 0 and the claim is 0, p = 1: s(0) = s(1) = 0, s_leading = slope * t_inf; t(-1) = 2 t_inf + 2 t(0) - t(1) = 2 t_inf in both branches
  // (derive_from_claim with tau != 0: t(1) = s(1) / (tau p) = 0; the tau = 0 fallback states it directly, :1244-1268)
  const fe_t s_0 = fe_zero(), s_1 = fe_zero();
  const fe_t s_leading = fe_mul<S>(slope, tinf);
  const fe_t s_m1 = fe_mul<S>(eqm1, fe_dbl<S>(tinf));
  const fe_t halfc = two_inv();
  const fe_t c1 = fe_sub<S>(fe_mul<S>(fe_sub<S>(s_1, s_m1), halfc), s_leading);
  const fe_t c2 = fe_sub<S>(fe_mul<S>(fe_add<S>(s_1, s_m1), halfc), s_0);
  const fe_t inner_2 = fe_add<S>(c2, fe_dbl<S>(s_leading));
  const fe_t eval_2 = fe_add<S>(s_0, fe_dbl<S>(fe_add<S>(c1, fe_dbl<S>(inner_2))));
  const fe_t c3_3 = fe_add<S>(fe_dbl<S>(s_leading), s_leading);
  const fe_t inner_3 = fe_add<S>(c2, c3_3);
  const fe_t mid_3 = fe_add<S>(fe_add<S>(c1, fe_dbl<S>(inner_3)), inner_3);
  const fe_t eval_3 = fe_add<S>(fe_add<S>(s_0, fe_dbl<S>(mid_3)), mid_3);
  store_fe(out, s_0);
  store_fe(out + 4, eval_2);
  store_fe(out + 8, eval_3);
  return SP_OK;
}

static bool table_dense(const sp_table* t) { return sp::eff_lo(t) == t->len / 2 && sp::eff_hi(t) == t->len / 2; }

static int quad_impl(sp_ctx* c, uint64_t claim_io[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_reduce_hook reduce, void* reduce_user,
                     sp_challenge_hook observe, void* observe_user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8], size_t run_rounds = 0);
int sp_sumcheck_quad(sp_ctx* c, const uint64_t claim_[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, uint64_t* out_cpolys,
                     uint64_t* out_r, uint64_t out_final[8]) {
  uint64_t claim_io[4];
  memcpy(claim_io, claim_, 32);
  // an opening announced on this context (sp_hyrax_prove_announce) learns the row challenges here, ten rounds before PCS::prove is called
  return quad_impl(c, claim_io, rounds, A, B, tr, nullptr, nullptr, sp::pcs_ahead_wants(c, rounds) ? &sp::pcs_ahead_on_challenge : nullptr, c, out_cpolys, out_r, out_final);
}
int sp_sumcheck_quad_observed(sp_ctx* c, const uint64_t claim_[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_challenge_hook observe,
                              void* user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8]) {
  uint64_t claim_io[4];
  memcpy(claim_io, claim_, 32);
  return quad_impl(c, claim_io, rounds, A, B, tr, nullptr, nullptr, observe, user, out_cpolys, out_r, out_final);
}
int sp_sumcheck_quad_sharded(sp_ctx* c, uint64_t claim_io[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_reduce_hook reduce, void* reduce_user,
                             uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8]) {
  return quad_impl(c, claim_io, rounds, A, B, tr, reduce, reduce_user, nullptr, nullptr, out_cpolys, out_r, out_final);
}

int sp_sumcheck_quad_sharded_observed(sp_ctx* c, uint64_t claim_io[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_reduce_hook reduce,
                                      void* reduce_user, sp_challenge_hook observe, void* observe_user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8]) {
  return quad_impl(c, claim_io, rounds, A, B, tr, reduce, reduce_user, observe, observe_user, out_cpolys, out_r, out_final);
}

int sp_sumcheck_quad_sharded_partial(sp_ctx* c, uint64_t claim_io[4], size_t rounds, size_t run_rounds, sp_table* A, sp_table* B, sp_transcript* tr,
                                     sp_reduce_hook reduce, void* reduce_user, sp_challenge_hook observe, void* observe_user, uint64_t* out_cpolys, uint64_t* out_r) {
  if (run_rounds == 0 || run_rounds >= rounds) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_quad (partial): 0 < run_rounds < rounds");
  uint64_t unused[8];
  return quad_impl(c, claim_io, rounds, A, B, tr, reduce, reduce_user, observe, observe_user, out_cpolys, out_r, unused, run_rounds);
}

// prove_quad on a slice of the tables (see sp_sumcheck_cubic3_sharded): per round the slice's (eval0, t_inf) are combined across ranks by `reduce`
// run_rounds (0 = all): stop after that many rounds - the tables are left bound to 2^(rounds - run_rounds) elements, the running claim goes out
// through claim_io and out_final is not written. A sharded prover runs the rounds whose tables are large (where sharding pays) on its slice and
// gathers the slices for the rest instead of exchanging sums in every remaining round. The resident tail is not used in a stopped call.
static int quad_impl(sp_ctx* c, uint64_t claim_io[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_reduce_hook reduce, void* reduce_user,
                     sp_challenge_hook observe, void* observe_user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8], size_t run_rounds) {
  const size_t vars = rounds;  // the tables have 2^vars elements
  if (tail_hand_n(false) > 32) sp::WalkPool::get().keep_hot(3000);  // the host's share of the rounds is spread over the polling threads: awake by then
  tr->join();  // (a long absorb may still be hashing on the library's thread)
  if (run_rounds > vars) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_quad: more rounds to run than variables");
  const bool stopped = run_rounds != 0 && run_rounds < vars;
  if (stopped) rounds = run_rounds;
  const uint64_t* claim_ = claim_io;
  if (A->len != B->len || A->len != ((size_t)1 << vars)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_quad: tables must have 2^rounds elements");
  fe_t claim = load_fe(claim_);
  const uint8_t lbl_c[1] = {'c'};
  const size_t chunk = 256 * spk::EVAL_PPT;
  int rc = c->ensure_scratch((A->len / 2 + chunk - 1) / chunk * 2 + (A->len / 4 / 64) * 3 + 64);  // block partials or 72-byte lazy wave partials
  if (rc) return rc;
  bool have_sums = false;  // true when a launch already in flight produces this round's sums
  bool in_tail = false;    // the persistent tail kernel owns the remaining rounds
  unsigned tail_nb0 = 1;  // blocks of the resident launch
  bool host_mode = false;  // ... and has handed the tables over: the remaining rounds run on the host (kernels_poly.hpp tail_hand_over)
  std::vector<fe_t> hA;    // host tables after a hand-over: A, then B
  fe_t* hB = nullptr;
  unsigned hand_seq = 0;
  unsigned last_answered = 0;  // sequence number answered by the most recent challenge
  TailLease lease;
  AheadGuard guard(c);
  *reinterpret_cast<volatile uint32_t*>(c->h_pinned + spk::TAIL_ERR_ELEM) = 0;
  // What follows round `round`'s challenge r: the resident tail takes it from the mailbox, a fused launch binds with it and evaluates round + 1, or
  // (last round, irregular zero structure) a plain bind. `r` == nullptr issues the work AHEAD of the challenge (mailbox in device memory): the
  // launch overhead and the kernel's table loads then overlap the host's transcript step. Returns 1 = issued, 0 = needs r on the host, < 0 error.
  auto issue = [&](size_t round, const fe_t* r, unsigned answers) -> int {
    const bool ahead = r == nullptr;
    const fe_t rv = r ? *r : fe_zero();
    if (in_tail) {  // the resident kernel binds (and evaluates the next round) as soon as it sees the challenge
      sp::after_bind(A);
      sp::after_bind(B);
      have_sums = false;
      if (round + 1 < rounds) {
        next_seq(c);
        have_sums = true;
        c->pending_slots = tail_blocks(A->len) > 1 ? tail_nb0 : 1u;  // every block of the launch publishes while more than one would (the kernel's local regime)
      }
      return 1;
    }
    if (round + 1 >= rounds) {
      if (ahead) return 0;
      sp_table* tabs[2] = {A, B};
      have_sums = false;
      int rc2 = launch_bind(c, tabs, 2, rv);
      return rc2 ? rc2 : 1;
    }
    const bool gated = ahead && !launch_ahead_ok(c, A->len);
    if (gated && !gate_ok(c)) return 0;
    if (!stopped && tail_enabled() && A->len <= TAIL_MAX_LEN && table_dense(A) && table_dense(B) && lease.take(tail_blocks(A->len / 2))) {
      spk::TailArgs ta;
      ta.A = A->d;
      ta.B = B->d;
      ta.C = nullptr;
      ta.len = A->len;
      ta.r0 = rv;
      ta.eq_pl = ta.eq_pr = nullptr;
      ta.ell = ta.first_half = ta.rnd0 = 0;
      ta.mail = c->d_mail;
      ta.mirror = c->d_mail_mirror;
      ta.r0_from_mail = ahead ? 1 : 0;
      ta.mapped = c->d_pinned;
      ta.seq0 = next_seq(c);
      ta.hand_n = tail_hand_n(false);
      tail_nb0 = tail_blocks(A->len / 2);
      hipLaunchKernelGGL((spk::k_sumcheck_tail<false>), dim3(tail_nb0), dim3(spk::TAIL_THREADS), 0, c->stream, ta);
      // the resident kernel stores the final claims into element 0 of its tables AFTER the host has them and has returned (hand-over): whoever rewrites
      // those tables on another stream - the next prove's sp_table_assemble_aside of z - orders itself behind this event
      if (!c->tail_ev) (void)hipEventCreateWithFlags(&c->tail_ev, hipEventDisableTiming);
      if (c->tail_ev && hipEventRecord(c->tail_ev, c->stream) == hipSuccess) c->tail_ev_pending = true;
      in_tail = true;
      have_sums = true;
      sp::after_bind(A);
      sp::after_bind(B);
      c->pending_slots = tail_blocks(A->len);
      return 1;
    }
    const bool dense = table_dense(A) && table_dense(B);
    const bool sparse_stream = !dense && A->len / 4 >= STREAM_MIN_Q && sp::eff_lo(A) == A->len / 2 && sp::eff_lo(B) == B->len / 2 && sp::eff_hi(A) <= A->len / 4 && sp::eff_hi(B) <= B->len / 4;
    if (gated && !dense && !sparse_stream) return 0;  // (the plain bind that follows takes its challenge as an argument)
    const spk::MailRef mref = gated ? gate_launch(c, answers) : mail_ref(c, ahead, answers);
    if (dense) {
      // fused: bind this round, evaluate the next (K1 + K3 in one pass over the tables)
      const size_t q = A->len / 4;
      size_t blocks = (q + chunk - 1) / chunk;
      if (q >= STREAM_MIN_Q) {  // streaming regime: wave-level lazy partials + lazy second stage
        const unsigned seq = next_seq(c);
        const spk::LazyOut lp = lazy_out(c, reinterpret_cast<spk::lazy9_t*>(c->d_scratch), q / 256, 2, nullptr, seq);
        c->timed("bind_stream_quad", 48ull * A->len * 2, [&] {
          hipLaunchKernelGGL(spk::k_bind_eval_quad_stream, dim3((unsigned)(q / 256)), dim3(256), 0, c->stream, A->d, B->d, q, rv, lp, mref);
        });
        sum_lazy_launch(c, lp, q / 256);
      } else {
        c->timed("bind", 48ull * A->len * 2, [&] {
          hipLaunchKernelGGL(spk::k_bind_eval_quad, dim3((unsigned)blocks), dim3(256), 0, c->stream, A->d, B->d, q, rv, c->d_scratch, c->d_pinned, next_seq(c), mref);
        });
        reduce_partials_launch(c, blocks, 2);
      }
      sp::after_bind(A);
      sp::after_bind(B);
      have_sums = true;
      return 1;
    }
    if (sparse_stream) {
      // full low half, (almost) empty high half: bind without reading the zeros, evaluate the next round from registers
      const size_t q = A->len / 4;
      const unsigned seq = next_seq(c);
      const spk::LazyOut lp = lazy_out(c, reinterpret_cast<spk::lazy9_t*>(c->d_scratch), q / 256, 2, nullptr, seq);
      c->timed("bind_stream_quad_sparse", 64ull * (A->len / 2) * 2, [&] {
        hipLaunchKernelGGL(spk::k_bind_eval_quad_stream_sparse, dim3((unsigned)(q / 256)), dim3(256), 0, c->stream, A->d, B->d, q, rv, sp::eff_hi(A), sp::eff_hi(B), lp,
                           mref);
      });
      sum_lazy_launch(c, lp, q / 256);
      sp::after_bind(A);
      sp::after_bind(B);
      have_sums = true;
      return 1;
    }
    if (ahead) return 0;
    sp_table* tabs[2] = {A, B};
    have_sums = false;
    int rc2 = launch_bind(c, tabs, 2, rv);
    return rc2 ? rc2 : 1;
  };
  for (size_t round = 0; round < rounds; ++round) {
    const size_t half = A->len / 2, len_now = A->len;
    const double tr0 = round_trace() ? now_us() : 0;
    fe_t sums[2] = {fe_zero(), fe_zero()};
    bool waiting = have_sums;
    if (!have_sums && !host_mode) {  // compute_eval_points_quad on the current tables (src/sumcheck.rs:128-174)
      size_t len = sp::eff_pairs(A);
      if (sp::eff_pairs(B) < len) len = sp::eff_pairs(B);
      if (half < len) len = half;
      if (len >= STREAM_MIN_Q && len % 1024 == 0) {  // streaming form: lazy sums, lazy second stage
        const unsigned seq = next_seq(c);
        const size_t hi_max = sp::eff_hi(A) > sp::eff_hi(B) ? sp::eff_hi(A) : sp::eff_hi(B);
        const bool lowhi = hi_max <= len / 2;  // short non-zero prefix in the high halves: dot-product form
        const size_t blocks = len / (256 * (size_t)4);
        const spk::LazyOut lp = lazy_out(c, reinterpret_cast<spk::lazy9_t*>(c->d_scratch), blocks, 2, nullptr, seq);
        c->timed("eval_quad", 64ull * len + 32ull * ((sp::eff_hi(A) < len ? sp::eff_hi(A) : len) + (sp::eff_hi(B) < len ? sp::eff_hi(B) : len)), [&] {
          if (lowhi)
            hipLaunchKernelGGL((spk::k_eval_quad_stream_lowhi<4>), dim3((unsigned)blocks), dim3(256), 0, c->stream, A->d, B->d, half, sp::eff_hi(A), sp::eff_hi(B), lp);
          else
            hipLaunchKernelGGL((spk::k_eval_quad_stream<4>), dim3((unsigned)blocks), dim3(256), 0, c->stream, A->d, B->d, half, sp::eff_hi(A), sp::eff_hi(B), lp);
        });
        sum_lazy_launch(c, lp, blocks);
        waiting = true;
      } else if (len > 0) {
        size_t blocks = (len + chunk - 1) / chunk;
        c->timed("eval_quad", 64ull * len + 32ull * ((sp::eff_hi(A) < len ? sp::eff_hi(A) : len) + (sp::eff_hi(B) < len ? sp::eff_hi(B) : len)),
                 [&] { hipLaunchKernelGGL(spk::k_eval_quad, dim3((unsigned)blocks), dim3(256), 0, c->stream, A->d, B->d, half, len, sp::eff_hi(A), sp::eff_hi(B), c->d_scratch, c->d_pinned, next_seq(c)); });
        reduce_partials_launch(c, blocks, 2);
        waiting = true;
      }
    }
    // this round's sums are in flight: remember how to wait for them, then issue the next launch ahead of the challenge where that is possible
    const unsigned wait_seq = c->result_seq, wait_slots = c->pending_slots;
    const bool wait_resident = in_tail;
    if (wait_resident && !host_mode && spk::tail_hand_over(len_now, tail_hand_n(false))) {
      // HAND-OVER (kernels_poly.hpp tail_hand_over): the resident kernel sent the tables themselves; this round and the ones after it run here
      hA.resize(2 * len_now);
      if ((rc = wait_hand_over(c, wait_seq, (int)(2 * len_now), hA.data()))) return rc;
      hB = hA.data() + len_now;
      host_mode = true;
      hand_seq = wait_seq;
    }
    if (host_mode) {  // compute_eval_points_quad (src/sumcheck.rs:128-174) on the host tables; dense by construction (the tail only takes dense tables)
      host_quad_eval(hA.data(), hB, half, sums);
      waiting = false;
    }
    if (!host_mode && wait_resident && spk::tail_double(false, len_now, tail_hand_n(false))) {
      // TWO ROUNDS IN THIS TRIP (kernels_poly.hpp TAIL_WIDE_VALS): the resident kernel sent the sums of this round and the coefficient sums, in this
      // round's challenge, of the next. S0 = sum a0 b0, S1 = a1 b1, S2 = (a2-a0)(b2-b0), S3 = (a3-a1)(b3-b1), S4 = a2 b2, S5 = U V, S6 = dU dV,
      // S7 = (U+dU)(V+dV) over the quarters a0..a3 of the table.
      fe_t S8[8];
      if ((rc = wait_wide(c, wait_seq, spk::TAIL_DOUBLE_SUMS_QUAD, S8))) return rc;
      if (reduce) {
        int hrc = reduce(reduce_user, reinterpret_cast<uint64_t*>(S8), 8);
        if (hrc) return fail(hrc, "prove_quad: the reduce hook failed");
      }
      fe_t rr[2];
      for (int half2 = 0; half2 < 2; ++half2) {
        fe_t e0, tinf;
        if (half2 == 0) {
          e0 = fe_add<S>(S8[0], S8[1]);
          tinf = fe_add<S>(S8[2], S8[3]);
        } else {  // eval_0(r) = S0 + r (S4 - S0 - S2) + r^2 S2, t_inf(r) = S5 + r (S7 - S5 - S6) + r^2 S6
          const fe_t r = rr[0];
          e0 = fe_add<S>(S8[0], fe_mul<S>(r, fe_add<S>(fe_sub<S>(fe_sub<S>(S8[4], S8[0]), S8[2]), fe_mul<S>(r, S8[2]))));
          tinf = fe_add<S>(S8[5], fe_mul<S>(r, fe_add<S>(fe_sub<S>(fe_sub<S>(S8[7], S8[5]), S8[6]), fe_mul<S>(r, S8[6]))));
        }
        // eval_2 = 2 claim - 3 eval_0 + 2 t_inf and the interpolation of (eval_0, claim - eval_0, eval_2) (src/sumcheck.rs:211-215, univariate.rs:84-93)
        // give c0 = eval_0, c2 = t_inf, c1 = claim - 2 eval_0 - t_inf: written down directly
        UniPoly poly;
        poly.n = 3;
        poly.c[0] = e0;
        poly.c[1] = fe_sub<S>(fe_sub<S>(claim, fe_dbl<S>(e0)), tinf);
        poly.c[2] = tinf;
        absorb_poly(tr->t, poly);
        if (!tr->t.squeeze<S>(lbl_c, 1, &rr[half2])) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
        tail_post_challenge(c, rr[half2], wait_seq + half2);  // (the first at once: the kernel's first bind runs under the second round's host step)
        store_fe(out_r + 4 * (round + half2), rr[half2]);
        store_fe(out_cpolys + 8 * (round + half2), poly.c[0]);
        store_fe(out_cpolys + 8 * (round + half2) + 4, poly.c[2]);
        claim = poly_eval(poly, rr[half2]);
      }
      // the kernel binds twice and (if rounds are left) evaluates again: its next result is tagged wait_seq + 2
      for (int k = 0; k < 2; ++k) {
        sp::after_bind(A);
        sp::after_bind(B);
        next_seq(c);
      }
      have_sums = round + 2 < rounds;
      c->pending_slots = have_sums ? 1u : 0u;
      last_answered = wait_seq + 1;
      guard.armed = round + 2 < rounds;
      if (observe)
        for (int k = 0; k < 2; ++k) {
          uint64_t rw[4];
          store_fe(rw, rr[k]);
          observe(observe_user, round + k, rw);
        }
      if (round_trace()) fprintf(stderr, "quad rounds %2zu+%2zu len %8zu tail 2 wait+host %7.1f us\n", round, round + 1, len_now, now_us() - tr0);
      ++round;
      continue;
    }
    int issued = 0;
    if (!host_mode && waiting && (in_tail || (c->mail_dev && !reduce))) {  // not with a reduce hook: it may run device work (a collective) beside the waiting kernel
      issued = issue(round, nullptr, wait_seq);
      if (issued < 0) return issued;
      guard.armed = issued != 0;
    }
    if (waiting) {
      const unsigned cur_seq = c->result_seq, cur_slots = c->pending_slots;
      c->result_seq = wait_seq;
      c->pending_slots = wait_slots;
      // (a launch issued ahead of its challenge sits on the stream behind these sums: like the resident tail it waits for the host, so the wait must
      // never fall back to a stream synchronise — that was the rare 8 s stall of round 2: a result a few ms late, the fallback, and host and kernel
      // waiting for each other until the kernel's watchdog)
      rc = reduce_partials_wait(c, 2, sums, wait_resident || issued != 0);
      if (issued) {  // back to the state of the launch issued ahead
        c->result_seq = cur_seq;
        c->pending_slots = cur_slots;
      }
      if (rc) return rc;
    }
    if (reduce) {
      int hrc = reduce(reduce_user, reinterpret_cast<uint64_t*>(sums), 2);
      if (hrc) return fail(hrc, "prove_quad: the reduce hook failed");
    }
    const double tr1 = round_trace() ? now_us() : 0;
    // BDDT: eval_2 = 2 claim - 3 eval_0 + 2 t_inf (src/sumcheck.rs:211-215)
    fe_t e0 = sums[0], tinf = sums[1];
    // (eval_0, claim - eval_0, eval_2 = 2 claim - 3 eval_0 + 2 t_inf) interpolated (univariate.rs:84-93) = c0 = eval_0, c2 = t_inf, c1 = claim - 2 eval_0 - t_inf
    UniPoly poly;
    poly.n = 3;
    poly.c[0] = e0;
    poly.c[1] = fe_sub<S>(fe_sub<S>(claim, fe_dbl<S>(e0)), tinf);
    poly.c[2] = tinf;
    absorb_poly(tr->t, poly);
    fe_t r_i;
    if (!tr->t.squeeze<S>(lbl_c, 1, &r_i)) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
    last_answered = wait_seq;
    if (host_mode) {
      host_bind_tables(hA.data(), (size_t)(hB - hA.data()), 2, len_now, r_i);
      sp::after_bind(A);
      sp::after_bind(B);
      have_sums = false;
    } else if (issued) {
      tail_post_challenge(c, r_i, wait_seq);  // whatever was issued ahead is waiting for exactly this: first, the book-keeping below runs under it
    } else {
      issued = issue(round, &r_i, wait_seq);
      if (issued < 0) return issued;
      if (wait_resident) tail_post_challenge(c, r_i, wait_seq);
    }
    store_fe(out_r + 4 * round, r_i);
    store_fe(out_cpolys + 8 * round, poly.c[0]);
    store_fe(out_cpolys + 8 * round + 4, poly.c[2]);
    claim = poly_eval(poly, r_i);
    guard.armed = in_tail && (host_mode || round + 1 < rounds);  // the resident kernel now waits for the next challenge (after a hand-over: for the final claims)
    if (observe) {  // after the device has been given this round's challenge: the observer's work runs under the next round
      uint64_t rw[4];
      store_fe(rw, r_i);
      observe(observe_user, round, rw);
    }
    if (round_trace()) fprintf(stderr, "quad round %2zu len %8zu tail %d wait %7.1f us host %6.1f us\n", round, len_now, (int)wait_resident, tr1 - tr0, now_us() - tr1);
  }
  if (stopped) {
    store_fe(claim_io, claim);
    return tail_check(c);
  }
  if (host_mode) {  // the final claims are here; the kernel, waiting at the mailbox, stores them into element 0 of its tables and leaves
    store_fe(out_final, hA[0]);
    store_fe(out_final + 4, hB[0]);
    tail_post_challenge(c, hA[0], hand_seq);
    tail_post_challenge(c, hB[0], hand_seq + 1);
    c->result_seq = hand_seq + 1;  // the two lines took sequence numbers of their own
    guard.armed = false;
  } else if (in_tail) {  // the resident kernel hands the final claims over itself
    fe_t fin[3];
    long spins = 0;
    if ((rc = wait_slot(c, c->h_pinned + spk::TAIL_FINAL_ELEM, last_answered, 2, fin, true, &spins))) return rc;
    store_fe(out_final, fin[0]);
    store_fe(out_final + 4, fin[1]);
  } else {
    rc = sp_table_read(c, A, 0, 1, out_final);
    if (rc) return rc;
    rc = sp_table_read(c, B, 0, 1, out_final + 4);
    if (rc) return rc;
  }
  store_fe(claim_io, claim);
  return tail_check(c);  // also set by a kernel launched ahead that gave up waiting for its challenge
}

// compute_eval_points_quad (src/sumcheck.rs:128-174) of the current tables -> (eval0, t_inf)
static int eval_quad_sums(sp_ctx* c, const sp_table* A, const sp_table* B, fe_t sums[2]) {
  const size_t chunk = 256 * spk::EVAL_PPT, half = A->len / 2;
  sums[0] = sums[1] = fe_zero();
  size_t len = sp::eff_pairs(A);
  if (sp::eff_pairs(B) < len) len = sp::eff_pairs(B);
  if (half < len) len = half;
  if (len == 0) return SP_OK;
  size_t blocks = (len + chunk - 1) / chunk;
  int rc = c->ensure_scratch(blocks * 2 + 64);
  if (rc) return rc;
  c->timed("eval_quad", 64ull * len + 32ull * ((sp::eff_hi(A) < len ? sp::eff_hi(A) : len) + (sp::eff_hi(B) < len ? sp::eff_hi(B) : len)),
           [&] { hipLaunchKernelGGL(spk::k_eval_quad, dim3((unsigned)blocks), dim3(256), 0, c->stream, A->d, B->d, half, len, sp::eff_hi(A), sp::eff_hi(B), c->d_scratch, c->d_pinned, next_seq(c)); });
  return reduce_partials(c, blocks, 2, sums);
}

// Both instances of a batched round in one launch and one wait (k_eval_*_pair). Returns 1 when done, 0 when the tables are too large for the slot
// path (the caller then evaluates the instances one after the other), < 0 on error.
static unsigned pair_pp(size_t len, unsigned* nb_out) {
  for (unsigned pp = 1; pp <= 8; pp <<= 1) {
    const size_t nb = (len + 256 * (size_t)pp - 1) / (256 * (size_t)pp);
    if (2 * nb <= (size_t)spk::HOST_SUM_MAX_BLOCKS) {
      *nb_out = (unsigned)(nb ? nb : 1);
      return pp;
    }
  }
  return 0;
}
static int eval_quad_sums_pair(sp_ctx* c, sp_table* const A[2], sp_table* const B[2], fe_t sums[2][2]) {
  const size_t half = A[0]->len / 2;
  if (A[1]->len != A[0]->len) return 0;
  spk::QuadPairArgs t;
  size_t maxlen = 0;
  for (int b = 0; b < 2; ++b) {
    size_t len = sp::eff_pairs(A[b]);
    if (sp::eff_pairs(B[b]) < len) len = sp::eff_pairs(B[b]);
    if (half < len) len = half;
    t.A[b] = A[b]->d;
    t.B[b] = B[b]->d;
    t.len[b] = len;
    t.hiA[b] = sp::eff_hi(A[b]);
    t.hiB[b] = sp::eff_hi(B[b]);
    if (len > maxlen) maxlen = len;
  }
  unsigned nb = 0;
  const unsigned pp = maxlen ? pair_pp(maxlen, &nb) : 0;
  if (!pp) return 0;
  const unsigned seq = next_seq(c);
  c->timed("eval_quad", 64ull * (t.len[0] + t.len[1]), [&] { hipLaunchKernelGGL(spk::k_eval_quad_pair, dim3(2 * nb), dim3(256), 0, c->stream, t, half, nb, pp, c->d_pinned, seq); });
  c->pending_slots = 2 * nb;
  fe_t out[4];
  int rc = reduce_partials_wait(c, 2, out, false, 2);
  if (rc) return rc;
  sums[0][0] = out[0];
  sums[0][1] = out[1];
  sums[1][0] = out[2];
  sums[1][1] = out[3];
  return 1;
}
static int eval_cubic_outer_pow_pair(sp_ctx* c, const sp_table* pl, const sp_table* pr, sp_table* const step[3], sp_table* const core[3], fe_t sums[2][3]) {
  const size_t len = step[0]->len / 2, left = pl->len;
  if (core[0]->len != step[0]->len || len == 0) return 0;
  const bool fallback = len < left;
  size_t right = 0;
  if (fallback) {
    if (left < 2 * len) return 0;
  } else {
    if (left == 0 || len % left) return 0;
    right = len / left;
    if (pr->len < 2 * right) return 0;
  }
  unsigned nb = 0;
  const unsigned pp = pair_pp(len, &nb);
  if (!pp) return 0;
  spk::CubicPairArgs t;
  for (int b = 0; b < 2; ++b) {
    sp_table* const* q = b == 0 ? step : core;
    t.A[b] = q[0]->d;
    t.B[b] = q[1]->d;
    t.C[b] = q[2]->d;
  }
  const unsigned seq = next_seq(c);
  c->timed("eval_cubic_pow", 2 * 224ull * len, [&] {
    if (fallback)
      hipLaunchKernelGGL((spk::k_eval_cubic_outer_pow_pair<true>), dim3(2 * nb), dim3(256), 0, c->stream, pl->d, left, pr ? pr->d : nullptr, right, t, len, nb, pp, c->d_pinned,
                         seq);
    else
      hipLaunchKernelGGL((spk::k_eval_cubic_outer_pow_pair<false>), dim3(2 * nb), dim3(256), 0, c->stream, pl->d, left, pr->d, right, t, len, nb, pp, c->d_pinned, seq);
  });
  c->pending_slots = 2 * nb;
  fe_t out[6];
  int rc = reduce_partials_wait(c, 3, out, false, 2);
  if (rc) return rc;
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 3; ++k) sums[b][k] = out[3 * b + k];
  return 1;
}

// Small tables (<= 2^13 elements): the bind of round i and the evaluation of round i + 1 of both instances in one launch with one product per lane
// (k_bind_eval_*_pair_small); dense tables of equal length only. Returns 1 when done (the tables are bound and `sums` holds the next round's), 0 when
// this form does not apply (the caller binds and evaluates separately), < 0 on error.
// groups of 64 pairs a block of the fused small-table launches takes one after the other (kernels_poly.hpp `chunks`; measured: a group is 8-10 us of
// latency, so more than one only loses - 1 unless SPARTAN_SMALL_PAIR_CHUNKS says otherwise), and whether the launch fits the result slots: the 64 ordinary
// ones up to q = 2048 pairs, the wide area (WIDE_SLOTS, added by the host) up to q = 8192. 0 = too large for the form.
static size_t small_pair_chunks(size_t q) {
  static const size_t cap = [] {
    const char* e = getenv("SPARTAN_SMALL_PAIR_CHUNKS");
    const int v = e ? atoi(e) : 1;
    return (size_t)(v < 1 ? 1 : (v > 16 ? 16 : v));
  }();
  static const bool wide_ok = [] {
    const char* e = getenv("SPARTAN_SMALL_PAIR_WIDE");  // "0": the round-5 reach (q <= 2048)
    return !(e && e[0] == '0');
  }();
  const size_t max_slots = wide_ok ? (size_t)spk::WIDE_SLOTS : (size_t)spk::HOST_SUM_MAX_BLOCKS;
  for (size_t ch = 1; ch <= cap; ch *= 2)
    if (2 * ((q + spk::SMALL_PAIR_PPB * ch - 1) / (spk::SMALL_PAIR_PPB * ch)) <= max_slots) return ch;
  return 0;
}
// The fused bind + evaluate of a batched round in two halves. _issue launches (returns 0 when the form does not apply: nothing was launched); with
// `r` == nullptr the launch is queued AHEAD of its challenge - the kernel waits at the mailbox for the challenge that answers the result in hand
// (`*answers` = its sequence number; the caller posts it with tail_post_challenge once its hook has drawn it). _collect waits for the sums.
struct SmallPairLaunch {
  size_t nb = 0;
  unsigned answers = 0, seq = 0;  // the result its challenge answers; the result it publishes
  bool ahead = false, valid = false;
};
// waits for the slots of launch L although a later launch may have been issued since (its sequence number and slot count are put back afterwards)
static int small_pair_wait(sp_ctx* c, const SmallPairLaunch& L, int nacc, fe_t* out) {
  const unsigned cur_seq = c->result_seq, cur_slots = c->pending_slots;
  c->result_seq = L.seq;
  c->pending_slots = (unsigned)(2 * L.nb);
  const int rc = reduce_partials_wait(c, nacc, out, true, 2, 2 * L.nb > (size_t)spk::HOST_SUM_MAX_BLOCKS);  // (resident: a later launch may be waiting at the mailbox)
  c->result_seq = cur_seq;
  c->pending_slots = cur_seq == L.seq ? 0u : cur_slots;
  return rc;
}
static int bind_eval_quad_pair_small_issue(sp_ctx* c, sp_table* const A[2], sp_table* const B[2], const fe_t* r, SmallPairLaunch* L) {
  const size_t len = A[0]->len;
  if (len < 4) return 0;
  for (int b = 0; b < 2; ++b)
    if (A[b]->len != len || B[b]->len != len || !table_dense(A[b]) || !table_dense(B[b])) return 0;
  const size_t q = len / 4, chunks = small_pair_chunks(q);
  if (!chunks) return 0;
  const size_t nb = (q + spk::SMALL_PAIR_PPB * chunks - 1) / (spk::SMALL_PAIR_PPB * chunks);
  spk::QuadPairBindArgs t;
  for (int b = 0; b < 2; ++b) {
    t.A[b] = A[b]->d;
    t.B[b] = B[b]->d;
  }
  L->nb = nb;
  L->ahead = r == nullptr;
  L->answers = c->result_seq;
  const spk::MailRef m = mail_ref(c, L->ahead, L->answers);
  const fe_t rv = r ? *r : fe_zero();
  const unsigned seq = next_seq(c);
  L->seq = seq;
  L->valid = true;
  c->timed("bind_eval_quad_pair_small", 2 * 192ull * q, [&] {
    hipLaunchKernelGGL(spk::k_bind_eval_quad_pair_small, dim3((unsigned)(2 * nb)), dim3(256), 0, c->stream, t, (unsigned)q, (unsigned)nb, (unsigned)chunks, rv, m, c->d_pinned, seq);
  });
  for (int b = 0; b < 2; ++b) {
    sp::after_bind(A[b]);
    sp::after_bind(B[b]);
  }
  c->pending_slots = (unsigned)(2 * nb);
  return 1;
}
static int bind_eval_quad_pair_small_collect(sp_ctx* c, const SmallPairLaunch& L, fe_t sums[2][2]) {
  fe_t out[4];
  int rc = small_pair_wait(c, L, 2, out);
  if (rc) return rc;
  sums[0][0] = out[0];
  sums[0][1] = out[1];
  sums[1][0] = out[2];
  sums[1][1] = out[3];
  return 1;
}
static int bind_eval_cubic_pow_pair_small_issue(sp_ctx* c, const sp_table* pl, const sp_table* pr, sp_table* const step[3], sp_table* const core[3], const fe_t* r,
                                                SmallPairLaunch* L) {
  const size_t len = step[0]->len, left = pl->len;
  if (len < 4) return 0;
  for (int k = 0; k < 3; ++k)
    if (step[k]->len != len || core[k]->len != len || !table_dense(step[k]) || !table_dense(core[k])) return 0;
  const size_t q = len / 4, chunks = small_pair_chunks(q);
  if (!chunks) return 0;
  const size_t nb = (q + spk::SMALL_PAIR_PPB * chunks - 1) / (spk::SMALL_PAIR_PPB * chunks);
  const bool fallback = q < left;  // as eval_cubic_outer_pow_pair, for q pairs
  size_t right = 0;
  if (fallback) {
    if (left < 2 * q) return 0;
  } else {
    if (left == 0 || q % left) return 0;
    right = q / left;
    if (pr->len < 2 * right) return 0;
  }
  spk::CubicPairBindArgs t;
  for (int b = 0; b < 2; ++b) {
    sp_table* const* tb = b == 0 ? step : core;
    t.A[b] = tb[0]->d;
    t.B[b] = tb[1]->d;
    t.C[b] = tb[2]->d;
  }
  L->nb = nb;
  L->ahead = r == nullptr;
  L->answers = c->result_seq;
  const spk::MailRef m = mail_ref(c, L->ahead, L->answers);
  const fe_t rv = r ? *r : fe_zero();
  const unsigned seq = next_seq(c);
  L->seq = seq;
  L->valid = true;
  c->timed("bind_eval_cubic_pow_pair_small", 2 * 288ull * q, [&] {
    if (fallback)
      hipLaunchKernelGGL((spk::k_bind_eval_cubic_pow_pair_small<true>), dim3((unsigned)(2 * nb)), dim3(256), 0, c->stream, pl->d, left, pr ? pr->d : nullptr, right, t, (unsigned)q,
                         (unsigned)nb, (unsigned)chunks, rv, m, c->d_pinned, seq);
    else
      hipLaunchKernelGGL((spk::k_bind_eval_cubic_pow_pair_small<false>), dim3((unsigned)(2 * nb)), dim3(256), 0, c->stream, pl->d, left, pr->d, right, t, (unsigned)q, (unsigned)nb,
                         (unsigned)chunks, rv, m, c->d_pinned, seq);
  });
  for (int k = 0; k < 3; ++k) {
    sp::after_bind(step[k]);
    sp::after_bind(core[k]);
  }
  c->pending_slots = (unsigned)(2 * nb);
  return 1;
}
static int bind_eval_cubic_pow_pair_small_collect(sp_ctx* c, const SmallPairLaunch& L, fe_t sums[2][3]) {
  fe_t out[6];
  int rc = small_pair_wait(c, L, 3, out);
  if (rc) return rc;
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 3; ++k) sums[b][k] = out[3 * b + k];
  return 1;
}
// may the next round's launch be queued ahead of the caller's round hook? (sp_ctx_round_hooks_host_only; the mailbox must be in device memory: a grid
// of up to 256 blocks polls it. SPARTAN_BATCHED_AHEAD=0: never, for A/B runs)
static bool batched_ahead_ok(const sp_ctx* c) {
  static const bool off = [] {
    const char* e = getenv("SPARTAN_BATCHED_AHEAD");
    return e && e[0] == '0';
  }();
  return c->hooks_host_only && c->mail_dev && !off;
}

// prove_quad_batched_zk (src/sumcheck.rs:702-782): two quadratic sum-checks (step, core) driven by one challenge per round; the challenge
// comes from the caller's `process_round` hook (:747-755).
int sp_sumcheck_quad_batched(sp_ctx* c, const uint64_t claims_[8], size_t num_rounds, sp_table* A0, sp_table* A1, sp_table* B0, sp_table* B1, size_t start_round,
                             sp_round_hook hook, void* user, uint64_t* out_r, uint64_t out_final[16]) {
  const size_t n = (size_t)1 << num_rounds;
  if (A0->len != n || A1->len != n || B0->len != n || B1->len != n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_quad_batched: tables must have 2^num_rounds elements");
  fe_t claim[2] = {load_fe(claims_), load_fe(claims_ + 4)};
  sp_table* br[2][2] = {{A0, B0}, {A1, B1}};
  bool have_next = false;  // `both` already holds this round's sums (the previous round's fused bind + evaluate)
  fe_t both[2][2];
  AheadGuard guard(c);  // a launch queued ahead of the hook is released on every early return
  SmallPairLaunch pre;  // the next round's fused launch when it was queued while the current one ran
  for (size_t j = 0; j < num_rounds; ++j) {
    UniPoly poly[2];
    uint64_t co[2][12];
    sp_table* const pa[2] = {A0, A1};
    sp_table* const pb[2] = {B0, B1};
    const int paired = have_next ? 1 : eval_quad_sums_pair(c, pa, pb, both);
    if (paired < 0) return paired;
    for (int b = 0; b < 2; ++b) {
      fe_t sums[2];
      if (paired) {
        sums[0] = both[b][0];
        sums[1] = both[b][1];
      } else {
        int rc = eval_quad_sums(c, br[b][0], br[b][1], sums);
        if (rc) return rc;
      }
      const fe_t e0 = sums[0], tinf = sums[1];
      const fe_t three_e0 = fe_add<S>(fe_add<S>(e0, e0), e0);
      fe_t ev[3] = {e0, fe_sub<S>(claim[b], e0), fe_add<S>(fe_add<S>(fe_sub<S>(fe_add<S>(claim[b], claim[b]), three_e0), tinf), tinf)};
      poly[b] = from_evals_deg2(ev);
      for (int q = 0; q < 3; ++q) store_fe(co[b] + 4 * q, poly[b].c[q]);
    }
    uint64_t r_raw[4];
    static const bool trace = [] {
      const char* e = getenv("SPARTAN_HOST_LAPS");
      return e && e[0] == '2';
    }();
    const double tq0 = trace ? now_us() : 0;
    // the next round's fused launch, queued ahead of the hook when the caller has promised a host-only hook: it waits at the mailbox for r_j
    SmallPairLaunch L;
    int issued = 0;
    if (pre.valid) {  // queued during the previous round, while that round's launch ran
      L = pre;
      pre.valid = false;
      issued = 1;
    } else if (j + 1 < num_rounds && batched_ahead_ok(c)) {
      issued = bind_eval_quad_pair_small_issue(c, pa, pb, nullptr, &L);
      if (issued < 0) return issued;
    }
    guard.armed = issued == 1;
    int hrc = hook(user, start_round + j, co[0], co[1], 3, r_raw);
    if (trace) fprintf(stderr, "inner batched round %zu (len %zu, %s%s): hook %.1f us\n", j, A0->len, have_next ? "fused" : "separate launches", issued ? ", next queued ahead" : "", now_us() - tq0);
    if (hrc) return fail(hrc, "prove_quad_batched: the round hook failed");
    const fe_t r_j = load_fe(r_raw);
    if (issued) {
      tail_post_challenge(c, r_j, L.answers);
      guard.armed = false;
    }
    store_fe(out_r + 4 * j, r_j);
    claim[0] = poly_eval(poly[0], r_j);
    claim[1] = poly_eval(poly[1], r_j);
    have_next = false;
    if (j + 1 < num_rounds) {
      if (!issued) {
        issued = bind_eval_quad_pair_small_issue(c, pa, pb, &r_j, &L);
        if (issued < 0) return issued;
      }
      if (issued) {
        if (j + 2 < num_rounds && batched_ahead_ok(c)) {  // the launch AFTER this one goes into the queue while this one runs: it waits for round j + 1's challenge
          const int more = bind_eval_quad_pair_small_issue(c, pa, pb, nullptr, &pre);
          if (more < 0) return more;
          if (more == 0) pre.valid = false;
          guard.armed = pre.valid;
        }
        const int fused = bind_eval_quad_pair_small_collect(c, L, both);
        if (fused < 0) return fused;
        have_next = true;
      }
    }
    if (!have_next) {
      sp_table* tabs[4] = {A0, B0, A1, B1};
      int rc = launch_bind(c, tabs, 4, r_j);
      if (rc) return rc;
    }
  }
  sp_table* fin[4] = {A0, A1, B0, B1};
  for (int q = 0; q < 4; ++q) {
    int rc = sp_table_read(c, fin[q], 0, 1, out_final + 4 * q);
    if (rc) return rc;
  }
  return SP_OK;
}

// prove_cubic_with_additive_term_batched_zk (src/sumcheck.rs:786-917): the batched NeutronNova outer sum-check over the folded step layers and the
// core layers with the split power-of-tau table; element 0 of pow_left receives base_tau at the end (:913).
int sp_sumcheck_cubic_outer_pow_batched(sp_ctx* c, size_t num_rounds, sp_table* pow_left, const sp_table* pow_right, sp_table* A_step, sp_table* B_step,
                                        sp_table* C_step, sp_table* A_core, sp_table* B_core, sp_table* C_core, const uint64_t t_out_step[4], size_t start_round,
                                        sp_round_hook hook, void* user, uint64_t* out_r) {
  const size_t n = (size_t)1 << num_rounds;
  sp_table* step[3] = {A_step, B_step, C_step};
  sp_table* core[3] = {A_core, B_core, C_core};
  for (int q = 0; q < 3; ++q)
    if (step[q]->len != n || core[q]->len != n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_batched: tables must have 2^num_rounds elements");
  const size_t left = pow_left->len, right = pow_right->len;
  if (left * right != n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_batched: pow tables must factor 2^num_rounds");
  std::vector<fe_t> pl(left), pr(right);
  int rc = sp_table_read(c, pow_left, 0, left, reinterpret_cast<uint64_t*>(pl.data()));
  if (rc) return rc;
  if ((rc = sp_table_read(c, pow_right, 0, right, reinterpret_cast<uint64_t*>(pr.data())))) return rc;
  const fe_t one = fe_one<S>();
  fe_t base_tau = one, claim[2] = {load_fe(t_out_step), fe_zero()};
  size_t len_pow_tau = n;
  static const bool trace = [] {
    const char* e = getenv("SPARTAN_HOST_LAPS");
    return e && e[0] == '2';
  }();
  auto nowus = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  bool have_next = false;  // `both` already holds this round's sums (the previous round's fused bind + evaluate)
  fe_t both[2][3];
  AheadGuard guard(c);
  SmallPairLaunch pre;
  for (size_t i = 0; i < num_rounds; ++i) {
    UniPoly poly[2];
    uint64_t co[2][16];
    const double tr0 = trace ? nowus() : 0;
    const int paired = have_next ? 1 : eval_cubic_outer_pow_pair(c, pow_left, pow_right, step, core, both);
    const double tr1 = trace ? nowus() : 0;
    if (paired < 0) return paired;
    for (int b = 0; b < 2; ++b) {
      sp_table** t = b == 0 ? step : core;
      uint64_t raw[12];
      if (paired) memcpy(raw, both[b], 96);
      else if ((rc = sp_eval_cubic_outer_pow(c, pow_left, pow_right, t[0], t[1], t[2], raw))) return rc;
      const fe_t e0 = fe_mul<S>(load_fe(raw), base_tau), e2 = fe_mul<S>(load_fe(raw + 4), base_tau), e3 = fe_mul<S>(load_fe(raw + 8), base_tau);
      fe_t ev[4] = {e0, fe_sub<S>(claim[b], e0), e2, e3};
      poly[b] = from_evals_deg3(ev);
      for (int q = 0; q < 4; ++q) store_fe(co[b] + 4 * q, poly[b].c[q]);
    }
    uint64_t r_raw[4];
    const double tr2 = trace ? nowus() : 0;
    SmallPairLaunch L;
    int issued = 0;
    if (pre.valid) {  // queued during the previous round (see sp_sumcheck_quad_batched)
      L = pre;
      pre.valid = false;
      issued = 1;
    } else if (i + 1 < num_rounds && batched_ahead_ok(c)) {  // queued ahead of the hook
      issued = bind_eval_cubic_pow_pair_small_issue(c, pow_left, pow_right, step, core, nullptr, &L);
      if (issued < 0) return issued;
    }
    guard.armed = issued == 1;
    int hrc = hook(user, start_round + i, co[0], co[1], 4, r_raw);
    if (trace) fprintf(stderr, "outer batched round %zu (len %zu): eval %.1f us, algebra %.1f us, hook %.1f us%s\n", i, step[0]->len, tr1 - tr0, tr2 - tr1, nowus() - tr2, issued ? " (next round queued ahead)" : "");
    if (hrc) return fail(hrc, "prove_cubic_batched: the round hook failed");
    const fe_t r_i = load_fe(r_raw);
    if (issued) {
      tail_post_challenge(c, r_i, L.answers);
      guard.armed = false;
    }
    store_fe(out_r + 4 * i, r_i);
    claim[0] = poly_eval(poly[0], r_i);
    claim[1] = poly_eval(poly[1], r_i);
    have_next = false;
    if (i + 1 < num_rounds) {
      const double tf0 = trace ? nowus() : 0;
      if (!issued) {
        issued = bind_eval_cubic_pow_pair_small_issue(c, pow_left, pow_right, step, core, &r_i, &L);
        if (issued < 0) return issued;
      }
      if (issued) {
        if (i + 2 < num_rounds && batched_ahead_ok(c)) {
          const int more = bind_eval_cubic_pow_pair_small_issue(c, pow_left, pow_right, step, core, nullptr, &pre);
          if (more < 0) return more;
          if (more == 0) pre.valid = false;
          guard.armed = pre.valid;
        }
        const int fused = bind_eval_cubic_pow_pair_small_collect(c, L, both);
        if (fused < 0) return fused;
        have_next = true;
      }
      if (trace) fprintf(stderr, "outer batched round %zu: bind + next evaluation in one launch %s, %.1f us\n", i, have_next ? "taken" : "not applicable", nowus() - tf0);
    }
    if (!have_next) {
      sp_table* t6[6] = {A_step, A_core, B_step, B_core, C_step, C_core};
      if ((rc = launch_bind(c, t6, 6, r_i))) return rc;
    }
    len_pow_tau >>= 1;
    const fe_t pw = fe_mul<S>(pl[len_pow_tau % left], pr[len_pow_tau / left]);
    base_tau = fe_mul<S>(base_tau, fe_add<S>(fe_mul<S>(fe_sub<S>(pw, one), r_i), one));
  }
  return sp_table_write(c, pow_left, 0, reinterpret_cast<const uint64_t*>(&base_tau), 1);
}

static int cubic_impl(sp_ctx* c, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus_, size_t ell, sp_table* A, sp_table* B, sp_table* C, sp_transcript* tr,
                      const uint64_t* scale_, sp_reduce_hook reduce, void* reduce_user, const sp_table* prod0, const sp_table* prod1, uint64_t* out_cpolys,
                      uint64_t* out_r, uint64_t out_final[12], size_t run_rounds = 0, sp_challenge_hook observe = nullptr, void* observe_user = nullptr);
struct EqAheadObserver {
  sp_ctx* c;
  size_t ell;
  uint64_t r[4 * 20];
};
static void eq_ahead_observe(void* user, size_t round, const uint64_t r[4]) {
  EqAheadObserver* o = (EqAheadObserver*)user;
  if (round + 4 >= o->ell) return;
  memcpy(o->r + 4 * round, r, 32);
  if (round + 5 == o->ell) sp_eq_table_begin(o->c, o->r, o->ell - 4, o->ell);  // a failure leaves no announcement: sp_eq_table_into builds it all
}
int sp_sumcheck_cubic3(sp_ctx* c, const uint64_t claim_[4], const uint64_t* taus_, size_t ell, sp_table* A, sp_table* B, sp_table* C, sp_transcript* tr,
                       uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]) {
  uint64_t claim_io[4], p_io[4];
  memcpy(claim_io, claim_, 32);
  const fe_t one = fe_one<S>();
  store_fe(p_io, one);
  // evals_rx (src/spartan.rs:316) follows this sum-check in the reference's order of calls: its two half pyramids are started here, under the last
  // four rounds (the ones the host runs itself after the resident kernel's hand-over), so that the caller's sp_eq_table_into(r_x) is one launch behind the last challenge (as sp_eq_table_begin / _finish for a caller
  // that announces it itself)
  EqAheadObserver ea{c, ell, {}};
  const bool ahead = ell >= 12 && ell <= 20;
  return cubic_impl(c, claim_io, p_io, taus_, ell, A, B, C, tr, nullptr, nullptr, nullptr, nullptr, nullptr, out_cpolys, out_r, out_final, 0,
                    ahead ? &eq_ahead_observe : nullptr, &ea);
}
// ---- the same provers on HOST tables (round 6) ------------------------------------------------------------------------------------------------------
// The relaxed-Spartan sum-checks over the ZK verifier circuit's instance (src/spartan_relaxed.rs:98-213 inside NeutronNovaZkSNARK::prove) run on tables of
// 2^9 (outer) and 2^12 (inner) elements that the driver holds on the HOST: staged to the device they cost three uploads, a resident kernel and a trip over
// the bus per round (~10 us each, whatever the size) - 0.16 + 0.12 ms at config 3 - for ~10 n products a round, which the process's polling host threads
// do in 1-3 us (walk_pool.hpp). These entry points run every round here: the same polynomials, the same transcript, the same final claims as
// sp_sumcheck_cubic3 / sp_sumcheck_quad on device copies of the tables (tests: against the oracle on the CPU, against the device provers on the GPU).
// The tables are bound in place (element 0 of each = the final claim).
int sp_sumcheck_cubic3_host(sp_ctx* c, const uint64_t claim_[4], const uint64_t* taus_, size_t ell, uint64_t* A, uint64_t* B, uint64_t* C, sp_transcript* tr,
                            uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]) {
  (void)c;
  if (!claim_ || !taus_ || !A || !B || !C || !tr || !out_cpolys || !out_r || !out_final || ell == 0 || ell > 24)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (host tables): null argument or more than 2^24 elements");
  tr->join();
  sp::WalkPool::get().keep_hot(2000);
  const size_t N = (size_t)1 << ell;
  fe_t *ha = reinterpret_cast<fe_t*>(A), *hb = reinterpret_cast<fe_t*>(B), *hc = reinterpret_cast<fe_t*>(C);
  std::vector<fe_t> taus(ell);
  for (size_t i = 0; i < ell; ++i) taus[i] = load_fe(taus_ + 4 * i);
  const fe_t one = fe_one<S>();
  const uint8_t lbl_c[1] = {'c'};
  // 1 / tau_k by one inversion; zeros stay zero (those rounds take the three-sum form), as in cubic_impl
  std::vector<fe_t> inv_tau(ell, fe_zero());
  {
    std::vector<fe_t> pref(ell);
    fe_t run = one;
    for (size_t i = 0; i < ell; ++i) {
      pref[i] = run;
      if (!fe_is_zero(taus[i])) run = fe_mul<S>(run, taus[i]);
    }
    fe_t inv = fe_inv_vartime<S>(run);
    for (size_t i = ell; i-- > 0;) {
      if (fe_is_zero(taus[i])) continue;
      inv_tau[i] = fe_mul<S>(inv, pref[i]);
      inv = fe_mul<S>(inv, taus[i]);
    }
  }
  HostEqLevels heq;
  heq.build(taus.data(), ell, N / 2 ? N / 2 : 1);
  fe_t claim = load_fe(claim_), p = one;
  for (size_t rnd = 1; rnd <= ell; ++rnd) {
    const size_t n = N >> (rnd - 1), hn = n / 2;
    const fe_t tau = taus[rnd - 1];
    const fe_t eq0 = fe_sub<S>(one, tau), slope = fe_sub<S>(tau, eq0), eqm1 = fe_sub<S>(eq0, slope);
    const fe_t* E = heq.level(rnd);
    if (!E) return fail(SP_ERR_INTERNAL, "prove_cubic_with_three_inputs (host tables): eq weights");
    fe_t sums[3];
    {  // the three tables are separate arrays here: the evaluation takes base pointers
      HostCubicEval q;
      q.a = ha;
      q.b = hb;
      q.c = hc;
      q.E = E;
      q.hn = hn;
      const unsigned np = host_parts(hn, 12);
      if (np > 1) sp::WalkPool::get().run(np, host_cubic_part, &q);
      else host_cubic_part(&q, 0, 1);
      for (int k = 0; k < 3; ++k) sums[k] = fe_zero();
      for (unsigned pp = 0; pp < np; ++pp)
        for (int k = 0; k < 3; ++k) sums[k] = fe_add<S>(sums[k], q.out[pp][k]);
    }
    const fe_t t0 = sums[0], tinf = sums[1];
    const fe_t l_1_p = fe_mul<S>(fe_add<S>(eq0, slope), p);
    fe_t s_0, s_1, s_leading, s_m1;
    if (!fe_is_zero(l_1_p)) {  // derive_from_claim (src/sumcheck.rs:1276-1324), the division by l(1) p = tau p through 1 / tau
      s_0 = fe_mul<S>(fe_mul<S>(eq0, p), t0);
      s_1 = fe_sub<S>(claim, s_0);
      s_leading = fe_mul<S>(fe_mul<S>(slope, p), tinf);
      const fe_t two_sum = fe_add<S>(fe_dbl<S>(tinf), fe_dbl<S>(t0));
      s_m1 = fe_mul<S>(eqm1, fe_sub<S>(fe_mul<S>(p, two_sum), fe_mul<S>(s_1, inv_tau[rnd - 1])));
    } else {  // fallback_three_inputs (:1327-1396)
      s_0 = fe_mul<S>(fe_mul<S>(eq0, p), t0);
      s_1 = fe_sub<S>(claim, s_0);
      s_leading = fe_mul<S>(fe_mul<S>(slope, p), tinf);
      s_m1 = fe_mul<S>(fe_mul<S>(eqm1, p), sums[2]);
    }
    const fe_t halfc = two_inv();
    UniPoly poly;
    poly.n = 4;
    poly.c[0] = s_0;
    poly.c[1] = fe_sub<S>(fe_mul<S>(fe_sub<S>(s_1, s_m1), halfc), s_leading);
    poly.c[2] = fe_sub<S>(fe_mul<S>(fe_add<S>(s_1, s_m1), halfc), s_0);
    poly.c[3] = s_leading;
    absorb_poly(tr->t, poly);
    fe_t r_i;
    if (!tr->t.squeeze<S>(lbl_c, 1, &r_i)) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
    const size_t ri = rnd - 1;
    store_fe(out_r + 4 * ri, r_i);
    store_fe(out_cpolys + 12 * ri, poly.c[0]);
    store_fe(out_cpolys + 12 * ri + 4, poly.c[2]);
    store_fe(out_cpolys + 12 * ri + 8, poly.c[3]);
    claim = poly_eval(poly, r_i);
    for (fe_t* T : {ha, hb, hc}) host_bind_tables(T, 0, 1, n, r_i);
    p = fe_mul<S>(p, fe_add<S>(fe_sub<S>(fe_sub<S>(one, tau), r_i), fe_dbl<S>(fe_mul<S>(r_i, tau))));
  }
  store_fe(out_final, ha[0]);
  store_fe(out_final + 4, hb[0]);
  store_fe(out_final + 8, hc[0]);
  return SP_OK;
}
int sp_sumcheck_quad_host(sp_ctx* c, const uint64_t claim_[4], size_t rounds, uint64_t* A, uint64_t* B, sp_transcript* tr, uint64_t* out_cpolys, uint64_t* out_r,
                          uint64_t out_final[8]) {
  (void)c;
  if (!claim_ || !A || !B || !tr || !out_cpolys || !out_r || !out_final || rounds == 0 || rounds > 24)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_quad (host tables): null argument or more than 2^24 elements");
  tr->join();
  sp::WalkPool::get().keep_hot(2000);
  const size_t N = (size_t)1 << rounds;
  fe_t *ha = reinterpret_cast<fe_t*>(A), *hb = reinterpret_cast<fe_t*>(B);
  const uint8_t lbl_c[1] = {'c'};
  fe_t claim = load_fe(claim_);
  for (size_t round = 0; round < rounds; ++round) {
    const size_t n = N >> round, half = n / 2;
    fe_t sums[2];
    host_quad_eval(ha, hb, half, sums);
    // BDDT: eval_2 = 2 claim - 3 eval_0 + 2 t_inf and its interpolation (src/sumcheck.rs:211-215, univariate.rs:84-93): c0 = eval_0, c2 = t_inf, c1 = claim - 2 eval_0 - t_inf
    UniPoly poly;
    poly.n = 3;
    poly.c[0] = sums[0];
    poly.c[1] = fe_sub<S>(fe_sub<S>(claim, fe_dbl<S>(sums[0])), sums[1]);
    poly.c[2] = sums[1];
    absorb_poly(tr->t, poly);
    fe_t r_i;
    if (!tr->t.squeeze<S>(lbl_c, 1, &r_i)) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
    store_fe(out_r + 4 * round, r_i);
    store_fe(out_cpolys + 8 * round, poly.c[0]);
    store_fe(out_cpolys + 8 * round + 4, poly.c[2]);
    claim = poly_eval(poly, r_i);
    host_bind_tables(ha, 0, 1, n, r_i);
    host_bind_tables(hb, 0, 1, n, r_i);
  }
  store_fe(out_final, ha[0]);
  store_fe(out_final + 4, hb[0]);
  return SP_OK;
}
int sp_sumcheck_cubic3_round0(sp_ctx* c, const uint64_t claim_[4], const uint64_t* taus_, size_t ell, sp_table* A, sp_table* B, sp_table* C, const sp_table* p0,
                              const sp_table* p1, sp_transcript* tr, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]) {
  if (!p0 || !p1 || ell == 0 || p0->len != ((size_t)1 << ell) / 2 || p1->len != p0->len)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (round-0 products): product tables must have 2^(ell-1) elements");
  uint64_t claim_io[4], p_io[4];
  memcpy(claim_io, claim_, 32);
  const fe_t one = fe_one<S>();
  store_fe(p_io, one);
  return cubic_impl(c, claim_io, p_io, taus_, ell, A, B, C, tr, nullptr, nullptr, nullptr, p0, p1, out_cpolys, out_r, out_final);
}
// the same with an observer called after every challenge has been handed to the device (as sp_sumcheck_quad_observed): lets the caller start work that
// needs only the first challenges — sp_poly_abc_begin — under the remaining rounds. p0 / p1 may be NULL (plain first evaluation).
int sp_sumcheck_cubic3_observed(sp_ctx* c, const uint64_t claim_[4], const uint64_t* taus_, size_t ell, sp_table* A, sp_table* B, sp_table* C, const sp_table* p0,
                                const sp_table* p1, sp_transcript* tr, sp_challenge_hook observe, void* user, uint64_t* out_cpolys, uint64_t* out_r,
                                uint64_t out_final[12]) {
  if ((p0 == nullptr) != (p1 == nullptr)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (observed): both product tables or neither");
  if (p0 && (ell == 0 || p0->len != ((size_t)1 << ell) / 2 || p1->len != p0->len))
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (round-0 products): product tables must have 2^(ell-1) elements");
  uint64_t claim_io[4], p_io[4];
  memcpy(claim_io, claim_, 32);
  const fe_t one = fe_one<S>();
  store_fe(p_io, one);
  return cubic_impl(c, claim_io, p_io, taus_, ell, A, B, C, tr, nullptr, nullptr, nullptr, p0, p1, out_cpolys, out_r, out_final, 0, observe, user);
}

// The same rounds on a SLICE of the tables (tables sharded on their last k variables, rank g holds Z[(j << k) | g]): the slice's
// sums are scaled by `scale` = eq(tau[ell..ell+k), bits of g) and combined across ranks by `reduce` before the claim algebra, so every rank
// derives the same polynomial and challenge. `claim_io` / `p_io` carry the running claim and eq(tau, r) product in and out, so the caller can
// finish the last k rounds on the gathered 2^k-element tables with a second call (scale = reduce = NULL there).
int sp_sumcheck_cubic3_sharded(sp_ctx* c, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus_, size_t ell, sp_table* A, sp_table* B, sp_table* C,
                               sp_transcript* tr, const uint64_t* scale_, sp_reduce_hook reduce, void* reduce_user, uint64_t* out_cpolys, uint64_t* out_r,
                               uint64_t out_final[12]) {
  return cubic_impl(c, claim_io, p_io, taus_, ell, A, B, C, tr, scale_, reduce, reduce_user, nullptr, nullptr, out_cpolys, out_r, out_final);
}
// The slice form with the tau-independent halves of the FIRST evaluation handed in (sp_multiply_vec_incremental_round0 on the slice's rows: the pairs
// (i, i + n/2) of a slice are the slice's own): round 1 reads 2 x 32 B a pair instead of 5 x 32. run_rounds = 0 / ell: all rounds (out_final written);
// otherwise the first run_rounds only, as sp_sumcheck_cubic3_sharded_partial.
int sp_sumcheck_cubic3_sharded_round0(sp_ctx* c, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus_, size_t ell, size_t run_rounds, sp_table* A, sp_table* B,
                                      sp_table* C, const sp_table* p0, const sp_table* p1, sp_transcript* tr, const uint64_t* scale_, sp_reduce_hook reduce,
                                      void* reduce_user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]) {
  if (!p0 || !p1 || ell == 0 || p0->len != ((size_t)1 << ell) / 2 || p1->len != p0->len)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (slice, round-0 products): product tables must have 2^(ell-1) elements");
  if (run_rounds > ell) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (slice, round-0 products): run_rounds <= ell");
  uint64_t unused[12];
  const bool all = run_rounds == 0 || run_rounds == ell;
  if (all && !out_final) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (slice, round-0 products): out_final");
  return cubic_impl(c, claim_io, p_io, taus_, ell, A, B, C, tr, scale_, reduce, reduce_user, p0, p1, out_cpolys, out_r, all ? out_final : unused, all ? 0 : run_rounds);
}
int sp_sumcheck_cubic3_sharded_partial(sp_ctx* c, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus_, size_t ell, size_t run_rounds, sp_table* A, sp_table* B,
                                       sp_table* C, sp_transcript* tr, const uint64_t* scale_, sp_reduce_hook reduce, void* reduce_user, uint64_t* out_cpolys,
                                       uint64_t* out_r) {
  if (run_rounds == 0 || run_rounds >= ell) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs (partial): 0 < run_rounds < ell");
  uint64_t unused[12];
  return cubic_impl(c, claim_io, p_io, taus_, ell, A, B, C, tr, scale_, reduce, reduce_user, nullptr, nullptr, out_cpolys, out_r, unused, run_rounds);
}
static int cubic_impl(sp_ctx* c, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus_, size_t ell, sp_table* A, sp_table* B, sp_table* C, sp_transcript* tr,
                      const uint64_t* scale_, sp_reduce_hook reduce, void* reduce_user, const sp_table* prod0, const sp_table* prod1, uint64_t* out_cpolys,
                      uint64_t* out_r, uint64_t out_final[12], size_t run_rounds, sp_challenge_hook observe, void* observe_user) {
  // run_rounds (0 = all): see quad_impl - stop after that many of the ell rounds, tables left at 2^(ell - run_rounds) elements
  if (tail_hand_n(true) > 16) sp::WalkPool::get().keep_hot(3000);
  tr->join();
  if (run_rounds > ell) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs: more rounds to run than variables");
  const bool stopped = run_rounds != 0 && run_rounds < ell;
  const size_t last_rnd = stopped ? run_rounds : ell;
  const uint64_t* claim_ = claim_io;
  const bool have_scale = scale_ != nullptr;
  const fe_t scale = have_scale ? load_fe(scale_) : fe_one<S>();
  const size_t N = (size_t)1 << ell;
  if (A->len != N || B->len != N || C->len != N) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs: tables must have 2^ell elements");
  if (ell == 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "prove_cubic_with_three_inputs: no rounds");
  const double tr_entry = round_trace() ? now_us() : 0;
  // EqSumCheckInstance::new (src/sumcheck.rs:956-1016)
  const size_t first_half = ell / 2, second_half = ell - first_half;
  std::vector<fe_t> taus(ell);
  for (size_t i = 0; i < ell; ++i) taus[i] = load_fe(taus_ + 4 * i);
  // device pyramids: left over taus[1..first_half), right over taus[first_half..ell)
  const size_t nleft = first_half > 0 ? first_half - 1 : 0;
  size_t pyr_left = (size_t)2 << nleft, pyr_right = (size_t)2 << second_half;
  const size_t chunk = 256 * spk::EVAL_PPT;
  size_t max_blocks = (N / 2 + chunk - 1) / chunk + 1;
  if (max_blocks * 3 * 32 < (N / 4 / 64) * 72 + 64) max_blocks = ((N / 4 / 64) * 72 + 64 + 95) / 96;  // room for lazy wave partials too
  if (nleft > 16 || second_half > 16) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sum-check over more than 2^32 rows");
  int rc = c->ensure_scratch(max_blocks * 3 + 32);  // the block partials
  if (rc) return rc;
  fe_t* d_part = c->d_scratch;
  // The two eq pyramids live in a buffer of their own (not in the scratch every launch shares): they are built on the eq stream beside whatever the
  // main stream still has queued in front of this sum-check, so nothing else may be using their memory then. Its only readers are the kernels of a
  // cubic sum-check on this context, and the previous one has delivered its last result to the host before this call can be made.
  if (c->cubic_eq_elems < pyr_left + pyr_right) {
    if (c->d_cubic_eq) {
      SP_HIP(sp::stream_sync(c->stream));  // (growing: once per context and size)
      hipFree(c->d_cubic_eq);
    }
    c->d_cubic_eq = nullptr;
    c->cubic_eq_elems = 0;
    SP_HIP(hipMalloc((void**)&c->d_cubic_eq, (pyr_left + pyr_right) * sizeof(fe_t)));
    c->cubic_eq_elems = pyr_left + pyr_right;
  }
  fe_t* d_pl = c->d_cubic_eq;
  fe_t* d_pr = d_pl + pyr_left;
  {
    spk::EqPairArgs ea;
    for (size_t i = 0; i < nleft; ++i) ea.v[0][i] = taus[1 + i];
    for (size_t i = 0; i < second_half; ++i) ea.v[1][i] = taus[first_half + i];
    ea.m[0] = (int)nleft;
    ea.m[1] = (int)second_half;
    ea.out[0] = d_pl;
    ea.out[1] = d_pr;
    // On the eq stream, beside whatever the main stream still has queued in front of this sum-check (the matrix-vector product and its round-0
    // products at the start of a prove: ~70 us): the pyramids need nothing but the taus; the main stream waits for them before the first evaluation.
    if (!c->cubic_ev) SP_HIP(hipEventCreateWithFlags(&c->cubic_ev, hipEventDisableTiming));
    hipLaunchKernelGGL(spk::k_eq_levels_pair, dim3(2), dim3(1024), 0, c->stream_eq, ea);
    SP_HIP(hipEventRecord(c->cubic_ev, c->stream_eq));
    SP_HIP(hipStreamWaitEvent(c->stream, c->cubic_ev, 0));
  }

  // eq tables of round `rnd` (1-based, src/sumcheck.rs:1011) for `half` pairs
  struct EqSel {
    const fe_t* eq_in;
    const fe_t* eq_out;
    int s, mode;
  };
  auto select_eq = [&](size_t rnd) {
    EqSel e;
    if (rnd < first_half) {  // poly_eqs_first_half (:1407-1420)
      e.eq_out = d_pl + spk::eq_level_offset((int)(first_half - rnd));
      e.eq_in = d_pr + spk::eq_level_offset((int)second_half);
      e.s = (int)second_half;
      e.mode = (((size_t)1 << e.s) >= chunk) ? 1 : 2;
    } else {  // poly_eq_right_last_half (:1422-1428)
      e.eq_in = d_pr + spk::eq_level_offset((int)(ell - rnd));
      e.eq_out = nullptr;
      e.s = 63;
      e.mode = 0;
    }
    return e;
  };
  auto launch_eval = [&](size_t rnd, bool with_m1) {
    const size_t half = A->len / 2;
    const EqSel e = select_eq(rnd);
    dim3 g((unsigned)((half + chunk - 1) / chunk)), b(256);
    const unsigned seq = next_seq(c);
#define SP_LAUNCH_EVAL(MODE, M1) \
  hipLaunchKernelGGL((spk::k_eval_cubic<MODE, M1>), g, b, 0, c->stream, A->d, B->d, C->d, half, e.eq_in, e.eq_out, e.s, d_part, c->d_pinned, seq)
    if (!with_m1) {
      if (e.mode == 0) SP_LAUNCH_EVAL(0, false);
      else if (e.mode == 1) SP_LAUNCH_EVAL(1, false);
      else SP_LAUNCH_EVAL(2, false);
    } else {
      if (e.mode == 0) SP_LAUNCH_EVAL(0, true);
      else if (e.mode == 1) SP_LAUNCH_EVAL(1, true);
      else SP_LAUNCH_EVAL(2, true);
    }
#undef SP_LAUNCH_EVAL
    return (size_t)g.x;
  };

  fe_t claim = load_fe(claim_);
  bool in_tail = false;  // the persistent tail kernel owns the remaining rounds
  unsigned tail_nb0 = 1;  // blocks of the resident launch
  bool host_mode = false;  // ... and has handed the tables over: the remaining rounds run on the host (kernels_poly.hpp tail_hand_over)
  std::vector<fe_t> hT, hE;  // host tables after a hand-over (A | B | C, n0 entries each) and the round's eq weights
  HostEqLevels heq;           // ... of every host round, built once (under the first wait for the device)
  size_t n0 = 0;
  unsigned hand_seq = 0;
  unsigned last_answered = 0;
  TailLease lease;
  AheadGuard guard(c);
  *reinterpret_cast<volatile uint32_t*>(c->h_pinned + spk::TAIL_ERR_ELEM) = 0;
  fe_t eval_eq_left = load_fe(p_io);
  const fe_t one = fe_one<S>();
  // slice sums -> batch sums (no-op for an unsharded call)
  auto combine = [&](fe_t* sums, int k) -> int {
    if (have_scale)
      for (int i = 0; i < k; ++i) sums[i] = fe_mul<S>(sums[i], scale);
    if (reduce) {
      int hrc = reduce(reduce_user, reinterpret_cast<uint64_t*>(sums), (size_t)k);
      if (hrc) return fail(hrc, "sum-check: the reduce hook failed");
    }
    return SP_OK;
  };
  const uint8_t lbl_c[1] = {'c'};
  // What follows round `rnd`'s challenge r (see the quadratic loop): the tail's bookkeeping, the tail launch, a fused bind + evaluation of round
  // rnd + 1, or the last plain bind. r == nullptr issues it ahead of the challenge. Returns 1 = issued, 0 = needs r on the host, < 0 error.
  auto issue = [&](size_t rnd, const fe_t* r, unsigned answers) -> int {
    const bool ahead = r == nullptr;
    const fe_t rv = r ? *r : fe_zero();
    if (in_tail) {
      sp::after_bind(A);
      sp::after_bind(B);
      sp::after_bind(C);
      if (rnd < ell) {
        next_seq(c);
        c->pending_slots = tail_blocks(A->len, true) > 1 ? tail_nb0 : 1u;  // every block of the launch publishes while more than one would (the kernel's local regime)
      }
      return 1;
    }
    if (rnd >= last_rnd) {
      if (ahead) return 0;
      sp_table* tabs[3] = {A, B, C};
      int rc2 = launch_bind(c, tabs, 3, rv);
      return rc2 ? rc2 : 1;
    }
    const bool gated = ahead && !launch_ahead_ok(c, A->len);
    if (gated && !gate_ok(c)) return 0;
    if (!stopped && tail_enabled() && A->len <= TAIL_MAX_LEN && tail_blocks(A->len / 2, true) <= (unsigned)spk::HOST_SUM_MAX_BLOCKS && lease.take(tail_blocks(A->len / 2, true))) {
      spk::TailArgs ta;
      ta.A = A->d;
      ta.B = B->d;
      ta.C = C->d;
      ta.len = A->len;
      ta.r0 = rv;
      ta.eq_pl = d_pl;
      ta.eq_pr = d_pr;
      ta.ell = (int)ell;
      ta.first_half = (int)first_half;
      ta.rnd0 = (int)rnd + 1;
      ta.mail = c->d_mail;
      ta.mirror = c->d_mail_mirror;
      ta.r0_from_mail = ahead ? 1 : 0;
      ta.mapped = c->d_pinned;
      ta.seq0 = next_seq(c);
      ta.hand_n = tail_hand_n(true);
      tail_nb0 = tail_blocks(A->len / 2, true);
      hipLaunchKernelGGL((spk::k_sumcheck_tail<true>), dim3(tail_nb0), dim3(spk::TAIL_THREADS), 0, c->stream, ta);
      in_tail = true;
      sp::after_bind(A);
      sp::after_bind(B);
      sp::after_bind(C);
      c->pending_slots = tail_blocks(A->len, true);
      return 1;
    }
    // K1 fused with next round's K2: bind with r, evaluate round rnd+1 from registers
    const spk::MailRef mref = gated ? gate_launch(c, answers) : mail_ref(c, ahead, answers);
    const size_t q = A->len / 4;
    const EqSel e = select_eq(rnd + 1);
    dim3 g((unsigned)((q + chunk - 1) / chunk)), b(256);
    const unsigned seq = next_seq(c);
    if (q >= STREAM_MIN_Q && (e.mode == 0 || (e.mode == 1 && e.s >= 8))) {
      // factored mode: 2^(s-8) consecutive blocks share one x_out; single-table mode: any grouping, no factor
      const spk::LazyOut lp = lazy_out(c, reinterpret_cast<spk::lazy9_t*>(d_part), q / 256, e.mode == 1 ? e.s - 8 : 2, e.mode == 1 ? e.eq_out : (const fe_t*)nullptr, seq);
      {
        const dim3 gs((unsigned)(q / 256));
        const uint64_t bytes = 48ull * A->len * 3;
        fe_t *pa = A->d, *pb = B->d, *pc = C->d;
        // tables of 2^23 and more (3 x 256 MiB: past the 256 MiB Infinity Cache) are accounted separately: their GB/s is unambiguously HBM
        const char* kname = A->len >= ((size_t)1 << 23) ? "bind_stream_cubic_hbm" : "bind_stream_cubic";
        const bool polls = ahead && !gated;
        if (e.mode == 0 && gated) c->timed_kernel(kname, bytes, spk::k_bind_eval_cubic_stream<0, false, true>, gs, b, pa, pb, pc, q, rv, e.eq_in, e.s, lp, mref);
        else if (gated) c->timed_kernel(kname, bytes, spk::k_bind_eval_cubic_stream<1, false, true>, gs, b, pa, pb, pc, q, rv, e.eq_in, e.s, lp, mref);
        else if (e.mode == 0 && polls) c->timed_kernel(kname, bytes, spk::k_bind_eval_cubic_stream<0, true>, gs, b, pa, pb, pc, q, rv, e.eq_in, e.s, lp, mref);
        else if (e.mode == 0) c->timed_kernel(kname, bytes, spk::k_bind_eval_cubic_stream<0, false>, gs, b, pa, pb, pc, q, rv, e.eq_in, e.s, lp, mref);
        else if (polls) c->timed_kernel(kname, bytes, spk::k_bind_eval_cubic_stream<1, true>, gs, b, pa, pb, pc, q, rv, e.eq_in, e.s, lp, mref);
        else c->timed_kernel(kname, bytes, spk::k_bind_eval_cubic_stream<1, false>, gs, b, pa, pb, pc, q, rv, e.eq_in, e.s, lp, mref);
      }
      sum_lazy_launch(c, lp, q / 256);
      sp::after_bind(A);
      sp::after_bind(B);
      sp::after_bind(C);
    } else {
      c->timed("bind", 48ull * A->len * 3, [&] {
        if (e.mode == 0)
          hipLaunchKernelGGL((spk::k_bind_eval_cubic<0>), g, b, 0, c->stream, A->d, B->d, C->d, q, rv, e.eq_in, e.eq_out, e.s, d_part, c->d_pinned, seq, mref);
        else if (e.mode == 1)
          hipLaunchKernelGGL((spk::k_bind_eval_cubic<1>), g, b, 0, c->stream, A->d, B->d, C->d, q, rv, e.eq_in, e.eq_out, e.s, d_part, c->d_pinned, seq, mref);
        else
          hipLaunchKernelGGL((spk::k_bind_eval_cubic<2>), g, b, 0, c->stream, A->d, B->d, C->d, q, rv, e.eq_in, e.eq_out, e.s, d_part, c->d_pinned, seq, mref);
      });
      sp::after_bind(A);
      sp::after_bind(B);
      sp::after_bind(C);
      reduce_partials_launch(c, g.x, 2);
    }
    return 1;
  };
  // round 1 sums from a plain evaluation pass; later rounds get theirs from the fused bind+eval of the previous round
  {
    const size_t half = A->len / 2;
    const EqSel e1 = select_eq(1);
    if (prod0 && prod1 && half % 256 == 0 && (e1.mode == 0 || (e1.mode == 1 && e1.s >= 8))) {
      // the per-pair products came with the matrix-vector product: weight them with the eq tables (2 x 32 B per pair instead of 5 x 32 B)
      const unsigned seq = next_seq(c);
      // four chunks of 256 pairs per block where the table and, in factored mode, the x_out group (2^s pairs) hold them; else one
      const bool four = half % 1024 == 0 && half >= ((size_t)1 << 16) && (e1.mode == 0 || e1.s >= 10);
      const size_t nblk = four ? half / 1024 : half / 256;
      const int gsh = four ? 10 : 8;  // log2 of the pairs of a block
      const spk::LazyOut lp = lazy_out(c, reinterpret_cast<spk::lazy9_t*>(d_part), nblk, e1.mode == 1 ? e1.s - gsh : 2, e1.mode == 1 ? e1.eq_out : (const fe_t*)nullptr, seq);
      const dim3 gs((unsigned)nblk), bs(256);
      c->timed("eval_cubic", 64ull * half, [&] {
        if (e1.mode == 0 && four) hipLaunchKernelGGL((spk::k_eval_products_stream<0, 4>), gs, bs, 0, c->stream, prod0->d, prod1->d, e1.eq_in, e1.s, lp);
        else if (e1.mode == 0) hipLaunchKernelGGL((spk::k_eval_products_stream<0, 1>), gs, bs, 0, c->stream, prod0->d, prod1->d, e1.eq_in, e1.s, lp);
        else if (four) hipLaunchKernelGGL((spk::k_eval_products_stream<1, 4>), gs, bs, 0, c->stream, prod0->d, prod1->d, e1.eq_in, e1.s, lp);
        else hipLaunchKernelGGL((spk::k_eval_products_stream<1, 1>), gs, bs, 0, c->stream, prod0->d, prod1->d, e1.eq_in, e1.s, lp);
      });
      sum_lazy_launch(c, lp, nblk);
    } else if (half >= STREAM_MIN_Q && half % 256 == 0 && (e1.mode == 0 || (e1.mode == 1 && e1.s >= 8))) {
      const unsigned seq = next_seq(c);
      const spk::LazyOut lp = lazy_out(c, reinterpret_cast<spk::lazy9_t*>(d_part), half / 256, e1.mode == 1 ? e1.s - 8 : 2, e1.mode == 1 ? e1.eq_out : (const fe_t*)nullptr, seq);
      const dim3 gs((unsigned)(half / 256)), bs(256);
      c->timed("eval_cubic", 160ull * half, [&] {
        if (e1.mode == 0) hipLaunchKernelGGL((spk::k_eval_cubic_stream<0>), gs, bs, 0, c->stream, A->d, B->d, C->d, half, e1.eq_in, e1.s, lp);
        else hipLaunchKernelGGL((spk::k_eval_cubic_stream<1>), gs, bs, 0, c->stream, A->d, B->d, C->d, half, e1.eq_in, e1.s, lp);
      });
      sum_lazy_launch(c, lp, half / 256);
    } else {
      size_t blocks = 0;
      c->timed("eval_cubic", 160ull * (A->len / 2), [&] { blocks = launch_eval(1, false); });
      reduce_partials_launch(c, blocks, 2);
    }
  }
  // 1 / tau_k for every round by one inversion (Montgomery's trick); zeros stay zero (those rounds take the three-sum fallback)
  std::vector<fe_t> inv_tau(ell, fe_zero());
  {
    std::vector<fe_t> pref(ell);
    fe_t run = one;
    for (size_t i = 0; i < ell; ++i) {
      pref[i] = run;
      if (!fe_is_zero(taus[i])) run = fe_mul<S>(run, taus[i]);
    }
    fe_t inv = fe_inv_vartime<S>(run);
    for (size_t i = ell; i-- > 0;) {
      if (fe_is_zero(taus[i])) continue;
      inv_tau[i] = fe_mul<S>(inv, pref[i]);
      inv = fe_mul<S>(inv, taus[i]);
    }
  }
  if (round_trace()) fprintf(stderr, "cubic setup %7.1f us\n", now_us() - tr_entry);
  for (size_t rnd = 1; rnd <= last_rnd; ++rnd) {
    // host work that only needs earlier challenges runs while the device computes this round's sums
    const double tr_top = round_trace() ? now_us() : 0;
    const fe_t tau = taus[rnd - 1];
    const fe_t eq0 = fe_sub<S>(one, tau);     // eq(tau, 0)
    const fe_t slope = fe_sub<S>(tau, eq0);   // 2 tau - 1
    const fe_t eqm1 = fe_sub<S>(eq0, slope);  // 2 - 3 tau
    const fe_t p = eval_eq_left;
    const fe_t l_0_p = fe_mul<S>(eq0, p);
    const fe_t l_1_p = fe_mul<S>(fe_add<S>(eq0, slope), p);
    const bool invertible = !fe_is_zero(l_1_p);
    // this round's sums are in flight: remember how to wait for them, then issue the next launch ahead of the challenge where that is possible.
    // Not when tau * p vanishes: that round re-evaluates with a third sum (fallback_three_inputs), which must not queue behind a waiting kernel.
    const unsigned wait_seq = c->result_seq, wait_slots = c->pending_slots;
    const bool wait_resident = in_tail;
    if (wait_resident && !host_mode && spk::tail_hand_over(A->len, tail_hand_n(true))) {
      // HAND-OVER (kernels_poly.hpp tail_hand_over): the resident kernel sent the tables themselves; this round and the ones after it run here
      n0 = A->len;
      hT.resize(3 * n0);
      if ((rc = wait_hand_over(c, wait_seq, (int)(3 * n0), hT.data()))) return rc;
      host_mode = true;
      hand_seq = wait_seq;
    }
    if (!host_mode && wait_resident && spk::tail_double(true, A->len, tail_hand_n(true))) {
      // TWO ROUNDS IN THIS TRIP (kernels_poly.hpp TAIL_WIDE_VALS). W[9..12) = t(0), t_inf, t(-1) of round rnd; W[0..9) = the coefficient sums of
      // round rnd + 1 in this round's challenge r: t'(0) = W0 + r (W1 - W0 - W2) + r^2 W2, t'_inf = W3 + r (W5 - W3 - W4) + r^2 W4,
      // t'(-1) = W6 + r (W8 - W6 - W7) + r^2 W7 (used by the fallback_three_inputs form only, as in the one-round path).
      fe_t W[12];
      const double trd = round_trace() ? now_us() : 0;
      const size_t len_d = A->len;
      if ((rc = wait_wide(c, wait_seq, spk::TAIL_DOUBLE_SUMS_CUBIC, W))) return rc;
      if ((rc = combine(W, 12))) return rc;
      fe_t rr[2];
      for (int h2 = 0; h2 < 2; ++h2) {
        const size_t rd = rnd + h2;
        const fe_t tau_h = taus[rd - 1];
        const fe_t q0 = fe_sub<S>(one, tau_h), sl = fe_sub<S>(tau_h, q0), qm1 = fe_sub<S>(q0, sl);
        const fe_t pp = eval_eq_left;
        fe_t t0, tinf, tm1;
        if (h2 == 0) {
          t0 = W[9];
          tinf = W[10];
          tm1 = W[11];
        } else {
          const fe_t r = rr[0];
          auto at_r = [&](const fe_t& k0, const fe_t& mid, const fe_t& k2) {  // k0 + r (mid - k0 - k2) + r^2 k2
            return fe_add<S>(k0, fe_mul<S>(r, fe_add<S>(fe_sub<S>(fe_sub<S>(mid, k0), k2), fe_mul<S>(r, k2))));
          };
          t0 = at_r(W[0], W[1], W[2]);
          tinf = at_r(W[3], W[5], W[4]);
          tm1 = at_r(W[6], W[8], W[7]);
        }
        // derive_from_claim (:1276-1324) when l(1) p = tau p is invertible - it uses the CLAIM, so a claim that is not the honest sum gives another
        // polynomial than the three-sum form, and the reference's choice must be followed - else fallback_three_inputs (:1327-1396) with t(-1)
        const fe_t s_0 = fe_mul<S>(fe_mul<S>(q0, pp), t0);
        const fe_t s_1 = fe_sub<S>(claim, s_0);
        const fe_t s_leading = fe_mul<S>(fe_mul<S>(sl, pp), tinf);
        fe_t s_m1;
        if (!fe_is_zero(fe_mul<S>(tau_h, pp))) {
          const fe_t two_sum = fe_add<S>(fe_dbl<S>(tinf), fe_dbl<S>(t0));
          s_m1 = fe_mul<S>(qm1, fe_sub<S>(fe_mul<S>(pp, two_sum), fe_mul<S>(s_1, inv_tau[rd - 1])));
        } else {
          s_m1 = fe_mul<S>(fe_mul<S>(qm1, pp), tm1);
        }
        const fe_t halfc = two_inv();
        const fe_t c1 = fe_sub<S>(fe_mul<S>(fe_sub<S>(s_1, s_m1), halfc), s_leading);
        const fe_t c2 = fe_sub<S>(fe_mul<S>(fe_add<S>(s_1, s_m1), halfc), s_0);
        // (s(0), s_leading, s(-1)) -> coefficients: the reference goes through (eval_0, eval_2, eval_3) and an interpolation (:1303-1320,
        // univariate.rs:102-118); c1 and c2 above ARE the coefficients of that polynomial, the same field elements without the detour
        UniPoly poly;
        poly.n = 4;
        poly.c[0] = s_0;
        poly.c[1] = c1;
        poly.c[2] = c2;
        poly.c[3] = s_leading;
        absorb_poly(tr->t, poly);
        if (!tr->t.squeeze<S>(lbl_c, 1, &rr[h2])) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
        const size_t ri = rd - 1;
        tail_post_challenge(c, rr[h2], wait_seq + h2);  // (the first at once: the kernel's first bind runs under the second round's host step)
        store_fe(out_r + 4 * ri, rr[h2]);
        store_fe(out_cpolys + 12 * ri, poly.c[0]);
        store_fe(out_cpolys + 12 * ri + 4, poly.c[2]);
        store_fe(out_cpolys + 12 * ri + 8, poly.c[3]);
        claim = poly_eval(poly, rr[h2]);
        eval_eq_left = fe_mul<S>(eval_eq_left, fe_add<S>(fe_sub<S>(fe_sub<S>(one, tau_h), rr[h2]), fe_dbl<S>(fe_mul<S>(rr[h2], tau_h))));
      }
      for (int k = 0; k < 2; ++k) {  // the kernel binds twice; its next result (if rounds are left) is tagged wait_seq + 2
        sp::after_bind(A);
        sp::after_bind(B);
        sp::after_bind(C);
        next_seq(c);
      }
      c->pending_slots = rnd + 1 < ell ? 1u : 0u;
      last_answered = wait_seq + 1;
      guard.armed = rnd + 1 < ell;
      if (observe)
        for (int k = 0; k < 2; ++k) {
          uint64_t rw[4];
          store_fe(rw, rr[k]);
          observe(observe_user, rnd - 1 + k, rw);
        }
      if (round_trace()) fprintf(stderr, "cubic rounds %2zu+%2zu len %8zu tail 2 wait+host %7.1f us\n", rnd, rnd + 1, len_d, now_us() - trd);
      ++rnd;
      continue;
    }
    int issued = 0;
    if (!host_mode && (in_tail || (c->mail_dev && invertible && !reduce))) {  // not with a reduce hook: it may run device work (a collective) beside the waiting kernel
      issued = issue(rnd, nullptr, wait_seq);
      if (issued < 0) return issued;
      guard.armed = issued != 0;
    }
    fe_t sums[3];
    const double tr0 = round_trace() ? now_us() : 0;
    const size_t len_now = A->len;
    if (host_mode) {
      // evaluation_points_cubic_with_three_inputs + t(-1) (src/sumcheck.rs:1025-1156, :1327-1396) on the host tables: pairs (x, x + n / 2) weighted with
      // E(rnd, x) = eq(taus[rnd ..), x) - what the split tables of either half multiply out to (select_eq)
      const size_t n = len_now, hn = n / 2;
      const fe_t *ha = hT.data(), *hb = ha + n0, *hc = hb + n0;
      const fe_t* E = heq.level(rnd);
      if (!E) {  // (not built: a hand-over the first wait did not foresee)
        hE.assign(1, one);
        for (size_t i = rnd; i < ell; ++i) {  // first variable = most significant bit of x
          const size_t m = hE.size();
          hE.resize(2 * m);
          for (size_t j = m; j-- > 0;) {
            const fe_t hi = fe_mul<S>(hE[j], taus[i]);
            hE[2 * j] = fe_sub<S>(hE[j], hi);
            hE[2 * j + 1] = hi;
          }
        }
        E = hE.data();
      }
      host_cubic_eval(ha, hb, hc, E, hn, sums);
    } else {
      const unsigned cur_seq = c->result_seq, cur_slots = c->pending_slots;
      c->result_seq = wait_seq;
      c->pending_slots = wait_slots;
      if (!heq.built && tail_hand_n(true) > 16) heq.build(taus.data(), ell, tail_hand_n(true) / 2);  // under the device's round: the weights of the host's rounds
      rc = reduce_partials_wait(c, wait_resident ? 3 : 2, sums, wait_resident || issued != 0);  // never a stream synchronise with a launch waiting at the mailbox
      if (issued) {  // back to the state of the launch issued ahead
        c->result_seq = cur_seq;
        c->pending_slots = cur_slots;
      }
      if (rc) return rc;
    }
    const double tr1 = round_trace() ? now_us() : 0;
    if ((rc = combine(sums, wait_resident ? 3 : 2))) return rc;
    const fe_t t0 = sums[0], tinf = sums[1];
    // derive_from_claim (:1276-1324)
    fe_t s_0, s_1, s_leading, s_m1;
    if (wait_resident && !invertible) {  // fallback_three_inputs (:1327-1396): the resident tail kernel always delivers t(-1) as its third sum
      s_0 = fe_mul<S>(fe_mul<S>(eq0, p), t0);
      s_1 = fe_sub<S>(claim, s_0);
      s_leading = fe_mul<S>(fe_mul<S>(slope, p), tinf);
      s_m1 = fe_mul<S>(fe_mul<S>(eqm1, p), sums[2]);
    } else if (invertible) {
      // t(1) = s(1) / (l(1) p), t(-1) = 2 t_inf + 2 t(0) - t(1), s(-1) = l(-1) p t(-1) (:1276-1324). With l(1) = tau the division by p cancels:
      // l(-1) p t(1) = l(-1) s(1) / tau — the same field element, from an inverse known before the first round (inv_tau) instead of a fresh
      // inversion per round (6.7 us of host time each, more than a tail round takes on the device).
      s_0 = fe_mul<S>(l_0_p, t0);
      s_1 = fe_sub<S>(claim, s_0);
      s_leading = fe_mul<S>(fe_mul<S>(slope, p), tinf);
      const fe_t two_sum = fe_add<S>(fe_dbl<S>(tinf), fe_dbl<S>(t0));
      s_m1 = fe_mul<S>(eqm1, fe_sub<S>(fe_mul<S>(p, two_sum), fe_mul<S>(s_1, inv_tau[rnd - 1])));
    } else {  // fallback_three_inputs (:1327-1396): third sum t(-1) computed directly on the (still unbound) tables
      size_t blocks = 0;
      c->timed("eval_cubic", 192ull * (A->len / 2), [&] { blocks = launch_eval(rnd, true); });
      rc = reduce_partials(c, blocks, 3, sums);
      if (rc) return rc;
      if ((rc = combine(sums, 3))) return rc;
      s_0 = fe_mul<S>(fe_mul<S>(eq0, p), t0);
      s_1 = fe_sub<S>(claim, s_0);
      s_leading = fe_mul<S>(fe_mul<S>(slope, p), tinf);
      s_m1 = fe_mul<S>(fe_mul<S>(eqm1, p), sums[2]);
    }
    // (s(0), s_leading, s(-1)) -> (eval_0, eval_2, eval_3) (:1303-1320)
    const fe_t halfc = two_inv();
    const fe_t c1 = fe_sub<S>(fe_mul<S>(fe_sub<S>(s_1, s_m1), halfc), s_leading);
    const fe_t c2 = fe_sub<S>(fe_mul<S>(fe_add<S>(s_1, s_m1), halfc), s_0);
    // (s(0), s_leading, s(-1)) -> coefficients: the reference goes through (eval_0, eval_2, eval_3) and an interpolation (:1303-1320,
    // univariate.rs:102-118); c1 and c2 above ARE the coefficients of that polynomial, the same field elements without the detour
    UniPoly poly;
    poly.n = 4;
    poly.c[0] = s_0;
    poly.c[1] = c1;
    poly.c[2] = c2;
    poly.c[3] = s_leading;
    absorb_poly(tr->t, poly);
    fe_t r_i;
    if (!tr->t.squeeze<S>(lbl_c, 1, &r_i)) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
    const size_t ri = rnd - 1;
    last_answered = wait_seq;
    if (host_mode) {
      host_bind_tables(hT.data(), n0, 3, len_now, r_i);
      sp::after_bind(A);
      sp::after_bind(B);
      sp::after_bind(C);
    } else if (issued) {
      tail_post_challenge(c, r_i, wait_seq);  // whatever was issued ahead is waiting for exactly this: first, the book-keeping below runs under it
    } else {
      issued = issue(rnd, &r_i, wait_seq);
      if (issued < 0) return issued;
    }
    store_fe(out_r + 4 * ri, r_i);
    store_fe(out_cpolys + 12 * ri, poly.c[0]);
    store_fe(out_cpolys + 12 * ri + 4, poly.c[2]);
    store_fe(out_cpolys + 12 * ri + 8, poly.c[3]);
    claim = poly_eval(poly, r_i);
    guard.armed = in_tail && (host_mode || rnd < ell);  // the resident kernel now waits for the next challenge (after a hand-over: for the final claims)
    if (observe) {  // after the device has been given this round's challenge: the observer's work runs under the next round
      uint64_t rw[4];
      store_fe(rw, r_i);
      observe(observe_user, ri, rw);
    }
    // bound (:1399-1405): p *= 1 - tau - r + 2 r tau
    eval_eq_left = fe_mul<S>(eval_eq_left, fe_add<S>(fe_sub<S>(fe_sub<S>(one, tau), r_i), fe_dbl<S>(fe_mul<S>(r_i, tau))));
    if (round_trace())
      fprintf(stderr, "cubic round %2zu len %8zu tail %d pre %5.1f wait %7.1f us host %6.1f us\n", rnd, len_now, (int)wait_resident, tr0 - tr_top, tr1 - tr0, now_us() - tr1);
  }
  if (stopped) {
    store_fe(claim_io, claim);
    store_fe(p_io, eval_eq_left);
    return tail_check(c);
  }
  if (host_mode) {  // the final claims are here; the kernel, waiting at the mailbox, stores them into element 0 of its tables and leaves
    for (int t = 0; t < 3; ++t) {
      store_fe(out_final + 4 * t, hT[t * n0]);
      tail_post_challenge(c, hT[t * n0], hand_seq + (unsigned)t);
    }
    c->result_seq = hand_seq + 2;  // the three lines took sequence numbers of their own
    guard.armed = false;
  } else if (in_tail) {  // the resident kernel hands the final claims over itself
    fe_t fin[3];
    long spins = 0;
    if ((rc = wait_slot(c, c->h_pinned + spk::TAIL_FINAL_ELEM, last_answered, 3, fin, true, &spins))) return rc;
    store_fe(out_final, fin[0]);
    store_fe(out_final + 4, fin[1]);
    store_fe(out_final + 8, fin[2]);
  } else {
    rc = sp_table_read(c, A, 0, 1, out_final);
    if (rc) return rc;
    rc = sp_table_read(c, B, 0, 1, out_final + 4);
    if (rc) return rc;
    rc = sp_table_read(c, C, 0, 1, out_final + 8);
    if (rc) return rc;
  }
  store_fe(claim_io, claim);
  store_fe(p_io, eval_eq_left);
  if (round_trace()) fprintf(stderr, "cubic total %7.1f us\n", now_us() - tr_entry);
#ifdef SP_TAIL_TRACE
  if (round_trace()) {  // the last eight one-round steps of the resident kernel: stations 0 top, 1 challenge seen, 2 bound, 3 products, 4 wave sums, 5 block barrier, 6 published
    const volatile unsigned long long* tt = reinterpret_cast<const volatile unsigned long long*>(c->h_pinned + 32);
    for (int k = 0; k < 8; ++k) {
      fprintf(stderr, "  tail step slot %d:", k);
      for (int i = 1; i < 7; ++i) fprintf(stderr, " %+6.2f", (double)(long long)(tt[8 * k + i] - tt[8 * k + i - 1]) / 100.0);
      fprintf(stderr, " us | top->top of next %6.2f\n", (double)(long long)(tt[8 * ((k + 1) & 7)] - tt[8 * k]) / 100.0);
    }
  }
#endif
  return tail_check(c);  // also set by a kernel launched ahead that gave up waiting for its challenge
}

}  // extern "C"
