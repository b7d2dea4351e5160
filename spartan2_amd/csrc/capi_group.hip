// libspartan_hip.so — group / MSM / Hyrax entry points of include/spartan_hip.h.
// Device: digit sort, bucket accumulation, per-window weighted reduction, binary row sums, fixed-base lookups,
// row-matrix product. Host (inside the library): window Horner, adding the blind term, batch normalisation.
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "core.hpp"
#include "kernels_msm.hpp"

using sp::fail;
typedef FqP S;

#include "group_common.hpp"
#include "walk_pool.hpp"
#include <sys/mman.h>

namespace {

// Montgomery's trick on the host (DlogGroup::batch_affine, src/provider/traits.rs:194-198)
void normalize_batch(const std::vector<jac_t>& pts, aff_t* out) {
  size_t n = pts.size();
  std::vector<fe_t> pref(n);
  fe_t acc = fe_one<B>();
  for (size_t i = 0; i < n; ++i) {
    pref[i] = acc;
    if (!jac_is_identity(pts[i])) acc = fe_mul<B>(acc, pts[i].z);
  }
  fe_t inv = fe_inv<B>(acc);
  for (size_t i = n; i-- > 0;) {
    if (jac_is_identity(pts[i])) {
      out[i].x = fe_zero();
      out[i].y = fe_zero();
      continue;
    }
    fe_t zi = fe_mul<B>(inv, pref[i]);
    inv = fe_mul<B>(inv, pts[i].z);
    fe_t zi2 = fe_sqr<B>(zi);
    out[i].x = fe_mul<B>(pts[i].x, zi2);
    out[i].y = fe_mul<B>(fe_mul<B>(pts[i].y, zi2), zi);
  }
}

// Pippenger on the device for n canonical scalars already in HBM. `lane` selects the stream + workspace set.
// msm_launch enqueues everything up to the device->host copy of the per-window sums; msm_finish waits and runs the window Horner.
struct MsmPending {
  int windows = 0;
  int lane = 0;
  int slot = 0;  // landing slot inside the lane's pinned buffer: several jobs may be in flight on one stream
  std::vector<jac_t> w;
};
static const int MSM_LANDING_SLOTS = 2;  // 8192-byte pinned buffer / (33 * 96 B) per job
static hipStream_t lane_stream(sp_ctx* c, int lane) { return lane ? c->stream2 : c->stream; }

int msm_launch(sp_ctx* c, const fe_t* d_canon_in, const aff_t* d_bases, size_t n, int windows, int lane, MsmPending* pend) {
  pend->windows = 0;
  pend->lane = lane;
  if (n == 0) return SP_OK;
  if (n >= (1u << 31)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "msm: n too large");
  hipStream_t st = lane_stream(c, lane);
  const fe_t* d_canon = d_canon_in;
  if (windows == spk::MSM_MAX_WINDOWS) {  // full-width scalars: fold signs so the top window is balanced
    fe_t* folded = (fe_t*)c->workspace(sp_ctx::WS_MSM_FOLDED, n * sizeof(fe_t), lane);
    if (!folded) return SP_ERR_NO_DEVICE;
    size_t fb = (n + 255) / 256;
    if (fb > 4096) fb = 4096;
    hipLaunchKernelGGL(spk::k_fold_sign, dim3((unsigned)fb), dim3(256), 0, st, d_canon_in, n, folded);
    d_canon = folded;
  }
  unsigned* order = (unsigned*)c->workspace(sp_ctx::WS_MSM_ORDER, (size_t)windows * n * 4, lane);
  unsigned* start = (unsigned*)c->workspace(sp_ctx::WS_MSM_START, (size_t)windows * (spk::MSM_BUCKETS + 1) * 4, lane);
  jac_t* buckets = (jac_t*)c->workspace(sp_ctx::WS_MSM_BUCKETS, (size_t)windows * spk::MSM_BUCKETS * sizeof(jac_t), lane);
  jac_t* wsum = (jac_t*)c->workspace(sp_ctx::WS_MSM_WSUM, (size_t)windows * sizeof(jac_t), lane);
  if (!order || !start || !buckets || !wsum) return SP_ERR_NO_DEVICE;
  auto run = [&](const char* what, uint64_t bytes, auto&& f) { c->timed_on(st, what, bytes, f); };
  signed char* digits = (signed char*)c->workspace(sp_ctx::WS_MSM_DIGITS, (size_t)windows * n, lane);
  if (!digits) return SP_ERR_NO_DEVICE;
  run("msm_sort", 32ull * n, [&] {
    hipLaunchKernelGGL(spk::k_msm_digits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_canon, (unsigned)n, windows, digits);
    hipLaunchKernelGGL(spk::k_msm_sort_digits, dim3(windows), dim3(256), 0, st, digits, d_canon, (unsigned)n, order, start);
  });
  unsigned lanes = (unsigned)windows * spk::MSM_BUCKETS * spk::MSM_LANES_PER_BUCKET;
  run("msm_bucket_sum", 96ull * n, [&] {
    hipLaunchKernelGGL(spk::k_msm_bucket_sum, dim3((lanes + 255) / 256), dim3(256), 0, st, d_bases, (unsigned)n, order, start, windows, buckets);
  });
  run("msm_window_reduce", 0, [&] {
    hipLaunchKernelGGL(spk::k_msm_window_reduce_coop, dim3(windows), dim3(4 * spk::MSM_BUCKETS), 0, st, buckets, wsum);
  });
  pend->windows = windows;
  pend->slot = (int)(c->msm_jobs_issued[lane]++ % MSM_LANDING_SLOTS);
  // pinned landing buffer: a device->host copy into pageable memory would block the host until the MSM is done
  SP_HIP(hipMemcpyAsync((char*)c->h_pinned_lane[lane] + pend->slot * 4096, wsum, windows * sizeof(jac_t), hipMemcpyDeviceToHost, st));
  // one event per in-flight job: finishing an early job must not wait for later work queued on the same stream
  hipEvent_t& ev = c->msm_ev[lane][pend->slot];
  if (!ev) SP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  SP_HIP(hipEventRecord(ev, st));
  return SP_OK;
}
int msm_finish(sp_ctx* c, MsmPending* pend, jac_t* result) {
  *result = jac_identity();
  if (pend->windows == 0) return SP_OK;
  SP_HIP(sp::event_sync(c->msm_ev[pend->lane][pend->slot]));
  pend->w.resize(pend->windows);
  memcpy(pend->w.data(), (char*)c->h_pinned_lane[pend->lane] + pend->slot * 4096, pend->windows * sizeof(jac_t));
  // Horner over windows, high to low (msm.rs:150-175): acc = 2^8 acc + W_w
  jac_t acc = jac_identity();
  for (int i = pend->windows - 1; i >= 0; --i) {
    for (int k = 0; k < spk::MSM_C; ++k) acc = jac_dbl(acc);
    acc = jac_add(acc, pend->w[i]);
  }
  *result = acc;
  return SP_OK;
}
int msm_device(sp_ctx* c, const fe_t* d_canon, const aff_t* d_bases, size_t n, int windows, jac_t* result) {
  MsmPending pend;
  int rc = msm_launch(c, d_canon, d_bases, n, windows, 0, &pend);
  if (rc) return rc;
  return msm_finish(c, &pend, result);
}

// Digit-path MSMs of many rows over one base vector (kernels_msm.hpp "batched row MSMs"): rows `sel` of the row-major canonical scalar
// array `canon` (cols per row, n scalars in all), `windows` = 33 (full scalars, signs folded) or 9 (values < 2^64). Results -> out[sel[i]].
int msm_rows_batched(sp_ctx* c, const fe_t* canon, size_t cols, size_t n, const std::vector<unsigned>& sel, int windows, const aff_t* d_bases,
                     std::vector<jac_t>& out) {
  if (sel.empty()) return SP_OK;
  const size_t CH = 512;  // rows per pass: 512 x 33 windows x (2048 x 4 B order + 128 x 96 B buckets) = ~350 MB of scratch
  const size_t ch = sel.size() < CH ? sel.size() : CH;
  DevBuf dsel, dscal, dorder, dstart, dbuckets, dwsum, drows;
  int rc;
  if ((rc = dsel.alloc(ch * 4)) || (rc = dscal.alloc(ch * cols * sizeof(fe_t))) || (rc = dorder.alloc(ch * windows * cols * 4)) ||
      (rc = dstart.alloc(ch * windows * (spk::MSM_BUCKETS + 1) * 4)) || (rc = dbuckets.alloc(ch * windows * spk::MSM_BUCKETS * sizeof(jac_t))) ||
      (rc = dwsum.alloc(sel.size() * windows * sizeof(jac_t))) || (rc = drows.alloc(sel.size() * sizeof(jac_t))))
    return rc;
  for (size_t base = 0; base < sel.size(); base += CH) {
    const size_t cnt = sel.size() - base < CH ? sel.size() - base : CH;
    SP_HIP(hipMemcpyAsync(dsel.p, sel.data() + base, cnt * 4, hipMemcpyHostToDevice, c->stream));
    if (windows == spk::MSM_MAX_WINDOWS) {
      hipLaunchKernelGGL(spk::k_fold_sign_rows, dim3((unsigned)((cols + 255) / 256), (unsigned)cnt), dim3(256), 0, c->stream, canon, dsel.as<unsigned>(), cols, n,
                         dscal.as<fe_t>());
    } else {  // narrow scalars: no folding, copy the rows into the dense [cnt][cols] layout the sort expects
      for (size_t i = 0; i < cnt; ++i) {
        const size_t lo = (size_t)sel[base + i] * cols, len = (lo + cols <= n) ? cols : n - lo;
        SP_HIP(hipMemcpyAsync(dscal.as<fe_t>() + i * cols, canon + lo, len * sizeof(fe_t), hipMemcpyDeviceToDevice, c->stream));
      }
    }
    c->timed("msm_rows_sort", 32ull * cnt * cols, [&] {
      hipLaunchKernelGGL(spk::k_msm_sort_rows, dim3(windows, (unsigned)cnt), dim3(256), 0, c->stream, dscal.as<fe_t>(), dsel.as<unsigned>(), cols, n,
                         dorder.as<unsigned>(), dstart.as<unsigned>());
    });
    const size_t total_buckets = cnt * windows * spk::MSM_BUCKETS;
    c->timed("msm_rows_bucket_sum", 96ull * cnt * cols, [&] {
      hipLaunchKernelGGL(spk::k_msm_bucket_sum_rows, dim3((unsigned)((total_buckets + 255) / 256)), dim3(256), 0, c->stream, d_bases, cols, dorder.as<unsigned>(),
                         dstart.as<unsigned>(), total_buckets, dbuckets.as<jac_t>());
    });
    const size_t nwin = cnt * windows;
    c->timed("msm_rows_window_reduce", 0, [&] {
      hipLaunchKernelGGL(spk::k_msm_window_reduce_seg, dim3((unsigned)((nwin * 8 + 255) / 256)), dim3(256), 0, c->stream, dbuckets.as<jac_t>(), nwin,
                         dwsum.as<jac_t>() + base * windows);
    });
    // the next pass reuses dsel / dscal / dorder: same stream, so it is ordered behind these kernels
  }
  // one Horner pass for all rows (one lane per row: 8 x 32 doublings + 33 additions, latency-bound, so it is worth doing once)
  c->timed("msm_rows_horner", 0, [&] {
    hipLaunchKernelGGL(spk::k_msm_horner_rows, dim3((unsigned)((sel.size() + 63) / 64)), dim3(64), 0, c->stream, dwsum.as<jac_t>(), windows, sel.size(),
                       drows.as<jac_t>());
  });
  std::vector<jac_t> res(sel.size());
  SP_HIP(hipMemcpyAsync(res.data(), drows.p, sel.size() * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  for (size_t i = 0; i < sel.size(); ++i) out[sel[i]] = res[i];
  return SP_OK;
}

int upload_canonical(sp_ctx* c, const uint64_t* scalars, size_t n, fe_t** canon_out, int lane = 0) {
  fe_t* raw = (fe_t*)c->workspace(sp_ctx::WS_SCALARS_RAW, n * sizeof(fe_t), lane);
  fe_t* canon = (fe_t*)c->workspace(sp_ctx::WS_SCALARS_CANON, n * sizeof(fe_t), lane);
  if (!raw || !canon) return SP_ERR_NO_DEVICE;
  hipStream_t st = lane ? c->stream2 : c->stream;
  if (n) {
    SP_HIP(hipMemcpyAsync(raw, scalars, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(spk::k_to_canonical, dim3((unsigned)blocks), dim3(256), 0, st, raw, n, canon);
  }
  SP_HIP(sp::stream_sync(st));  // `scalars` is a borrowed host buffer (pageable: the copy is staged synchronously anyway)
  *canon_out = canon;
  return SP_OK;
}

void store_aff(uint64_t* out, const aff_t& a) { memcpy(out, &a, 64); }
aff_t load_aff(const uint64_t* p) {
  aff_t a;
  memcpy(&a, p, 64);
  return a;
}

}  // namespace

extern "C" {

int sp_msm(sp_ctx* c, const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_aff[8]) {
  fe_t* canon;
  int rc;
  if ((rc = upload_canonical(c, scalars, n, &canon))) return rc;
  aff_t* dbases = (aff_t*)c->workspace(sp_ctx::WS_BASES_TMP, n * sizeof(aff_t));
  if (!dbases) return SP_ERR_NO_DEVICE;
  if (n) SP_HIP(hipMemcpyAsync(dbases, bases, n * sizeof(aff_t), hipMemcpyHostToDevice, c->stream));
  jac_t r;
  if (n >= sp::PIPPENGER_MIN) rc = sp::msm_pippenger(c, canon, dbases, n, true, 0, &r);
  else rc = msm_device(c, canon, dbases, n, spk::MSM_MAX_WINDOWS, &r);
  if (rc) return rc;
  store_aff(out_aff, jac_to_affine(r));
  return SP_OK;
}

int sp_weights_from_r(const uint64_t* r_bs, size_t ell, size_t n, uint64_t* out) {  // src/r1cs/mod.rs:153-166
  const fe_t one = fe_one<S>();
  for (size_t i = 0; i < n; ++i) {
    fe_t wi = one;
    size_t k = i;
    for (size_t t = 0; t < ell; ++t) {
      fe_t r;
      memcpy(&r, r_bs + 4 * t, 32);
      wi = fe_mul<S>(wi, (k & 1) ? r : fe_sub<S>(one, r));
      k >>= 1;
    }
    memcpy(out + 4 * i, &wi, 32);
  }
  return SP_OK;
}

int sp_fold_tables(sp_ctx* c, const sp_table* const* Ws, size_t n, const uint64_t* weights, size_t len, sp_table* out) {
  if (n == 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "fold_multiple: empty witness list");
  if (out->cap < len) return fail(SP_ERR_INVALID_INPUT_LENGTH, "fold_multiple: output too short");
  std::vector<const fe_t*> ptrs(n);
  for (size_t i = 0; i < n; ++i) {
    if (Ws[i]->cap < len) return fail(SP_ERR_INVALID_INPUT_LENGTH, "fold_multiple: all W vectors must have the same length");
    ptrs[i] = Ws[i]->d;
  }
  DevBuf dp, dw;
  int rc;
  if ((rc = dp.alloc(n * sizeof(fe_t*)))) return rc;
  if ((rc = dw.alloc(n * sizeof(fe_t)))) return rc;
  SP_HIP(hipMemcpyAsync(dp.p, ptrs.data(), n * sizeof(fe_t*), hipMemcpyHostToDevice, c->stream));
  SP_HIP(hipMemcpyAsync(dw.p, weights, n * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  size_t blocks = (len + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  c->timed("fold_tables", 32ull * (n + 1) * len, [&] {
    hipLaunchKernelGGL(spk::k_fold_tables, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const fe_t* const*)dp.p, dw.as<fe_t>(), n, len, out->d);
  });
  SP_HIP(sp::stream_sync(c->stream));
  out->len = len;
  out->lo_eff = out->hi_eff = (size_t)-1;
  return SP_OK;
}

// lane 0: the context's main stream and workspaces (the owner's thread); lane 1: the auxiliary stream and its workspaces - callable from a helper thread
// beside the owner's calls on the main stream (sp_msm_shared_weights_aux: the commitment fold of a NIFS beside its witness and layer folds)
static int msm_shared_weights_on(sp_ctx* c, int lane, const uint64_t* weights, size_t n, const uint64_t* bases_rows, size_t rows, uint64_t* out_rows_aff) {
  hipStream_t const st = lane ? c->stream2 : c->stream;
  if (rows == 0) return SP_OK;
  if (n == 0) {
    memset(out_rows_aff, 0, rows * sizeof(aff_t));
    return SP_OK;
  }
  const int windows = spk::MSM_MAX_WINDOWS;
  static const bool laps = [] {
    const char* e = getenv("SPARTAN_HOST_LAPS");
    return e && e[0] == '1';
  }();
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_lap = now();
  auto lap = [&](const char* what) {
    if (!laps) return;
    (void)sp::stream_sync(st);
    const double t = now();
    fprintf(stderr, "msm_shared_weights lap %-20s %8.3f ms\n", what, t - t_lap);
    t_lap = t;
  };
  fe_t* canon;
  int rc;
  if ((rc = upload_canonical(c, weights, n, &canon, lane))) return rc;
  // grow-only context workspaces (no hipMalloc / hipFree on the path)
  aff_t* dbases = (aff_t*)c->workspace(sp_ctx::WS_BASES_TMP, rows * n * sizeof(aff_t), lane);
  fe_t* folded = (fe_t*)c->workspace(sp_ctx::WS_MSM_FOLDED, n * sizeof(fe_t), lane);
  unsigned* order = (unsigned*)c->workspace(sp_ctx::WS_MSM_ORDER, (size_t)windows * n * 4, lane);
  unsigned* start = (unsigned*)c->workspace(sp_ctx::WS_MSM_START, (size_t)windows * (spk::MSM_BUCKETS + 1) * 4, lane);
  jac_t* buckets = (jac_t*)c->workspace(sp_ctx::WS_MSM_BUCKETS, rows * windows * spk::MSM_BUCKETS * sizeof(jac_t), lane);
  jac_t* wsum = (jac_t*)c->workspace(sp_ctx::WS_MSM_WSUM, rows * windows * sizeof(jac_t), lane);
  jac_t* drows = (jac_t*)c->workspace(sp_ctx::WS_COMMIT_ROWS, rows * sizeof(jac_t), lane);
  if (!dbases || !folded || !order || !start || !buckets || !wsum || !drows) return SP_ERR_NO_DEVICE;
  SP_HIP(hipMemcpyAsync(dbases, bases_rows, rows * n * sizeof(aff_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(spk::k_fold_sign, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, canon, n, folded);
  // the digit decomposition / bucket order is shared by every row (msm.rs:266-300); buckets and window sums are per row
  lap("upload");
  hipLaunchKernelGGL(spk::k_msm_sort, dim3(windows), dim3(256), 0, st, folded, (unsigned)n, order, start);
  if (rows >= 64) {  // throughput regime: one lane per (row, window, bucket), work-efficient window sums
    const size_t total = rows * (size_t)windows * spk::MSM_BUCKETS, nwin = rows * (size_t)windows;
    c->timed_on(st, "msm_shared_bucket_sum", 64ull * n * rows, [&] {
      hipLaunchKernelGGL(spk::k_msm_bucket_sum_shared, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dbases, (unsigned)n, order, start, windows, total, buckets);
    });
    lap("sort + bucket sums");
    hipLaunchKernelGGL(spk::k_msm_window_reduce_seg, dim3((unsigned)((nwin * 8 + 255) / 256)), dim3(256), 0, st, buckets, nwin, wsum);
  } else {
    unsigned lanes = (unsigned)windows * spk::MSM_BUCKETS * spk::MSM_LANES_PER_BUCKET;
    c->timed_on(st, "msm_shared_bucket_sum", 64ull * n * rows, [&] {
      hipLaunchKernelGGL(spk::k_msm_bucket_sum, dim3((lanes + 255) / 256, (unsigned)rows), dim3(256), 0, st, dbases, (unsigned)n, order, start, windows, buckets);
    });
    lap("sort + bucket sums");
    if (rows * (size_t)windows <= 1024)  // few (row, window) pairs: the block-cooperative form (chain latency is all there is)
      hipLaunchKernelGGL(spk::k_msm_window_reduce_coop, dim3(windows, (unsigned)rows), dim3(4 * spk::MSM_BUCKETS), 0, st, buckets, wsum);
    else
      hipLaunchKernelGGL(spk::k_msm_window_reduce, dim3(windows, (unsigned)rows), dim3(spk::MSM_BUCKETS), 0, st, buckets, wsum);
  }
  lap("window sums");
  std::vector<jac_t> res(rows);
  if (rows <= 64) {
    // few rows: the 256-doubling window Horner is a latency chain (~3 ms for one lane per row on the device, ~60 us per row on the host)
    std::vector<jac_t> ws(rows * windows);
    SP_HIP(hipMemcpyAsync(ws.data(), wsum, ws.size() * sizeof(jac_t), hipMemcpyDeviceToHost, st));
    SP_HIP(sp::stream_sync(st));
    auto horner = [&](size_t r) {
      jac_t acc = jac_identity();
      for (int w = windows - 1; w >= 0; --w) {
        for (int k = 0; k < spk::MSM_C; ++k) acc = jac_dbl(acc);
        acc = jac_add(acc, ws[r * windows + w]);
      }
      res[r] = acc;
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 8) nt = 8;
    if (nt > rows) nt = (unsigned)rows;
    sp::WalkPool& pool = sp::WalkPool::get();
    if (rows >= 4 && pool.walkers() > 0) {
      // on the process's polling threads (no thread is created inside a prove: eight clones and eight 8 MiB stack mappings a call take the address
      // space's lock against every page fault of the process)
      struct H {
        decltype(horner)* f;
        size_t rows;
      } h{&horner, rows};
      pool.keep_hot(2000);
      pool.run((unsigned)std::min<size_t>(rows, (size_t)pool.walkers() + 1), [](void* a, unsigned p, unsigned np) {
        H& x = *static_cast<H*>(a);
        for (size_t r = x.rows * p / np; r < x.rows * (p + 1) / np; ++r) (*x.f)(r);
      }, &h);
    } else if (rows < 4 || nt < 2) {
      for (size_t r = 0; r < rows; ++r) horner(r);
    } else {  // ~60 us per row: the 16-row commitment fold of a NeutronNova batch would spend a millisecond here on one core
      std::vector<std::thread> th;
      for (unsigned k = 0; k < nt; ++k)
        th.emplace_back([&, k] {
          for (size_t r = k; r < rows; r += nt) horner(r);
        });
      for (auto& t : th) t.join();
    }
  } else {
    hipLaunchKernelGGL(spk::k_msm_horner_rows, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, wsum, windows, rows, drows);
    SP_HIP(hipMemcpyAsync(res.data(), drows, rows * sizeof(jac_t), hipMemcpyDeviceToHost, st));
    SP_HIP(sp::stream_sync(st));
  }
  lap("horner");
  std::vector<aff_t> a(rows);
  normalize_batch(res, a.data());
  memcpy(out_rows_aff, a.data(), rows * sizeof(aff_t));
  lap("normalize");
  return SP_OK;
}

int sp_msm_shared_weights(sp_ctx* c, const uint64_t* weights, size_t n, const uint64_t* bases_rows, size_t rows, uint64_t* out_rows_aff) {
  return msm_shared_weights_on(c, 0, weights, n, bases_rows, rows, out_rows_aff);
}
int sp_msm_shared_weights_aux(sp_ctx* c, const uint64_t* weights, size_t n, const uint64_t* bases_rows, size_t rows, uint64_t* out_rows_aff) {
  return msm_shared_weights_on(c, 1, weights, n, bases_rows, rows, out_rows_aff);
}

int sp_point_sum(const uint64_t* points_aff, size_t n, uint64_t out_aff[8]) {
  jac_t acc = jac_identity();
  for (size_t i = 0; i < n; ++i) {
    aff_t p = load_aff(points_aff + 8 * i);
    if (!aff_on_curve(p)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_point_sum: point not on the curve");
    acc = jac_add_mixed(acc, p);
  }
  store_aff(out_aff, jac_to_affine(acc));
  return SP_OK;
}

int sp_msm_small_u64(sp_ctx* c, const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_aff[8]) {
  // u64 scalars are canonical 256-bit values with zero upper limbs: 8 byte windows + the carry window
  std::vector<fe_t> canon_h(n);
  for (size_t i = 0; i < n; ++i) {
    canon_h[i] = fe_zero();
    canon_h[i].v[0] = (uint32_t)scalars[i];
    canon_h[i].v[1] = (uint32_t)(scalars[i] >> 32);
  }
  DevBuf canon, dbases;
  int rc;
  if ((rc = canon.alloc(n * sizeof(fe_t)))) return rc;
  if ((rc = dbases.alloc(n * sizeof(aff_t)))) return rc;
  if (n) {
    SP_HIP(hipMemcpyAsync(canon.p, canon_h.data(), n * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
    SP_HIP(hipMemcpyAsync(dbases.p, bases, n * sizeof(aff_t), hipMemcpyHostToDevice, c->stream));
  }
  jac_t r;
  if (n >= sp::PIPPENGER_MIN) rc = sp::msm_pippenger(c, canon.as<fe_t>(), dbases.as<aff_t>(), n, false, 0, &r);
  else rc = msm_device(c, canon.as<fe_t>(), dbases.as<aff_t>(), n, 9, &r);
  if (rc) return rc;
  store_aff(out_aff, jac_to_affine(r));
  return SP_OK;
}

static const size_t HOST16_BASES = 16;  // leading bases of a narrow key whose 16-bit-window tables are kept on the host too (64 MiB each)
static bool fbtables_old_build() {
  static const bool on = [] {
    const char* e = getenv("SPARTAN_FBTABLES_OLD");  // "1": the round-5 build (k_fixed_base_tables + one inversion per entry) for A/B runs and the parity test
    return e && e[0] == '1';
  }();
  return on;
}
static bool fb_mapped_enabled();
static int fb_mapped_ensure(sp_ctx* c, int lane);
// 16-bit window tables of the latency paths (kernels_msm.hpp k_fixed_base_tables16), found by the address of the 8-bit table set they shadow.
static std::mutex g_t16_mu;
static std::map<const aff_t*, const aff_t*> g_t16;
static bool fb_window16_enabled() { return true; }
static const aff_t* tables16_of(const aff_t* t8) {
  std::lock_guard<std::mutex> l(g_t16_mu);
  auto it = g_t16.find(t8);
  return it == g_t16.end() ? nullptr : it->second;
}
// The host copy of a key's 16-bit-window tables (64 MiB a table): one per distinct base set and process - the contexts of a multi-context run create the same
// keys, and ck / ck_s are the same labels in every one - held weakly here and strongly by the keys, so that the last key to go frees it. Filled by one
// device -> host copy into memory that is not value-initialised first (1.5 GB of zero-filled vectors in the eight-context mode).
static int host_tables16_of(const std::vector<aff_t>& bases, const aff_t* d_t16, const std::vector<size_t>& which, std::shared_ptr<aff_t[]>* out) {
  static const bool on = [] {
    const char* e = getenv("SPARTAN_HOST_T16");
    return !(e && e[0] == '0');
  }();
  out->reset();
  if (!on || which.empty()) return SP_OK;
  static std::mutex mu;
  static std::map<std::string, std::weak_ptr<aff_t[]>> cache;
  std::string key;
  for (size_t t : which) key.append(reinterpret_cast<const char*>(&bases[t]), sizeof(aff_t));
  std::lock_guard<std::mutex> l(mu);
  auto it = cache.find(key);
  if (it != cache.end()) {
    if (auto sp = it->second.lock()) {
      *out = sp;
      return SP_OK;
    }
    cache.erase(it);
  }
  const size_t per16 = (size_t)16 * 65535, bytes = which.size() * per16 * sizeof(aff_t);
  // 2 MiB-aligned and advised for huge pages: a walk touches 16 random lines of a 64 MiB table (4 KiB pages: a TLB miss each)
  void* raw = aligned_alloc((size_t)2 << 20, (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1));
  if (!raw) return SP_OK;  // (no host copy: the 8-bit tables / the device form serve)
  {
    static const bool thp = [] {
      const char* e = getenv("SPARTAN_HOST_T16_THP");  // "0": ordinary pages
      return !(e && e[0] == '0');
    }();
    if (thp) madvise(raw, bytes, MADV_HUGEPAGE);
  }
  std::shared_ptr<aff_t[]> buf(static_cast<aff_t*>(raw), [](aff_t* p) { free(p); });
  for (size_t k = 0; k < which.size(); ++k) {
    size_t run = 1;  // consecutive tables in one copy
    while (k + run < which.size() && which[k + run] == which[k] + run) ++run;
    SP_HIP(hipMemcpy(buf.get() + k * per16, d_t16 + which[k] * per16, run * per16 * sizeof(aff_t), hipMemcpyDeviceToHost));
    k += run - 1;
  }
  cache[key] = buf;
  *out = buf;
  return SP_OK;
}
int sp_ck_create(sp_ctx* c, const uint64_t* ck_aff, size_t num_cols, const uint64_t h_aff[8], sp_ck** out) {
  if (num_cols == 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_ck_create: empty key");
  sp_ck* k = new sp_ck();
  k->ctx = c;
  k->num_cols = num_cols;
  k->h = load_aff(h_aff);
  SP_HIP(hipMalloc((void**)&k->d_bases, num_cols * sizeof(aff_t)));
  SP_HIP(hipMemcpyAsync(k->d_bases, ck_aff, num_cols * sizeof(aff_t), hipMemcpyHostToDevice, c->stream));
  // FixedBaseMul::precompute(., 8) (msm.rs:653-689): always for h; for every base too when the key is narrow (<= 64)
  const size_t ntab = (num_cols <= 64) ? num_cols + 1 : 1;
  const size_t per = 32 * 255;
  int rc;
  std::vector<aff_t> hb(ntab);  // the tables' bases: the key's (narrow keys only), then h
  for (size_t t = 0; t < ntab; ++t) hb[t] = (t + 1 == ntab) ? k->h : load_aff(ck_aff + 8 * t);
  DevBuf dpts;
  if ((rc = dpts.alloc(ntab * sizeof(aff_t)))) return rc;
  SP_HIP(hipMemcpyAsync(dpts.p, hb.data(), ntab * sizeof(aff_t), hipMemcpyHostToDevice, c->stream));
  aff_t* tables;
  SP_HIP(hipMalloc((void**)&tables, ntab * per * sizeof(aff_t)));
  if (fbtables_old_build()) {
    DevBuf tj;
    if ((rc = tj.alloc(ntab * per * sizeof(jac_t)))) return rc;
    for (size_t t = 0; t < ntab; ++t) sp::launch_fixed_base_table(c->stream, hb[t], tj.as<jac_t>() + t * per);
    sp::launch_jac_to_affine(c->stream, tj.as<jac_t>(), ntab * per, tables);
    SP_HIP(sp::stream_sync(c->stream));
  } else {  // the three-launch build for all bases at once (capi_bulk.hip launch_window_tables)
    DevBuf scratch;
    if ((rc = scratch.alloc(sp::window_tables_scratch(ntab < sp::WT_CHUNK ? ntab : sp::WT_CHUNK)))) return rc;
    sp::launch_window_tables(c->stream, dpts.as<aff_t>(), ntab, (char*)scratch.p, tables);
    SP_HIP(sp::stream_sync(c->stream));
  }
  if (fb_mapped_enabled() && ((rc = fb_mapped_ensure(c, 0)) || (rc = fb_mapped_ensure(c, 1)))) return rc;
  k->n_tables = ntab;
  k->h_tables.resize(ntab * per);
  SP_HIP(hipMemcpy(k->h_tables.data(), tables, ntab * per * sizeof(aff_t), hipMemcpyDeviceToHost));
  if (ntab > 1) {
    k->d_cktables = tables;
    k->d_htable = tables + (ntab - 1) * per;
  } else {
    k->d_htable = tables;
  }
  if (fb_mapped_enabled() && fb_window16_enabled()) {
    // the same bases with 16-bit windows (4 MiB per window, 64 MiB per base) for calls of <= 128 scalars: one tree level less per call
    const size_t per16 = (size_t)16 * 65535;
    DevBuf tj16, pre16;
    if ((rc = tj16.alloc(ntab * per16 * sizeof(jac_t))) || (rc = pre16.alloc(ntab * per16 * sizeof(fe_t)))) return rc;
    aff_t* t16 = nullptr;
    SP_HIP(hipMalloc((void**)&t16, ntab * per16 * sizeof(aff_t)));
    sp::launch_fixed_base_tables16(c->stream, dpts.as<aff_t>(), ntab, tj16.as<jac_t>());
    sp::launch_jac_to_affine(c->stream, tj16.as<jac_t>(), ntab * per16, t16, fbtables_old_build() ? nullptr : pre16.as<fe_t>());
    SP_HIP(sp::stream_sync(c->stream));
    k->d_tables16 = t16;
    {
      // host copy: every table of a key of <= 2 (the single multiplications: commitments of one value - eval_W, beta, a blind's term); of a narrow key
      // the leading bases and h when the process keeps table walkers (walk_pool.hpp: the round commitments of the ZK verifier circuit use <= 15 columns)
      std::vector<size_t> which;
      k->h16_bases = ntab <= 2 ? ntab - 1 : (sp::WalkPool::get().walkers() > 0 ? std::min(ntab - 1, HOST16_BASES) : (size_t)0);
      if (ntab <= 2 || k->h16_bases) {
        for (size_t t = 0; t < k->h16_bases; ++t) which.push_back(t);
        which.push_back(ntab - 1);
      }
      if ((rc = host_tables16_of(hb, t16, which, &k->h_tables16))) return rc;
    }
    std::lock_guard<std::mutex> l(g_t16_mu);
    if (ntab > 1) g_t16[k->d_cktables] = t16;
    g_t16[k->d_htable] = t16 + (ntab - 1) * per16;
  }
  *out = k;
  return SP_OK;
}
void sp_ck_free(sp_ck* k) {
  if (!k) return;
  if (k->h_tables16) sp::WalkPool::get().quiesce();  // (a walker that lost its core mid-walk may still be reading the host tables)
  if (k->d_tables16) {
    {
      std::lock_guard<std::mutex> l(g_t16_mu);
      if (k->d_cktables) g_t16.erase(k->d_cktables);
      g_t16.erase(k->d_htable);
    }
    hipFree(k->d_tables16);
  }
  if (k->d_bases) hipFree(k->d_bases);
  if (k->d_comb) hipFree(k->d_comb);
  if (k->d_keytables) hipFree(k->d_keytables);
  if (k->d_cktables) hipFree(k->d_cktables);
  else if (k->d_htable) hipFree(k->d_htable);
  delete k;
}

// FixedBaseMul::mul on the host (msm.rs:691-725): <= 32 mixed additions
static jac_t fixed_base_mul_host(const aff_t* table, const fe_t& scalar) {
  const fe_t c = fe_to_canonical<S>(scalar);
  jac_t acc = jac_identity();
  for (int j = 0; j < 32; ++j) {
    unsigned digit = (c.v[j >> 2] >> (8 * (j & 3))) & 0xffu;
    if (digit) acc = jac_add_mixed(acc, table[(size_t)j * 255 + digit - 1]);
  }
  return acc;
}
// The same over the 16-bit-window tables: 16 mixed additions; the sixteen entries (each a miss in a 64 MiB table) are requested before the first is used.
static jac_t fixed_base_mul_host16(const aff_t* table16, const fe_t& scalar) {
  const fe_t c = fe_to_canonical<S>(scalar);
  const aff_t* ent[16];
  for (int j = 0; j < 16; ++j) {
    const unsigned digit = (c.v[j >> 1] >> (16 * (j & 1))) & 0xffffu;
    ent[j] = digit ? table16 + (size_t)j * 65535 + digit - 1 : nullptr;
    if (ent[j]) __builtin_prefetch(ent[j], 0, 0);
  }
  jac_t acc = jac_identity();
  for (int j = 0; j < 16; ++j)
    if (ent[j]) acc = jac_add_mixed(acc, *ent[j]);
  return acc;
}
// table t of a key (t = n_tables - 1: h) times a scalar, on the host: over the 16-bit windows when the key keeps them here
static jac_t ck_mul_host(const sp_ck* ck, size_t t, const fe_t& scalar) {
  const aff_t* t16 = ck->host_table16(t);
  return t16 ? fixed_base_mul_host16(t16, scalar) : fixed_base_mul_host(ck->host_table(t), scalar);
}
// few scalars (the latency case: the blinds of the zero rows): block-cooperative additions; many: one half-wave per scalar (throughput).
static void launch_fixed_base_rows(hipStream_t st, const fe_t* ds, size_t n, const aff_t* tables, size_t ntables, jac_t* dout) {
  if (n <= 2048) {  // 512 blocks of four scalars: one resident wave of blocks at two per CU
    hipLaunchKernelGGL(spk::k_fixed_base_rows_coop, dim3((unsigned)((n + 3) / 4)), dim3(512), 0, st, ds, n, tables, ntables, dout);
  } else {
    const size_t threads = n * 32;
    hipLaunchKernelGGL(spk::k_fixed_base_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, ds, n, tables, ntables, dout);
  }
}
static const size_t VEC_CTRL_BYTES = 64;    // control line of an armed scaled sum (k_scale_add_wait), behind the flags
static const size_t VEC_FLAG_BYTES = 4096;  // per-block arrival flags of sp_rowmat_vec_eq_finish_scaled (1024 blocks of 256 columns)
static const size_t FIXED_BASE_HOST_MAX = 8;  // below this many scalars one CPU core beats the launch + single-wave latency

// table[i % ntables] * scalars[i] on the device (Jacobian results in host memory); ntables == 1 for h
// <= 128 scalars through mapped memory (kernels_msm.hpp k_fixed_base_rows_coop_mapped): launch on lane 0 = the main stream / lane 1 = the auxiliary stream,
// then poll the n self-validating result slots. Larger calls keep the copy / launch / copy / synchronise form.
static const size_t FB_MAPPED_MAX = 128, FB_SLOT_BYTES = 4 * spk::FB_SLOT_WORDS;
static const size_t FB_MAPPED_CAP = 640;  // slots the mapped pages hold: the row commitments of host vectors on a narrow key take up to 16 rows x 33 (sp_hyrax_commit_rows_host)
static bool fb_mapped_enabled() { return true; }
// (allocated when a key is created, not inside a prove: an allocation call can wait for the device, and with it for resident kernels of other contexts
// that are themselves waiting for host threads the call may be holding up)
static int fb_mapped_ensure(sp_ctx* c, int lane) {
  const size_t bytes = FB_MAPPED_CAP * FB_SLOT_BYTES + FB_MAPPED_CAP * sizeof(fe_t);
  if (!c->h_fbm[lane]) {
    SP_HIP(hipHostMalloc(&c->h_fbm[lane], bytes, hipHostMallocMapped));
    memset(c->h_fbm[lane], 0, bytes);
    SP_HIP(hipHostGetDevicePointer(&c->d_fbm[lane], c->h_fbm[lane], 0));
  }
  return SP_OK;
}
static int fb_mapped_launch(sp_ctx* c, int lane, const aff_t* d_tables, size_t ntables, const uint64_t* scalars, size_t n, bool xyzz_out = false) {
  int erc = fb_mapped_ensure(c, lane);
  if (erc) return erc;
  memcpy((char*)c->h_fbm[lane] + FB_MAPPED_CAP * FB_SLOT_BYTES, scalars, n * sizeof(fe_t));
  if (++c->fbm_seq[lane] == 0) ++c->fbm_seq[lane];
  hipStream_t st = lane ? c->stream2 : c->stream;
  const fe_t* ds = reinterpret_cast<const fe_t*>((char*)c->d_fbm[lane] + FB_MAPPED_CAP * FB_SLOT_BYTES);
  unsigned* dslots = reinterpret_cast<unsigned*>(c->d_fbm[lane]);
  const unsigned seq = c->fbm_seq[lane];
  const aff_t* t16 = fb_window16_enabled() ? tables16_of(d_tables) : nullptr;
  c->timed_on(st, "fixed_base", 32ull * n, [&] {
    if (t16 && xyzz_out) hipLaunchKernelGGL((spk::k_fixed_base_rows_coop_mapped<64, 16, true>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ds, n, t16, ntables, dslots, seq);
    else if (t16) hipLaunchKernelGGL((spk::k_fixed_base_rows_coop_mapped<64, 16, false>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ds, n, t16, ntables, dslots, seq);
    else if (xyzz_out) hipLaunchKernelGGL((spk::k_fixed_base_rows_coop_mapped<64, 8, true>), dim3((unsigned)((n + 1) / 2)), dim3(256), 0, st, ds, n, d_tables, ntables, dslots, seq);
    else hipLaunchKernelGGL((spk::k_fixed_base_rows_coop_mapped<64, 8, false>), dim3((unsigned)((n + 1) / 2)), dim3(256), 0, st, ds, n, d_tables, ntables, dslots, seq);
  });
  return SP_OK;
}
// `seq` = the sequence number fb_mapped_launch gave the job being collected (c->fbm_seq[lane] right after the launch)
static int fb_mapped_collect(sp_ctx* c, int lane, size_t n, void* out_, unsigned seq, int D = 24) {  // D = 24: jac_t results, 32: xyzz_t
  unsigned* out = reinterpret_cast<unsigned*>(out_);
  hipStream_t st = lane ? c->stream2 : c->stream;
  bool synced = false;
  const int T = spk::FB_SLOT_TAG;
  for (size_t i = 0; i < n; ++i) {
    volatile const unsigned* slot = reinterpret_cast<volatile const unsigned*>((char*)c->h_fbm[lane] + FB_SLOT_BYTES * i);
    unsigned w[32];
    for (long spins = 0;; ++spins) {
      if (slot[T] == seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        unsigned a = seq, b = seq * 0x9E3779B1u;
        for (int k = 0; k < D; ++k) {
          w[k] = slot[k];
          a += w[k];
          b += (unsigned)(k + 1) * w[k];
        }
        if (slot[T] == seq && slot[T + 1] == a && slot[T + 2] == b && slot[T + 3] == seq) break;
      }
      if (spins > 4000000) {
        sp::slow_note("fb_mapped_collect", spins);
        if (synced) return fail(SP_ERR_INTERNAL, "fixed-base rows: the kernel did not deliver a result slot");
        SP_HIP(sp::stream_sync(st));  // e.g. under a profiler
        synced = true;
        spins = 0;
      }
      sp::relax();
    }
    memcpy(out + (size_t)D * i, w, 4 * (size_t)D);
  }
  return SP_OK;
}

static int fixed_base_rows(sp_ctx* c, const aff_t* d_tables, size_t ntables, const uint64_t* scalars, size_t n, std::vector<jac_t>& out) {
  out.assign(n, jac_identity());
  if (n == 0) return SP_OK;
  fe_t* ds = (fe_t*)c->workspace(sp_ctx::WS_FB_SCALARS, n * sizeof(fe_t));
  jac_t* dout = (jac_t*)c->workspace(sp_ctx::WS_FB_OUT, n * sizeof(jac_t));
  if (!ds || !dout) return SP_ERR_NO_DEVICE;
  if (n <= FB_MAPPED_MAX && fb_mapped_enabled()) {  // the latency case (one call per round of the ZK verifier circuit): no copies, no synchronise
    int rc = fb_mapped_launch(c, 0, d_tables, ntables, scalars, n);
    return rc ? rc : fb_mapped_collect(c, 0, n, out.data(), c->fbm_seq[0]);
  }
  if (n <= 1024) {
    // both copies through pinned memory, so neither stages through a bounce buffer
    if (!c->h_pinned_fbs) SP_HIP(hipHostMalloc(&c->h_pinned_fbs, 1024 * (sizeof(jac_t) + sizeof(fe_t))));
    jac_t* hp = (jac_t*)c->h_pinned_fbs;
    fe_t* hs = (fe_t*)((char*)c->h_pinned_fbs + 1024 * sizeof(jac_t));
    memcpy(hs, scalars, n * sizeof(fe_t));
    SP_HIP(hipMemcpyAsync(ds, hs, n * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
    c->timed("fixed_base", 32ull * n, [&] { launch_fixed_base_rows(c->stream, ds, n, d_tables, ntables, dout); });
    SP_HIP(hipMemcpyAsync(hp, dout, n * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
    SP_HIP(sp::stream_sync(c->stream));
    memcpy(out.data(), hp, n * sizeof(jac_t));
    return SP_OK;
  }
  SP_HIP(hipMemcpyAsync(ds, scalars, n * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  c->timed("fixed_base", 32ull * n, [&] {
    launch_fixed_base_rows(c->stream, ds, n, d_tables, ntables, dout);
  });
  SP_HIP(hipMemcpyAsync(out.data(), dout, n * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  return SP_OK;
}

struct sp_fb_job {
  size_t n = 0;
  std::vector<jac_t> host_pts;  // filled directly for small n
  bool on_device = false, mapped = false;
  unsigned seq = 0;  // mapped: the sequence number of THIS job's result slots
  jac_t* pinned = nullptr;
};
int sp_fixed_base_mul_h_begin(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, sp_fb_job** out) {
  // the asynchronous forms share one landing area per context (the mapped scalar / slot page of lane 1, the pinned buffer + event of the copy form):
  // a second job before the first one's finish would overwrite its scalars and hand the first finish the second job's points
  if (n > FIXED_BASE_HOST_MAX && c->fb_async_busy)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "fixed_base_mul_h_begin: an asynchronous job is already outstanding on this context (finish it first)");
  sp_fb_job* job = new sp_fb_job();
  job->n = n;
  if (n <= FIXED_BASE_HOST_MAX) {
    job->host_pts.resize(n);
    for (size_t i = 0; i < n; ++i) {
      fe_t sc;
      memcpy(&sc, scalars + 4 * i, 32);
      job->host_pts[i] = ck_mul_host(ck, ck->n_tables - 1, sc);
    }
  } else if (n <= FB_MAPPED_MAX && fb_mapped_enabled()) {
    int rc = fb_mapped_launch(c, 1, ck->d_htable, 1, scalars, n);
    if (rc) {
      delete job;
      return rc;
    }
    job->mapped = true;
    job->seq = c->fbm_seq[1];
    c->fb_async_busy = true;
  } else {
    if (n * sizeof(jac_t) > 4096 * sizeof(jac_t)) {
      delete job;
      return fail(SP_ERR_INVALID_INPUT_LENGTH, "fixed_base_mul_h_begin: at most 4096 scalars per asynchronous job");
    }
    if (!c->h_pinned_fb) SP_HIP(hipHostMalloc(&c->h_pinned_fb, 4096 * (sizeof(jac_t) + sizeof(fe_t))));
    fe_t* stage = (fe_t*)((char*)c->h_pinned_fb + 4096 * sizeof(jac_t));
    memcpy(stage, scalars, n * sizeof(fe_t));  // pinned staging: the caller's buffer is released on return
    // auxiliary stream: the job overlaps whatever the caller does next on the main stream
    fe_t* ds = (fe_t*)c->workspace(sp_ctx::WS_FB_SCALARS, n * sizeof(fe_t), 1);
    jac_t* dout = (jac_t*)c->workspace(sp_ctx::WS_FB_OUT, n * sizeof(jac_t), 1);
    if (!ds || !dout) {
      delete job;
      return SP_ERR_NO_DEVICE;
    }
    SP_HIP(hipMemcpyAsync(ds, stage, n * sizeof(fe_t), hipMemcpyHostToDevice, c->stream2));
    c->timed_on(c->stream2, "fixed_base", 32ull * n, [&] { launch_fixed_base_rows(c->stream2, ds, n, ck->d_htable, (size_t)1, dout); });
    SP_HIP(hipMemcpyAsync(c->h_pinned_fb, dout, n * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream2));
    SP_HIP(hipEventRecord(c->fb_event(), c->stream2));
    job->on_device = true;
    job->pinned = (jac_t*)c->h_pinned_fb;
    c->fb_async_busy = true;
  }
  *out = job;
  return SP_OK;
}
int sp_fixed_base_mul_h_finish(sp_ctx* c, sp_fb_job* job, uint64_t* out_aff) {
  std::vector<jac_t> pts;
  if (job->mapped || job->on_device) c->fb_async_busy = false;
  if (job->mapped) {
    pts.resize(job->n);
    int rc = fb_mapped_collect(c, 1, job->n, pts.data(), job->seq);
    if (rc) {
      delete job;
      return rc;
    }
  } else if (job->on_device) {
    SP_HIP(sp::event_sync(c->fb_event()));
    pts.assign(job->pinned, job->pinned + job->n);
  } else {
    pts.swap(job->host_pts);
  }
  const size_t n = job->n;
  delete job;
  std::vector<aff_t> a(n);
  normalize_batch(pts, a.data());
  if (n) memcpy(out_aff, a.data(), n * sizeof(aff_t));
  return SP_OK;
}

int sp_fixed_base_mul_h(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, uint64_t* out_aff) {
  std::vector<jac_t> pts;
  if (n <= FIXED_BASE_HOST_MAX) {
    pts.resize(n);
    for (size_t i = 0; i < n; ++i) {
      fe_t sc;
      memcpy(&sc, scalars + 4 * i, 32);
      pts[i] = ck_mul_host(ck, ck->n_tables - 1, sc);
    }
  } else {
    int rc = fixed_base_rows(c, ck->d_htable, 1, scalars, n, pts);
    if (rc) return rc;
  }
  std::vector<aff_t> a(n);
  normalize_batch(pts, a.data());
  if (n) memcpy(out_aff, a.data(), n * sizeof(aff_t));
  return SP_OK;
}

// The rows of PCS::commit as Jacobian sums: per-row MSM (+ h * blind[row] when blinds are given; without them the raw MSMs of commit_without_blind,
// hyrax_pc.rs:533-568, an all-zero row being the identity).
static int commit_rows(sp_ctx* c, const sp_ck* ck, const sp_table* v, size_t off, size_t n, const uint64_t* blinds, std::vector<jac_t>& out_rows) {
  if (off + n > v->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "hyrax commit: range exceeds the table");
  const size_t cols = ck->num_cols, rows = (n + cols - 1) / cols;
  out_rows.clear();
  if (rows == 0) return SP_OK;
  int rc;
  if (ck->d_cktables) {
    // keys of <= 64 bases: per-base FixedBaseMul tables, as the reference does (hyrax_pc.rs:221-260 -> multi_mul, msm.rs:727-773): every
    // (row, column) scalar and every row blind walks its own table in ONE launch; the cols + 1 points of a row are added on the host side
    const size_t per = cols + 1, total = rows * per;
    // grow-only context workspaces (no hipMalloc / hipFree on the path); the `per` points of a row are added by one wave per row on the device
    fe_t* ds = (fe_t*)c->workspace(sp_ctx::WS_NARROW_SCALARS, total * sizeof(fe_t));
    jac_t* dout = (jac_t*)c->workspace(sp_ctx::WS_NARROW_OUT, (total + rows) * sizeof(jac_t));
    fe_t* dbl = (fe_t*)c->workspace(sp_ctx::WS_NARROW_BLINDS, rows * sizeof(fe_t));
    if (!ds || !dout || !dbl) return SP_ERR_NO_DEVICE;
    jac_t* drow = dout + total;
    SP_HIP(hipMemsetAsync(ds, 0, total * sizeof(fe_t), c->stream));
    const size_t full_rows = n / cols;
    if (full_rows) SP_HIP(hipMemcpy2DAsync(ds, per * sizeof(fe_t), v->d + off, cols * sizeof(fe_t), cols * sizeof(fe_t), full_rows, hipMemcpyDeviceToDevice, c->stream));
    if (n % cols) SP_HIP(hipMemcpyAsync(ds + full_rows * per, v->d + off + full_rows * cols, (n % cols) * sizeof(fe_t), hipMemcpyDeviceToDevice, c->stream));
    if (blinds) SP_HIP(hipMemcpyAsync(dbl, blinds, rows * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
    else SP_HIP(hipMemsetAsync(dbl, 0, rows * sizeof(fe_t), c->stream));
    SP_HIP(hipMemcpy2DAsync(ds + cols, per * sizeof(fe_t), dbl, sizeof(fe_t), sizeof(fe_t), rows, hipMemcpyDeviceToDevice, c->stream));
    c->timed("fixed_base", 32ull * total, [&] {
      launch_fixed_base_rows(c->stream, ds, total, ck->d_cktables, per, dout);
      hipLaunchKernelGGL(spk::k_sum_rows_of_points, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, dout, rows, (unsigned)per, drow);
    });
    out_rows.resize(rows);
    SP_HIP(hipMemcpyAsync(out_rows.data(), drow, rows * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
    SP_HIP(sp::stream_sync(c->stream));
    return SP_OK;
  }
  fe_t* canon = (fe_t*)c->workspace(sp_ctx::WS_COMMIT_CANON, n * sizeof(fe_t));
  unsigned* flags = (unsigned*)c->workspace(sp_ctx::WS_COMMIT_FLAGS, rows * 4);
  jac_t* rowsum = (jac_t*)c->workspace(sp_ctx::WS_COMMIT_ROWS, rows * sizeof(jac_t));
  if (!canon || !flags || !rowsum) return SP_ERR_NO_DEVICE;
  SP_HIP(hipMemsetAsync(flags, 0, rows * 4, c->stream));
  {
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(spk::k_to_canonical, dim3((unsigned)blocks), dim3(256), 0, c->stream, v->d + off, n, canon);
    hipLaunchKernelGGL(spk::k_classify_rows, dim3((unsigned)rows), dim3(256), 0, c->stream, canon, n, cols, flags);
  }
  c->timed("msm_binary_rows", 72ull * n, [&] {
    hipLaunchKernelGGL(spk::k_msm_binary_rows, dim3((unsigned)rows), dim3(256), 0, c->stream, canon, n, cols, ck->d_bases, flags, rowsum);
  });
  std::vector<unsigned> hflags(rows);
  std::vector<jac_t> msm_rows(rows);
  SP_HIP(hipMemcpyAsync(hflags.data(), flags, rows * 4, hipMemcpyDeviceToHost, c->stream));
  SP_HIP(hipMemcpyAsync(msm_rows.data(), rowsum, rows * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  // msm_10 / msm_small_rest / full msm rows (msm.rs:367-409, :187-222): digit path. One or two such rows take the single-MSM (latency) path;
  // more are batched over the row dimension (throughput path).
  std::vector<unsigned> full_rows, narrow_rows;
  for (size_t r = 0; r < rows; ++r)
    if (hflags[r] > 1u) ((hflags[r] & 4u) ? full_rows : narrow_rows).push_back((unsigned)r);
  for (int pass = 0; pass < 2; ++pass) {
    const std::vector<unsigned>& sel = pass == 0 ? full_rows : narrow_rows;
    const int windows = pass == 0 ? spk::MSM_MAX_WINDOWS : 9;
    if (sel.size() > 2) {
      // many rows over the one key: the fixed-base comb table (built once per key) when this commit alone justifies it or it exists already
      int have = 1;
      if (ck->d_comb || full_rows.size() + narrow_rows.size() >= sp::comb_min_rows()) {
        have = sp::comb_ensure(c, ck);
        if (have < 0) return have;
      }
      if (have == 0) rc = sp::comb_rows(c, ck, canon, cols, n, sel, pass == 0 ? 256 : 64, msm_rows);
      else rc = msm_rows_batched(c, canon, cols, n, sel, windows, ck->d_bases, msm_rows);
      if (rc) return rc;
    } else {
      for (unsigned r : sel) {
        size_t lo = (size_t)r * cols, len = (lo + cols <= n) ? cols : n - lo;
        if ((rc = msm_device(c, canon + lo, ck->d_bases, len, windows, &msm_rows[r]))) return rc;
      }
    }
  }
  if (blinds) {
    std::vector<jac_t> hb;
    if ((rc = fixed_base_rows(c, ck->d_htable, 1, blinds, rows, hb))) return rc;
    for (size_t r = 0; r < rows; ++r) msm_rows[r] = jac_add(msm_rows[r], hb[r]);
  }
  out_rows.swap(msm_rows);
  return SP_OK;
}
static int rows_out(std::vector<jac_t>& rows, uint64_t* out_rows_aff) {
  if (rows.empty()) return SP_OK;
  std::vector<aff_t> a(rows.size());
  normalize_batch(rows, a.data());
  memcpy(out_rows_aff, a.data(), rows.size() * sizeof(aff_t));
  return SP_OK;
}
int sp_hyrax_commit(sp_ctx* c, const sp_ck* ck, const sp_table* v, size_t off, size_t n, const uint64_t* blinds, int /*is_small: auto-detected*/,
                    uint64_t* out_rows_aff) {
  if (!blinds && n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "hyrax commit: null blinds");
  std::vector<jac_t> rows;
  int rc = commit_rows(c, ck, v, off, n, blinds, rows);
  return rc ? rc : rows_out(rows, out_rows_aff);
}
// PCS::commit_without_blind (hyrax_pc.rs:533-568): the cacheable, randomness-free part of a commitment
int sp_hyrax_commit_without_blind(sp_ctx* c, const sp_ck* ck, const sp_table* v, size_t off, size_t n, int /*is_small: auto-detected*/, uint64_t* out_rows_aff) {
  std::vector<jac_t> rows;
  int rc = commit_rows(c, ck, v, off, n, nullptr, rows);
  return rc ? rc : rows_out(rows, out_rows_aff);
}
// PCS::commit_incremental (hyrax_pc.rs:570-607): out[i] = raw[i] (identity beyond nraw) + MSM(delta row i) + h * blind[i]
int sp_hyrax_commit_incremental(sp_ctx* c, const sp_ck* ck, const uint64_t* raw_rows_aff, size_t nraw, const sp_table* delta, size_t off, size_t n,
                                const uint64_t* blinds, uint64_t* out_rows_aff) {
  if ((!blinds && n) || (!raw_rows_aff && nraw)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_incremental: null argument");
  std::vector<jac_t> rows;
  int rc = commit_rows(c, ck, delta, off, n, blinds, rows);
  if (rc) return rc;
  const aff_t* raw = reinterpret_cast<const aff_t*>(raw_rows_aff);
  for (size_t i = 0; i < rows.size() && i < nraw; ++i) rows[i] = jac_add_mixed(rows[i], raw[i]);
  return rows_out(rows, out_rows_aff);
}

// ---- vartime_scalar_mul / two-term fold / rerandomize ---------------------------------------------------------------------------------------
// wNAF-5 digits of a canonical scalar, least significant first (msm.rs:796-846)
static int wnaf5_digits(const fe_t& canon, signed char out[260]) {
  uint64_t limbs[4];
  for (int i = 0; i < 4; ++i) limbs[i] = (uint64_t)canon.v[2 * i] | ((uint64_t)canon.v[2 * i + 1] << 32);
  int len = 0;
  while (limbs[0] | limbs[1] | limbs[2] | limbs[3]) {
    int digit = 0;
    if (limbs[0] & 1) {
      digit = (int)(limbs[0] & 31);
      if (digit >= 16) {
        digit -= 32;
        uint64_t add = (uint64_t)(-digit), old = limbs[0];
        limbs[0] += add;
        if (limbs[0] < old)
          for (int k = 1; k < 4; ++k)
            if (++limbs[k] != 0) break;
      } else {
        limbs[0] -= (uint64_t)digit;
      }
    }
    out[len++] = (signed char)digit;
    for (int i = 0; i < 3; ++i) limbs[i] = (limbs[i] >> 1) | (limbs[i + 1] << 63);
    limbs[3] >>= 1;
  }
  return len;
}
static jac_t wnaf_mul_host(const aff_t& p, const signed char* d, int len) {
  jac_t tab[16];
  tab[0] = jac_from_affine(p);
  const jac_t dbl = jac_dbl(tab[0]);
  for (int k = 1; k < 16; ++k) tab[k] = jac_add(tab[k - 1], dbl);
  jac_t acc = jac_identity();
  bool started = false;
  for (int k = len - 1; k >= 0; --k) {
    if (started) acc = jac_dbl(acc);
    if (d[k] > 0) {
      started = true;
      acc = jac_add(acc, tab[(d[k] - 1) / 2]);
    } else if (d[k] < 0) {
      started = true;
      jac_t q = tab[(-d[k] - 1) / 2];
      q.y = fe_neg<B>(q.y);
      acc = jac_add(acc, q);
    }
  }
  return acc;
}
static const size_t WNAF_HOST_MAX = 48;  // below this many points the host side finishes sooner than one device lane per point
static int scalar_mul_rows(sp_ctx* c, const uint64_t* points_aff, size_t n, const uint64_t scalar[4], std::vector<jac_t>& out) {
  out.assign(n, jac_identity());
  if (n == 0) return SP_OK;
  fe_t sc;
  memcpy(&sc, scalar, 32);
  spk::WnafArgs w;
  w.len = wnaf5_digits(fe_to_canonical<S>(sc), w.d);
  const aff_t* pts = reinterpret_cast<const aff_t*>(points_aff);
  if (n <= WNAF_HOST_MAX) {
    // ~90 us per point on one core (256 doublings + ~51 additions): a handful of points (the 16-row commitment folds of the NeutronNova opening)
    // go over up to eight short-lived threads
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 8) nt = 8;
    if (nt > n) nt = (unsigned)n;
    sp::WalkPool& pool = sp::WalkPool::get();
    if (n >= 4 && pool.walkers() > 0) {  // (the polling threads instead of threads created here: see msm_shared_weights_on)
      struct M {
        const aff_t* pts;
        const spk::WnafArgs* w;
        std::vector<jac_t>* out;
        size_t n;
      } m{pts, &w, &out, n};
      pool.keep_hot(2000);
      pool.run((unsigned)std::min<size_t>(n, (size_t)pool.walkers() + 1), [](void* a, unsigned p, unsigned np) {
        M& x = *static_cast<M*>(a);
        for (size_t i = x.n * p / np; i < x.n * (p + 1) / np; ++i) (*x.out)[i] = wnaf_mul_host(x.pts[i], x.w->d, x.w->len);
      }, &m);
      return SP_OK;
    }
    if (n < 4 || nt < 2) {
      for (size_t i = 0; i < n; ++i) out[i] = wnaf_mul_host(pts[i], w.d, w.len);
      return SP_OK;
    }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k)
      th.emplace_back([&, k] {
        for (size_t i = k; i < n; i += nt) out[i] = wnaf_mul_host(pts[i], w.d, w.len);
      });
    for (auto& t : th) t.join();
    return SP_OK;
  }
  DevBuf dp, dout;
  int rc;
  if ((rc = dp.alloc(n * sizeof(aff_t))) || (rc = dout.alloc(n * sizeof(jac_t)))) return rc;
  SP_HIP(hipMemcpyAsync(dp.p, pts, n * sizeof(aff_t), hipMemcpyHostToDevice, c->stream));
  c->timed("wnaf_rows", 64ull * n, [&] { hipLaunchKernelGGL(spk::k_wnaf_rows, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, dp.as<aff_t>(), n, w, dout.as<jac_t>()); });
  SP_HIP(hipMemcpyAsync(out.data(), dout.p, n * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  return SP_OK;
}
int sp_vartime_scalar_mul(sp_ctx* c, const uint64_t* points_aff, size_t n, const uint64_t scalar[4], uint64_t* out_aff) {
  if (n && (!points_aff || !scalar || !out_aff)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_vartime_scalar_mul: null argument");
  std::vector<jac_t> pts;
  int rc = scalar_mul_rows(c, points_aff, n, scalar, pts);
  if (rc) return rc;
  std::vector<aff_t> a(n);
  normalize_batch(pts, a.data());
  if (n) memcpy(out_aff, a.data(), n * sizeof(aff_t));
  return SP_OK;
}
int sp_fold_commitments2(sp_ctx* c, const uint64_t* p_rows_aff, const uint64_t* q_rows_aff, size_t rows, const uint64_t w[4], uint64_t* out_rows_aff) {
  if (rows && (!p_rows_aff || !q_rows_aff || !w || !out_rows_aff)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fold_commitments2: null argument");
  std::vector<jac_t> pts;
  int rc = scalar_mul_rows(c, q_rows_aff, rows, w, pts);
  if (rc) return rc;
  const aff_t* p = reinterpret_cast<const aff_t*>(p_rows_aff);
  for (size_t i = 0; i < rows; ++i) pts[i] = jac_add_mixed(pts[i], p[i]);
  std::vector<aff_t> a(rows);
  normalize_batch(pts, a.data());
  if (rows) memcpy(out_rows_aff, a.data(), rows * sizeof(aff_t));
  return SP_OK;
}
// ---- the two-term fold with a late weight: doubling ladders of q's rows built ahead on the walkers (walk_pool.hpp) ------------------------------------------
struct sp_fold2_job {
  static constexpr size_t LAD = 257;  // 2^j q for j = 0 .. 256: a non-adjacent form of a 256-bit scalar has up to 257 digits
  size_t rows = 0;
  std::vector<aff_t> q, ladder;  // ladder[i * LAD + j] = 2^j q_i (affine)
  sp::WalkPool::Batch* batch = nullptr;
  bool built = false;
  // the second region (finish): the weight's digits, p's rows (copied: a straggler may still read them) and the results
  signed char naf[LAD + 1];
  int naf_len = 0;
  std::vector<jac_t> out;
  std::vector<aff_t> p;
  // one reference for the owner and one for every region a straggler may still be inside (walk_pool.hpp collect_fn / end_fn)
  std::atomic<int> refs{1};
};
static void fold2_unref(void* a) {
  sp_fold2_job* J = static_cast<sp_fold2_job*>(a);
  if (J->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete J;
}
static const long long FOLD2_LATE_NS = 60000;  // a part is ~13-60 us of arithmetic: one that is later than this has lost its core
// (both part functions are idempotent: a late part that the owner runs again writes the same values to the same places as its straggler)
static void fold2_ladder_part(void* arg, unsigned part, unsigned np) {
  sp_fold2_job& J = *static_cast<sp_fold2_job*>(arg);
  constexpr size_t LAD = sp_fold2_job::LAD;
  std::vector<jac_t> jp(LAD);
  std::vector<aff_t> row(LAD);
  for (size_t i = J.rows * part / np; i < J.rows * (part + 1) / np; ++i) {
    jp[0] = jac_from_affine(J.q[i]);
    for (size_t j = 1; j < LAD; ++j) jp[j] = jac_dbl(jp[j - 1]);
    normalize_batch(jp, row.data());  // (an identity row stays a row of identities)
    memcpy(J.ladder.data() + i * LAD, row.data(), LAD * sizeof(aff_t));
  }
}
static void fold2_sum_part(void* arg, unsigned part, unsigned np) {
  sp_fold2_job& J = *static_cast<sp_fold2_job*>(arg);
  constexpr size_t LAD = sp_fold2_job::LAD;
  for (size_t i = J.rows * part / np; i < J.rows * (part + 1) / np; ++i) {
    const aff_t* L = J.ladder.data() + i * LAD;
    xyzz_t acc = xyzz_identity();
    for (int j = 0; j < J.naf_len; ++j)
      if (J.naf[j]) acc = xyzz_add_mixed(acc, J.naf[j] > 0 ? L[j] : aff_neg(L[j]));
    acc = xyzz_add_mixed(acc, J.p[i]);
    J.out[i] = xyzz_to_jac(acc);
  }
}
int sp_fold_commitments2_begin(sp_ctx* c, const uint64_t* q_rows_aff, size_t rows, sp_fold2_job** out) {
  (void)c;
  if (rows && !q_rows_aff) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fold_commitments2_begin: null argument");
  sp_fold2_job* J = new sp_fold2_job();
  J->rows = rows;
  J->q.resize(rows);
  if (rows) memcpy(J->q.data(), q_rows_aff, rows * sizeof(aff_t));
  sp::WalkPool& pool = sp::WalkPool::get();
  if (rows && pool.walkers() > 0) {
    J->ladder.resize(rows * sp_fold2_job::LAD);
    pool.keep_hot(4000);
    J->batch = pool.post_fn((unsigned)std::min<size_t>(rows, sp::WalkPool::MAX_PARTS), fold2_ladder_part, J);
    if (J->batch) J->refs.fetch_add(1, std::memory_order_relaxed);  // the ladder region's
    else J->ladder.clear();                                         // (no slot: the plain form at finish)
  }
  *out = J;
  return SP_OK;
}
// the ladders are in (late parts built again by the caller); the region's reference goes with its last straggler
static void fold2_collect_ladders(sp_fold2_job* J) {
  if (!J->batch) return;
  sp::WalkPool& pool = sp::WalkPool::get();
  (void)pool.collect_fn(J->batch, FOLD2_LATE_NS);
  pool.end_fn(J->batch, fold2_unref, J);
  J->batch = nullptr;
  J->built = true;
}
int sp_fold_commitments2_finish(sp_ctx* c, sp_fold2_job* J, const uint64_t* p_rows_aff, const uint64_t w[4], uint64_t* out_rows_aff) {
  struct Unref {
    sp_fold2_job* j;
    ~Unref() { fold2_unref(j); }
  } owner{J};
  sp::WalkPool& pool = sp::WalkPool::get();
  fold2_collect_ladders(J);
  if (J->rows && (!p_rows_aff || !w || !out_rows_aff)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fold_commitments2_finish: null argument");
  if (!J->built) return sp_fold_commitments2(c, p_rows_aff, reinterpret_cast<const uint64_t*>(J->q.data()), J->rows, w, out_rows_aff);
  // non-adjacent form of the canonical weight: digit j in {-1, 0, 1}, no two adjacent non-zero
  fe_t sc;
  memcpy(&sc, w, 32);
  const fe_t k0 = fe_to_canonical<S>(sc);
  uint32_t k[9];
  for (int i = 0; i < 8; ++i) k[i] = k0.v[i];
  k[8] = 0;
  int len = 0;
  auto is_zero = [&] {
    uint32_t o = 0;
    for (int i = 0; i < 9; ++i) o |= k[i];
    return o == 0;
  };
  while (!is_zero() && len < (int)sp_fold2_job::LAD) {
    int d = 0;
    if (k[0] & 1u) {
      d = 2 - (int)(k[0] & 3u);  // 1 or -1
      if (d > 0) {
        k[0] -= 1u;  // (k odd: no borrow)
      } else {       // k += 1
        for (int i = 0; i < 9; ++i)
          if (++k[i] != 0u) break;
      }
    }
    J->naf[len++] = (signed char)d;
    for (int i = 0; i < 8; ++i) k[i] = (k[i] >> 1) | (k[i + 1] << 31);
    k[8] >>= 1;
  }
  J->naf_len = len;
  J->out.assign(J->rows, jac_identity());
  J->p.resize(J->rows);
  if (J->rows) memcpy(J->p.data(), p_rows_aff, J->rows * sizeof(aff_t));
  pool.keep_hot(2000);
  const unsigned np = (unsigned)std::min<size_t>(J->rows, (size_t)pool.walkers() + 1);
  sp::WalkPool::Batch* b2 = np > 1 ? pool.post_fn(np, fold2_sum_part, J) : nullptr;
  std::vector<jac_t> res;
  if (b2) {
    J->refs.fetch_add(1, std::memory_order_relaxed);
    (void)pool.collect_fn(b2, FOLD2_LATE_NS);
    res = J->out;  // (every entry has been written by now - by its part or by the caller's second run of it; a straggler rewrites the same values)
    pool.end_fn(b2, fold2_unref, J);
  } else {
    for (unsigned p = 0; p < (np ? np : 1u); ++p) fold2_sum_part(J, p, np ? np : 1u);
    res = J->out;
  }
  std::vector<aff_t> a(J->rows);
  normalize_batch(res, a.data());
  if (J->rows) memcpy(out_rows_aff, a.data(), J->rows * sizeof(aff_t));
  return SP_OK;
}
void sp_fold_commitments2_drop(sp_fold2_job* J) {
  if (!J) return;
  if (J->batch) {
    sp::WalkPool& pool = sp::WalkPool::get();
    (void)pool.collect_fn(J->batch, FOLD2_LATE_NS);
    pool.end_fn(J->batch, fold2_unref, J);
    J->batch = nullptr;
  }
  fold2_unref(J);
}

int sp_hyrax_rerandomize(sp_ctx* c, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const uint64_t* r_old, const uint64_t* r_new, uint64_t* out_rows_aff) {
  if (rows && (!comm_rows_aff || !r_old || !r_new || !out_rows_aff)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "rerandomize_commitment: null argument");
  std::vector<fe_t> diff(rows);
  for (size_t i = 0; i < rows; ++i) {
    fe_t a, b;
    memcpy(&a, r_new + 4 * i, 32);
    memcpy(&b, r_old + 4 * i, 32);
    diff[i] = fe_sub<S>(a, b);
  }
  std::vector<jac_t> pts(rows);
  if (rows <= FIXED_BASE_HOST_MAX) {
    for (size_t i = 0; i < rows; ++i) pts[i] = ck_mul_host(ck, ck->n_tables - 1, diff[i]);
  } else {
    int rc = fixed_base_rows(c, ck->d_htable, 1, reinterpret_cast<const uint64_t*>(diff.data()), rows, pts);
    if (rc) return rc;
  }
  const aff_t* p = reinterpret_cast<const aff_t*>(comm_rows_aff);
  std::vector<aff_t> a(rows);
  // hundreds of rows (every precommitted row of every instance of a NeutronNova batch: ~500 at config 3, 0.18 ms of additions and normalisation on one
  // thread at the head of every prove): ranges of rows on the polling host threads, each with an inversion of its own
  struct Part {
    std::vector<jac_t>* pts;
    const aff_t* p;
    aff_t* a;
    size_t rows;
  } part{&pts, p, a.data(), rows};
  auto fn = [](void* arg, unsigned k, unsigned np) {
    Part& P = *static_cast<Part*>(arg);
    const size_t lo = P.rows * k / np, hi = P.rows * (k + 1) / np;
    std::vector<jac_t> mine(hi - lo);
    for (size_t i = lo; i < hi; ++i) mine[i - lo] = jac_add_mixed((*P.pts)[i], P.p[i]);
    normalize_batch(mine, P.a + lo);
  };
  sp::WalkPool& pool = sp::WalkPool::get();
  const unsigned np = rows >= 64 && pool.walkers() > 0 ? (unsigned)std::min<size_t>(rows / 32, (size_t)pool.walkers() + 1) : 1u;
  if (np > 1) pool.run(np, fn, &part);
  else fn(&part, 0, 1);
  if (rows) memcpy(out_rows_aff, a.data(), rows * sizeof(aff_t));
  return SP_OK;
}

using sp::launch_rowmat_vec;  // bind_with_delayed (hyrax_pc.rs:38-54): capi_bulk.hip
int sp_rowmat_vec(sp_ctx* c, const sp_table* poly, size_t rows, size_t cols, const uint64_t* L, uint64_t* out) {
  if (rows * cols > poly->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "bind_with_delayed: poly shorter than rows*cols");
  if (rows == 0 || cols == 0) return SP_OK;
  size_t splits = rows < 64 ? rows : 64;
  fe_t* dL = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_L, rows * sizeof(fe_t));
  fe_t* part = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_PART, splits * cols * sizeof(fe_t));
  fe_t* dout = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_OUT, cols * sizeof(fe_t));
  if (!dL || !part || !dout) return SP_ERR_NO_DEVICE;
  SP_HIP(hipMemcpyAsync(dL, L, rows * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  c->timed("rowmat_vec", 32ull * (rows * cols + rows + cols), [&] { launch_rowmat_vec(c->stream, poly->d, rows, cols, dL, part, splits, dout); });
  SP_HIP(hipMemcpyAsync(out, dout, cols * sizeof(fe_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  return SP_OK;
}

int sp_msm_ck(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t* blind, uint64_t out_aff[8]) {
  if (n > ck->num_cols) return fail(SP_ERR_INVALID_INPUT_LENGTH, "MSM: Coefficients and bases must have the same length");
  fe_t* canon;
  int rc;
  if ((rc = upload_canonical(c, scalars, n, &canon))) return rc;
  jac_t r;
  if ((rc = msm_device(c, canon, ck->d_bases, n, spk::MSM_MAX_WINDOWS, &r))) return rc;
  if (blind) {
    fe_t b;
    memcpy(&b, blind, 32);
    r = jac_add(r, ck_mul_host(ck, ck->n_tables - 1, b));
  }
  store_aff(out_aff, jac_to_affine(r));
  return SP_OK;
}

// Asynchronous form of sp_msm_ck on the auxiliary stream: begin() enqueues the device work and returns, finish() waits,
// runs the host-side Horner tail and adds h * blind.
struct sp_msm_job {
  MsmPending pend;
};
int sp_msm_ck_begin(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, sp_msm_job** out) {
  if (n > ck->num_cols) return fail(SP_ERR_INVALID_INPUT_LENGTH, "MSM: Coefficients and bases must have the same length");
  fe_t* canon;
  int rc;
  if ((rc = upload_canonical(c, scalars, n, &canon, 1))) return rc;
  sp_msm_job* job = new sp_msm_job();
  if ((rc = msm_launch(c, canon, ck->d_bases, n, spk::MSM_MAX_WINDOWS, 1, &job->pend))) {
    delete job;
    return rc;
  }
  *out = job;
  return SP_OK;
}
int sp_msm_ck_range_begin(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t first, size_t n, sp_msm_job** out) {
  if (first > ck->num_cols || n > ck->num_cols - first) return fail(SP_ERR_INVALID_INPUT_LENGTH, "MSM: the base range lies outside the key");
  fe_t* canon;
  int rc;
  if ((rc = upload_canonical(c, scalars, n, &canon, 1))) return rc;
  sp_msm_job* job = new sp_msm_job();
  if ((rc = msm_launch(c, canon, ck->d_bases + first, n, spk::MSM_MAX_WINDOWS, 1, &job->pend))) {
    delete job;
    return rc;
  }
  *out = job;
  return SP_OK;
}
int sp_msm_ck_finish(sp_ctx* c, const sp_ck* ck, sp_msm_job* job, const uint64_t* blind, uint64_t out_aff[8]) {
  jac_t r;
  int rc = msm_finish(c, &job->pend, &r);
  delete job;
  if (rc) return rc;
  if (blind) {
    fe_t b;
    memcpy(&b, blind, 32);
    r = jac_add(r, ck_mul_host(ck, ck->n_tables - 1, b));
  }
  store_aff(out_aff, jac_to_affine(r));
  return SP_OK;
}

// ---- MSM over caller-supplied points with eq-table weights (the homomorphic form of HyraxPCS::prove's comm_LZ) ---------------------------
struct sp_points {
  aff_t* d = nullptr;
  size_t n = 0, cap = 0;
};
int sp_points_upload(sp_ctx* c, const uint64_t* aff, size_t n, sp_points** io) {
  if (!io || (!aff && n)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_points_upload: null argument");
  sp_points* p = *io ? *io : new sp_points();
  if (p->cap < n) {
    if (p->d) {
      sp::stream_sync(c->stream2);
      hipFree(p->d);
      p->d = nullptr;
      p->cap = 0;
    }
    if (hipMalloc((void**)&p->d, (n ? n : 1) * sizeof(aff_t)) != hipSuccess) {
      if (!*io) delete p;
      return fail(SP_ERR_NO_DEVICE, "hipMalloc failed for a point vector");
    }
    p->cap = n ? n : 1;
  }
  p->n = n;
  *io = p;
  if (n) {
    SP_HIP(hipMemcpyAsync(p->d, aff, n * sizeof(aff_t), hipMemcpyHostToDevice, c->stream2));
    SP_HIP(sp::stream_sync(c->stream2));  // `aff` is a borrowed host buffer
  }
  return SP_OK;
}
void sp_points_free(sp_points* p) {
  if (!p) return;
  if (p->d) hipFree(p->d);
  delete p;
}
// vartime_multiscalar_mul on operands resident in HBM: scalars = n elements of a table at `off` (Montgomery form), bases = points [first, first + n)
int sp_msm_points(sp_ctx* c, const sp_table* scalars, size_t off, size_t n, const sp_points* bases, size_t first, int window, uint64_t out_aff[8]) {
  if (!c || !scalars || !bases || !out_aff) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_msm_points: null argument");
  if (off + n > scalars->cap || first + n > bases->n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_msm_points: range exceeds the operands");
  jac_t r = jac_identity();
  if (n) {
    fe_t* canon = (fe_t*)c->workspace(sp_ctx::WS_SCALARS_CANON, n * sizeof(fe_t));
    if (!canon) return SP_ERR_NO_DEVICE;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(spk::k_to_canonical, dim3((unsigned)blocks), dim3(256), 0, c->stream, scalars->d + off, n, canon);
    int rc;
    if (window || n >= sp::PIPPENGER_MIN) rc = sp::msm_pippenger(c, canon, bases->d + first, n, true, window, &r);
    else rc = msm_device(c, canon, bases->d + first, n, spk::MSM_MAX_WINDOWS, &r);
    if (rc) return rc;
  }
  store_aff(out_aff, jac_to_affine(r));
  return SP_OK;
}
// eq table of k variables on the host, r[0] on the index MSB (EqPolynomial::evals_from_points, src/polys/eq.rs:59-93)
static void eq_table_host(const fe_t* r, size_t k, fe_t* out) {
  out[0] = fe_one<spk::SF>();
  for (size_t j = 0; j < k; ++j) {
    const size_t half = (size_t)1 << j;
    for (size_t i = half; i-- > 0;) {
      const fe_t hi = fe_mul<spk::SF>(out[i], r[j]);
      out[2 * i] = fe_sub<spk::SF>(out[i], hi);
      out[2 * i + 1] = hi;
    }
  }
}
int sp_msm_eq_begin(sp_ctx* c, const sp_points* pts, const uint64_t* r, size_t ell, sp_msm_job** out) {
  if (!pts || !out || (!r && ell)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_msm_eq_begin: null argument");
  if (ell > 20) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_msm_eq_begin: more than 2^20 points");
  const size_t n = (size_t)1 << ell;
  if (n != pts->n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "MSM: Coefficients and bases must have the same length");
  std::vector<fe_t> rr(ell);
  for (size_t i = 0; i < ell; ++i) memcpy(&rr[i], r + 4 * i, 32);
  fe_t* canon = nullptr;
  int rc;
  if (ell <= 10) {
    canon = (fe_t*)c->workspace(sp_ctx::WS_SCALARS_CANON, n * sizeof(fe_t), 1);
    if (!canon) return SP_ERR_NO_DEVICE;
    spk::EqTensorArgs a;
    const size_t hb = ell / 2, lb = ell - hb;
    eq_table_host(rr.data(), hb, a.left);
    eq_table_host(rr.data() + hb, lb, a.right);
    a.lo_bits = (int)lb;
    a.n = (unsigned)n;
    hipLaunchKernelGGL(spk::k_eq_tensor<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream2, a, canon);
  } else {
    std::vector<fe_t> w(n);
    eq_table_host(rr.data(), ell, w.data());
    if ((rc = upload_canonical(c, reinterpret_cast<const uint64_t*>(w.data()), n, &canon, 1))) return rc;
  }
  sp_msm_job* job = new sp_msm_job();
  if ((rc = msm_launch(c, canon, pts->d, n, spk::MSM_MAX_WINDOWS, 1, &job->pend))) {
    delete job;
    return rc;
  }
  *out = job;
  return SP_OK;
}
// bind_with_delayed with L = eq(r, .) generated on the device, on a stream of its own; the result lands in pinned memory
struct sp_vec_job {
  size_t cols = 0;
  const fe_t* d_out = nullptr;  // the product on the device
  const fe_t* d_add = nullptr;  // the addend of the _scaled finish on the device (null: none given)
  unsigned armed = 0;           // != 0: a k_scale_add_wait with this sequence number is queued behind the product and waits for the scale
};
namespace {
// out[i] = scale * x[i] + add[i] straight into mapped pinned host memory, one flag per block behind a system-scope fence: the host polls the flags, no
// stream synchronisation (z_vec = r * LZ + d of InnerProductArgumentLinear::prove, ipa.rs:160-163, is the last step of a prove: 2048 products were 33 us
// on three host threads)
__global__ void __launch_bounds__(256) k_scale_add_to_host(const fe_t* __restrict__ x, const fe_t* __restrict__ add, fe_t scale, size_t n, fe_t* __restrict__ out,
                                                           volatile unsigned* __restrict__ flags, unsigned seq) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fe_add<S>(fe_mul<S>(scale, x[i]), add[i]);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) flags[blockIdx.x] = seq;
}
// The same launched AHEAD of its scale: queued behind the product (and the upload of the addend), every block's first wave waits for the scale in the
// mapped control line - words 0..7 the scale, 8 = sequence number, 9 = sequence + sum of the words, 11 = sequence * K + position-weighted sum, 12 = the
// sequence number again (a poll that straddles the host's stores fails a check and is repeated), 10 = abort (a sequence number: the job was finished
// without a scale) - and the sum goes out as above. The launch call, the
// dispatch and the kernel's start-up (~10 us of the 20 that z_vec took behind the IPA's challenge, the last step of a prove) happen while the opening's
// walks still run. Gives up after SCALE_WAIT_TICKS (flag = ~seq: the host then launches the ordinary kernel).
// 20 ms at the 100 MHz wall clock. Short on purpose: the scale normally follows within a few hundred microseconds, a waiter that gives up only costs the
// ordinary launch behind the scale (scale_add_fire falls back to it), and while it waits every device-wide synchronisation of the process - a hipFree
// between the sum-check and the opening, say - waits with it.
constexpr unsigned long long SCALE_WAIT_TICKS = 2000000ull;
constexpr unsigned SCALE_CHK_K = 0x9E3779B1u;
__global__ void __launch_bounds__(256) k_scale_add_wait(const fe_t* __restrict__ x, const fe_t* __restrict__ add, const unsigned* __restrict__ ctrl, size_t n,
                                                        fe_t* __restrict__ out, volatile unsigned* __restrict__ flags, unsigned seq) {
  __shared__ fe_t sc_sh;
  __shared__ int ok_sh;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe_t xv = fe_zero(), av = fe_zero();
  if (i < n) {  // operands first: they are in registers when the scale arrives
    xv = x[i];
    av = add[i];
  }
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    int ok = 0;
    for (;;) {
      unsigned w = 0;
      if (lane < 13) w = __hip_atomic_load(ctrl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned sq = __shfl(w, 8, 64), chk = __shfl(w, 9, 64), ab = __shfl(w, 10, 64), chk2 = __shfl(w, 11, 64), sq2 = __shfl(w, 12, 64);
      unsigned sum = lane < 8 ? w : 0u, wsum = lane < 8 ? (unsigned)(lane + 1) * w : 0u;
#pragma unroll
      for (int m = 4; m >= 1; m >>= 1) {
        sum += __shfl_xor(sum, m, 64);
        wsum += __shfl_xor(wsum, m, 64);
      }
      sum = __shfl(sum, 0, 64);
      wsum = __shfl(wsum, 0, 64);
      // the challenge mailbox's line format (kernels_poly.hpp): the sequence number twice, around the payload, and two independent check words - a torn
      // read of the write-combined line would have to keep both sums (one additive word let words that cancel mod 2^32 through)
      if (sq == seq && sq2 == seq && chk == seq + sum && chk2 == seq * SCALE_CHK_K + wsum) {
        if (lane < 8) sc_sh.v[lane] = w;
        ok = 1;
        break;
      }
      if (ab == seq || wall_clock64() - t0 > SCALE_WAIT_TICKS) break;
      __builtin_amdgcn_s_sleep(16);
    }
    if (lane == 0) ok_sh = ok;
  }
  __syncthreads();
  if (!ok_sh) {
    if (threadIdx.x == 0) flags[blockIdx.x] = ~seq;
    return;
  }
  const fe_t scale = sc_sh;
  if (i < n) out[i] = fe_add<S>(fe_mul<S>(scale, xv), av);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) flags[blockIdx.x] = seq;
}
}  // namespace
// landing buffer of the product | staging of the addend | landing of the scaled sum | its per-block flags: one mapped pinned allocation, grow-only
static int ensure_pinned_vec(sp_ctx* c, size_t cols) {
  if (!c->stream3) SP_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
  if (!c->vec_ev) SP_HIP(hipEventCreateWithFlags(&c->vec_ev, hipEventDisableTiming));
  if (c->h_pinned_vec_bytes < 3 * cols * sizeof(fe_t) + VEC_FLAG_BYTES + VEC_CTRL_BYTES) {
    if (c->h_pinned_vec) {
      SP_HIP(sp::stream_sync(c->stream3));
      SP_HIP(sp::stream_sync(c->stream2));
      hipHostFree(c->h_pinned_vec);
    }
    c->h_pinned_vec = nullptr;
    c->h_pinned_vec_bytes = 0;
    SP_HIP(hipHostMalloc(&c->h_pinned_vec, 3 * cols * sizeof(fe_t) + VEC_FLAG_BYTES + VEC_CTRL_BYTES, hipHostMallocMapped));
    memset(c->h_pinned_vec, 0, 3 * cols * sizeof(fe_t) + VEC_FLAG_BYTES + VEC_CTRL_BYTES);
    c->h_pinned_vec_bytes = 3 * cols * sizeof(fe_t) + VEC_FLAG_BYTES + VEC_CTRL_BYTES;
    c->h_pinned_vec_cols = cols;
  }
  return SP_OK;
}
// The armed form: _arm queues k_scale_add_wait on `st` (behind whatever produces x and add there) and returns its sequence number; _fire hands it the
// scale through the control line and collects the sum; _abort releases a waiter whose scale will never come. One armed job per context at a time.
// The control line lives where the challenge mailbox does (core.hpp): in fine-grained DEVICE memory that the host writes through the PCIe BAR when the
// system has a large BAR - the waiting blocks then poll their own memory - and otherwise in the mapped page (every poll a bus read: eight contexts'
// waiters polling host memory took a third off the eight-context throughput, 0.62 -> 0.84 ms per proof).
static const size_t VEC_CTRL_MAIL_WORD = 512;  // word offset of the line in the 4 KiB mailbox page (ring: words 0..127, diagnostics: 256..259)
static volatile unsigned* vec_ctrl(sp_ctx* c) {
  if (c->mail_dev) return reinterpret_cast<volatile unsigned*>(c->h_mail) + VEC_CTRL_MAIL_WORD;
  return reinterpret_cast<volatile unsigned*>(reinterpret_cast<char*>(c->h_pinned_vec) + 3 * c->h_pinned_vec_cols * sizeof(fe_t) + VEC_FLAG_BYTES);
}
static void vec_ctrl_flush(sp_ctx* c) {
  if (c->mail_dev) __builtin_ia32_sfence();  // BAR memory is write-combining: push the line out now
  else std::atomic_thread_fence(std::memory_order_seq_cst);
}
static bool scale_add_armed_enabled() {
  static const bool on = [] {
    const char* e = getenv("SPARTAN_ZVEC_ARMED");  // "0": launch behind the scale (A/B)
    return !(e && e[0] == '0');
  }();
  return on;
}
static unsigned scale_add_arm(sp_ctx* c, hipStream_t st, const fe_t* dx, const fe_t* da, size_t cols) {
  const size_t nblocks = (cols + 255) / 256;
  // One context in the process = one prove at a time (the latency case this is for). With several, streams of different contexts share hardware queues and
  // a kernel that WAITS in one holds up whatever another context queued behind it: eight contexts went from 0.61 to 0.83-0.94 ms per proof with it.
  if (!scale_add_armed_enabled() || sp::live_contexts() != 1 || nblocks * sizeof(unsigned) > VEC_FLAG_BYTES || cols > c->h_pinned_vec_cols) return 0;
  if (++c->vec_seq == 0) ++c->vec_seq;
  const unsigned seq = c->vec_seq;
  void* d_base = nullptr;
  if (hipHostGetDevicePointer(&d_base, c->h_pinned_vec, 0) != hipSuccess) return 0;
  fe_t* d_hout = reinterpret_cast<fe_t*>(d_base) + 2 * c->h_pinned_vec_cols;
  unsigned* d_flags = reinterpret_cast<unsigned*>(reinterpret_cast<fe_t*>(d_base) + 3 * c->h_pinned_vec_cols);
  const unsigned* d_ctrl = c->mail_dev ? c->d_mail + VEC_CTRL_MAIL_WORD : reinterpret_cast<const unsigned*>(reinterpret_cast<char*>(d_flags) + VEC_FLAG_BYTES);
  hipLaunchKernelGGL(k_scale_add_wait, dim3((unsigned)nblocks), dim3(256), 0, st, dx, da, d_ctrl, cols, d_hout, d_flags, seq);
  return seq;
}
static void scale_add_abort(sp_ctx* c, unsigned seq) {
  if (!seq || !c->h_pinned_vec) return;
  volatile unsigned* ctl = vec_ctrl(c);
  ctl[10] = seq;
  vec_ctrl_flush(c);
}
static int scale_add_to_host(sp_ctx* c, hipStream_t st, const fe_t* dx, const fe_t* da, const fe_t& sc, size_t cols, uint64_t* out, const char* site);
static int scale_add_fire(sp_ctx* c, hipStream_t st, unsigned seq, const fe_t* dx, const fe_t* da, const fe_t& sc, size_t cols, uint64_t* out, const char* site) {
  const size_t nblocks = (cols + 255) / 256;
  volatile unsigned* ctl = vec_ctrl(c);
  unsigned sum = 0, wsum = 0;
  for (int i = 0; i < 8; ++i) {
    ctl[i] = sc.v[i];
    sum += sc.v[i];
    wsum += (unsigned)(i + 1) * sc.v[i];
  }
  ctl[9] = seq + sum;
  ctl[11] = seq * SCALE_CHK_K + wsum;
  // the payload before the sequence words: a C++ fence does not order write-combined stores to BAR memory, sfence does
  if (c->mail_dev) __builtin_ia32_sfence();
  else std::atomic_thread_fence(std::memory_order_release);
  ctl[8] = seq;
  ctl[12] = seq;
  vec_ctrl_flush(c);
  fe_t* h_out = reinterpret_cast<fe_t*>(c->h_pinned_vec) + 2 * c->h_pinned_vec_cols;
  volatile unsigned* h_flags = reinterpret_cast<volatile unsigned*>(reinterpret_cast<fe_t*>(c->h_pinned_vec) + 3 * c->h_pinned_vec_cols);
  bool synced = false;
  for (size_t b = 0; b < nblocks; ++b) {
    for (long spins = 0; h_flags[b] != seq; ++spins) {
      if (h_flags[b] == ~seq) {  // the waiter's watchdog ran out before the scale came (a late caller): the ordinary launch behind the scale
        for (int i = 0; i < 13; ++i)
          if (i != 10) ctl[i] = 0;
        sp::slow_note(site, -1);
        return scale_add_to_host(c, st, dx, da, sc, cols, out, site);
      }
      if (spins > 4000000) {
        sp::slow_note(site, spins);
        if (synced) return fail(SP_ERR_INTERNAL, "bind_with_delayed: the scaled sum did not arrive");
        SP_HIP(sp::stream_sync(st));  // e.g. under a profiler
        synced = true;
        spins = 0;
      }
      sp::relax();
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  memcpy(out, h_out, cols * sizeof(fe_t));
  for (int i = 0; i < 13; ++i)
    if (i != 10) ctl[i] = 0;  // (the scale is the IPA's public challenge; cleared all the same)
  return SP_OK;
}
// scale * x + addend -> mapped host memory, polled through the per-block arrival flags (the tail of sp_rowmat_vec_eq_finish_scaled and of an announced opening)
static int scale_add_to_host(sp_ctx* c, hipStream_t st, const fe_t* dx, const fe_t* da, const fe_t& sc, size_t cols, uint64_t* out, const char* site) {
  const size_t nblocks = (cols + 255) / 256;
  if (nblocks * sizeof(unsigned) > VEC_FLAG_BYTES) return fail(SP_ERR_INVALID_INPUT_LENGTH, "bind_with_delayed (scaled): more than 2^18 columns");
  if (++c->vec_seq == 0) ++c->vec_seq;
  const unsigned seq = c->vec_seq;
  fe_t* h_out = reinterpret_cast<fe_t*>(c->h_pinned_vec) + 2 * c->h_pinned_vec_cols;
  volatile unsigned* h_flags = reinterpret_cast<volatile unsigned*>(reinterpret_cast<fe_t*>(c->h_pinned_vec) + 3 * c->h_pinned_vec_cols);
  void* d_base = nullptr;
  SP_HIP(hipHostGetDevicePointer(&d_base, c->h_pinned_vec, 0));
  fe_t* d_hout = reinterpret_cast<fe_t*>(d_base) + 2 * c->h_pinned_vec_cols;
  unsigned* d_flags = reinterpret_cast<unsigned*>(reinterpret_cast<fe_t*>(d_base) + 3 * c->h_pinned_vec_cols);
  hipLaunchKernelGGL(k_scale_add_to_host, dim3((unsigned)nblocks), dim3(256), 0, st, dx, da, sc, cols, d_hout, d_flags, seq);
  bool synced = false;
  for (size_t b = 0; b < nblocks; ++b) {
    for (long spins = 0; h_flags[b] != seq; ++spins) {
      if (spins > 4000000) {
        sp::slow_note(site, spins);
        if (synced) return fail(SP_ERR_INTERNAL, "bind_with_delayed: the scaled sum did not arrive");
        SP_HIP(sp::stream_sync(st));  // e.g. under a profiler
        synced = true;
        spins = 0;
      }
      sp::relax();
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  memcpy(out, h_out, cols * sizeof(fe_t));
  return SP_OK;
}
int sp_rowmat_vec_eq_begin(sp_ctx* c, const sp_table* poly, const uint64_t* r, size_t ell, size_t cols, sp_vec_job** out) {
  return sp_rowmat_vec_eq_begin_with(c, poly, r, ell, cols, nullptr, out);
}
int sp_rowmat_vec_eq_begin_with(sp_ctx* c, const sp_table* poly, const uint64_t* r, size_t ell, size_t cols, const uint64_t* addend, sp_vec_job** out) {
  if (!poly || !out || (!r && ell)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_rowmat_vec_eq_begin: null argument");
  if (ell > 20) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_rowmat_vec_eq_begin: more than 2^20 rows");
  const size_t rows = (size_t)1 << ell;
  if (rows * cols > poly->cap || cols == 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "bind_with_delayed: poly shorter than rows*cols");
  int rc_pin;
  if ((rc_pin = ensure_pinned_vec(c, cols))) return rc_pin;
  const size_t splits = rows < 64 ? rows : 64;
  fe_t* dL = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_L, rows * sizeof(fe_t), 1);
  fe_t* part = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_PART, splits * cols * sizeof(fe_t), 1);
  fe_t* dout = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_OUT, 2 * cols * sizeof(fe_t), 1);  // the product | the addend
  if (!dL || !part || !dout) return SP_ERR_NO_DEVICE;
  fe_t rr[20];
  for (size_t i = 0; i < ell; ++i) memcpy(&rr[i], r + 4 * i, 32);
  hipStream_t st = c->stream3;
  if (ell <= 10) {  // two half tables by value, the product on the device
    spk::EqTensorArgs a;
    const size_t hb = ell / 2, lb = ell - hb;
    eq_table_host(rr, hb, a.left);
    eq_table_host(rr + hb, lb, a.right);
    a.lo_bits = (int)lb;
    a.n = (unsigned)rows;
    hipLaunchKernelGGL(spk::k_eq_tensor<false>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, a, dL);
  } else {  // more rows than the argument block holds halves for: the table from the host
    std::vector<fe_t> w(rows);
    eq_table_host(rr, ell, w.data());
    SP_HIP(hipMemcpyAsync(dL, w.data(), rows * sizeof(fe_t), hipMemcpyHostToDevice, st));
    SP_HIP(sp::stream_sync(st));  // w is a local
  }
  c->timed_on(st, "rowmat_vec", 32ull * (rows * cols + rows + cols), [&] { launch_rowmat_vec(st, poly->d, rows, cols, dL, part, splits, dout); });
  SP_HIP(hipMemcpyAsync(c->h_pinned_vec, dout, cols * sizeof(fe_t), hipMemcpyDeviceToHost, st));
  SP_HIP(hipEventRecord(c->vec_ev, st));
  sp_vec_job* job = new sp_vec_job();
  job->cols = cols;
  job->d_out = dout;
  if (addend) {  // staged in the pinned block (the caller's buffer is free on return), uploaded behind the product: nowhere near anybody's critical path
    fe_t* stage = reinterpret_cast<fe_t*>(c->h_pinned_vec) + c->h_pinned_vec_cols;
    memcpy(stage, addend, cols * sizeof(fe_t));
    SP_HIP(hipMemcpyAsync(dout + cols, stage, cols * sizeof(fe_t), hipMemcpyHostToDevice, st));
    job->d_add = dout + cols;
    job->armed = scale_add_arm(c, st, dout, dout + cols, cols);  // waits on the stream, behind the product and the upload, for _finish_scaled's scale
  }
  *out = job;
  return SP_OK;
}
// The addend of a job begun with one is the IPA's mask d (ipa.rs:139-149): together with the public z = r LZ + d it gives LZ away, so neither its pinned staging
// copy nor its device copy outlives the job (the announced opening wipes its copies the same way). Called when nothing reads them any more.
static void vec_addend_wipe(sp_ctx* c, const fe_t* d_add, size_t cols, bool upload_may_be_running) {
  if (!d_add) return;
  if (upload_may_be_running) (void)sp::stream_sync(c->stream3);
  explicit_bzero(reinterpret_cast<fe_t*>(c->h_pinned_vec) + c->h_pinned_vec_cols, cols * sizeof(fe_t));
  (void)hipMemsetAsync(const_cast<fe_t*>(d_add), 0, cols * sizeof(fe_t), c->stream3);
}
int sp_rowmat_vec_eq_finish(sp_ctx* c, sp_vec_job* job, uint64_t* out) {
  if (!job || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_rowmat_vec_eq_finish: null argument");
  const size_t cols = job->cols;
  const fe_t* da = job->d_add;
  scale_add_abort(c, job->armed);  // (a job begun with an addend and finished without its scale: the waiting kernel leaves)
  delete job;
  SP_HIP(sp::event_sync(c->vec_ev));
  memcpy(out, c->h_pinned_vec, cols * sizeof(fe_t));
  vec_addend_wipe(c, da, cols, true);
  return SP_OK;
}
int sp_rowmat_vec_eq_finish_scaled(sp_ctx* c, sp_vec_job* job, const uint64_t scale[4], uint64_t* out) {
  if (!job || !out || !scale) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_rowmat_vec_eq_finish_scaled: null argument");
  const size_t cols = job->cols;
  const fe_t *dx = job->d_out, *da = job->d_add;
  const unsigned armed = job->armed;
  delete job;
  if (!da) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_rowmat_vec_eq_finish_scaled: the job was begun without an addend");
  fe_t sc;
  memcpy(&sc, scale, 32);
  // (unarmed: on the job's own stream, behind the product and the upload of the addend, which ended long ago)
  const int rc = armed ? scale_add_fire(c, c->stream3, armed, dx, da, sc, cols, out, "rowmat_vec_eq_finish_scaled")
                       : scale_add_to_host(c, c->stream3, dx, da, sc, cols, out, "rowmat_vec_eq_finish_scaled");
  vec_addend_wipe(c, da, cols, rc != SP_OK);  // (the sum has arrived: the kernel that read the addend is done, and so is its upload)
  return rc;
}
int sp_msm_job_finish(sp_ctx* c, sp_msm_job* job, uint64_t out_aff[8]) { return sp_msm_ck_finish(c, nullptr, job, nullptr, out_aff); }

// ---- FixedBaseMul tables over arbitrary points (msm.rs:653-689, 727-773) ---------------------------------------------------------------------------
struct sp_fbtables {
  size_t n = 0;
  aff_t* d_tables = nullptr;  // n x 32 x 255 affine multiples
  // sp_fbtables_create_async: the build is queued on the context's table stream; `ready_ev` is recorded behind it and `ready` remembers that it was seen
  int device = 0;
  hipEvent_t ready_ev = nullptr;
  mutable std::atomic<int> ready{1};
  aff_t* d_points = nullptr;        // the bases on the device (the build's input; freed with the tables)
  std::vector<aff_t> h_points;      // the source of their (possibly still running) upload
};
// the context's table stream (lowest priority: a build runs beside whatever the context proves) and its grow-only scratch
static int table_stream_scratch(sp_ctx* c, size_t bytes, char** scratch) {
  if (!c->stream_tab) {
    int lo = 0, hi = 0;
    SP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    SP_HIP(hipStreamCreateWithPriority(&c->stream_tab, hipStreamNonBlocking, lo));
  }
  if (bytes > c->tab_scratch_bytes) {
    SP_HIP(sp::stream_sync(c->stream_tab));  // a build still running reads the old scratch
    if (c->tab_scratch) hipFree(c->tab_scratch);
    c->tab_scratch = nullptr;
    c->tab_scratch_bytes = 0;
    SP_HIP(hipMalloc(&c->tab_scratch, bytes));
    c->tab_scratch_bytes = bytes;
  }
  *scratch = (char*)c->tab_scratch;
  return SP_OK;
}
static int build_window_tables(sp_ctx* c, const aff_t* host_points, size_t n, aff_t** out_tables) {
  const size_t per = 32 * 255;
  aff_t* tables = nullptr;
  if (fbtables_old_build()) {
    DevBuf pts, tj;
    int rc;
    if ((rc = pts.alloc(n * sizeof(aff_t))) || (rc = tj.alloc(n * per * sizeof(jac_t)))) return rc;
    SP_HIP(hipMalloc((void**)&tables, n * per * sizeof(aff_t)));
    hipError_t e = hipMemcpyAsync(pts.p, host_points, n * sizeof(aff_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      sp::launch_fixed_base_tables(c->stream, pts.as<aff_t>(), n, tj.as<jac_t>());
      sp::launch_jac_to_affine(c->stream, tj.as<jac_t>(), n * per, tables);
      e = sp::stream_sync(c->stream);
    }
    if (e != hipSuccess) {
      hipFree(tables);
      return fail(SP_ERR_NO_DEVICE, std::string("window tables: ") + hipGetErrorString(e));
    }
    *out_tables = tables;
    return SP_OK;
  }
  std::lock_guard<std::mutex> lk(c->tab_mu);
  char* scratch = nullptr;
  int rc = table_stream_scratch(c, sp::window_tables_scratch(n < sp::WT_CHUNK ? n : sp::WT_CHUNK), &scratch);
  if (rc) return rc;
  DevBuf pts;
  if ((rc = pts.alloc(n * sizeof(aff_t)))) return rc;
  SP_HIP(hipMalloc((void**)&tables, n * per * sizeof(aff_t)));
  hipError_t e = hipMemcpy(pts.p, host_points, n * sizeof(aff_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    sp::launch_window_tables(c->stream_tab, pts.as<aff_t>(), n, scratch, tables);
    e = sp::stream_sync(c->stream_tab);
  }
  if (e != hipSuccess) {
    hipFree(tables);
    return fail(SP_ERR_NO_DEVICE, std::string("window tables: ") + hipGetErrorString(e));
  }
  *out_tables = tables;
  return SP_OK;
}
int sp_fbtables_create(sp_ctx* c, const uint64_t* points_aff, size_t n, sp_fbtables** out) {
  if (n == 0 || n > 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_create: 1 .. 4096 points");
  int erc = sp::multi_mul_ensure(c, 1);
  if (erc) return erc;
  aff_t* tables = nullptr;
  if ((erc = build_window_tables(c, reinterpret_cast<const aff_t*>(points_aff), n, &tables))) return erc;
  sp_fbtables* t = new sp_fbtables();
  t->n = n;
  t->device = c->device;
  t->d_tables = tables;
  *out = t;
  return SP_OK;
}
// The same, returning as soon as the build is QUEUED (on the context's lowest-priority table stream, beside whatever else the context runs): the tables of a
// prepared witness's row commitments are first read at the end of the first prove on it, ~2 ms of device work that prep_prove need not wait for.
// sp_fbtables_ready tells (or waits); every entry point that takes tables waits by itself, so a caller that never asks stays correct.
int sp_fbtables_create_async(sp_ctx* c, const uint64_t* points_aff, size_t n, sp_fbtables** out) {
  if (fbtables_old_build()) return sp_fbtables_create(c, points_aff, n, out);
  if (n == 0 || n > 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_create_async: 1 .. 4096 points");
  int rc = sp::multi_mul_ensure(c, 1);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(c->tab_mu);
  char* scratch = nullptr;
  if ((rc = table_stream_scratch(c, sp::window_tables_scratch(n < sp::WT_CHUNK ? n : sp::WT_CHUNK), &scratch))) return rc;
  std::unique_ptr<sp_fbtables> t(new sp_fbtables());
  t->n = n;
  t->device = c->device;
  t->h_points.assign(reinterpret_cast<const aff_t*>(points_aff), reinterpret_cast<const aff_t*>(points_aff) + n);
  hipError_t e = hipMalloc((void**)&t->d_points, n * sizeof(aff_t));
  if (e == hipSuccess) e = hipMalloc((void**)&t->d_tables, n * 32 * 255 * sizeof(aff_t));
  if (e == hipSuccess) e = hipEventCreateWithFlags(&t->ready_ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMemcpyAsync(t->d_points, t->h_points.data(), n * sizeof(aff_t), hipMemcpyHostToDevice, c->stream_tab);
  if (e == hipSuccess) {
    sp::launch_window_tables(c->stream_tab, t->d_points, n, scratch, t->d_tables);
    e = hipEventRecord(t->ready_ev, c->stream_tab);
  }
  if (e != hipSuccess) {
    sp_fbtables_free(t.release());
    return fail(SP_ERR_NO_DEVICE, std::string("sp_fbtables_create_async: ") + hipGetErrorString(e));
  }
  t->ready.store(0, std::memory_order_release);
  *out = t.release();
  return SP_OK;
}
int sp_fbtables_ready(const sp_fbtables* t, int wait) {
  if (!t) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_ready: null tables");
  if (t->ready.load(std::memory_order_acquire)) return 1;
  hipError_t e = wait ? sp::event_sync(t->ready_ev) : hipEventQuery(t->ready_ev);
  if (e == hipErrorNotReady) return 0;
  if (e != hipSuccess) return fail(SP_ERR_NO_DEVICE, std::string("sp_fbtables_ready: ") + hipGetErrorString(e));
  t->ready.store(1, std::memory_order_release);
  return 1;
}
void sp_fbtables_free(sp_fbtables* t) {
  if (!t) return;
  if (t->ready_ev) {
    if (!t->ready.load(std::memory_order_acquire)) sp::event_sync(t->ready_ev);  // the build writes d_tables and reads d_points / the host copy until then
    hipEventDestroy(t->ready_ev);
  }
  if (t->d_tables) hipFree(t->d_tables);
  if (t->d_points) hipFree(t->d_points);
  delete t;
}
// test / diagnostic access: `count` table entries (affine points, 64 bytes each) starting at entry `first` of the n x 32 x 255 array
int sp_fbtables_read(const sp_fbtables* t, size_t first, size_t count, uint64_t* out) {
  if (!t || !out || first + count > t->n * 32 * 255) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_read: range outside the tables");
  int r = sp_fbtables_ready(t, 1);
  if (r < 0) return r;
  SP_HIP(hipMemcpy(out, t->d_tables + first, count * sizeof(aff_t), hipMemcpyDeviceToHost));
  return SP_OK;
}
}  // extern "C"
namespace sp {
// sum_i scalars[i] * point_i in ONE launch (kernels_msm.hpp k_multi_mul_coop): scalars and result through mapped pinned pages, no copies, no host-side
// tail. Two lanes per context, each with its own pages / ticket / sequence number: lane 1 on the auxiliary stream (callable from a helper thread
// beside the owner's calls on the main stream, like sp_msm_eq_begin), lane 0 on the main stream. The caller polls the self-validating result slot;
// a poll that runs long (profiler, debugger) falls back to a stream synchronise, after which the slot must be valid.
// GROUPS (k_multi_mul_coop): the blocks of a walk of >= MM_GROUP_MIN_BLOCKS blocks join in MM_GROUPS groups, each with a result slot of its own (128 bytes
// apart, behind the scalars of the lane's mapped page); multi_mul_collect adds the groups' sums on the host.
static const unsigned MM_GROUPS = 8, MM_GROUP_MIN_BLOCKS = 32;
static const size_t MM_GROUP_SLOTS_OFF = 256 + 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS * sizeof(fe_t);
static bool mm_groups_enabled() {
  static const bool on = [] {
    const char* e = getenv("SPARTAN_WALK_GROUPS");  // "0": one join on the device (A/B)
    return !(e && e[0] == '0');
  }();
  return on;
}
int multi_mul_ensure(sp_ctx* c, int lane) {
  const size_t page = MM_GROUP_SLOTS_OFF + 128 * MM_GROUPS;
  if (!c->h_mm[lane]) {
    SP_HIP(hipHostMalloc(&c->h_mm[lane], page, hipHostMallocMapped));
    memset(c->h_mm[lane], 0, page);
    SP_HIP(hipHostGetDevicePointer(&c->d_mm[lane], c->h_mm[lane], 0));
    SP_HIP(hipMalloc(&c->d_mm_work[lane], 256 + spk::MULTI_MUL_MAX_BLOCKS * sizeof(xyzz_t)));
    SP_HIP(hipMemsetAsync(c->d_mm_work[lane], 0, 256, c->stream2));  // the ticket; every launch leaves it at zero again
    SP_HIP(sp::stream_sync(c->stream2));
  }
  return SP_OK;
}
// `scalars`: n host scalars (copied into the lane's mapped page), or — `d_scalars` given — n - 1 scalars already in device memory followed by `last`
// `raw_blocks` > 0: `scalars` are that many 64-byte uniform blocks (scalar i = from_uniform of block i, reduced on the device; scalars beyond them are
// zero up to `last`): needs the wide kernel (n >= its threshold) and `last`
size_t multi_mul_wide_min() { return 1024; }  // scalars from which the 1024-item blocks (k_multi_mul_wide) are used
int multi_mul_launch(sp_ctx* c, int lane, const aff_t* d_tables, const uint64_t* scalars, size_t n, unsigned* seq_out, const fe_t* d_scalars, const fe_t* last,
                     size_t raw_blocks, bool expand) {
  if (expand && (raw_blocks || d_scalars || !last || !scalars || n >= multi_mul_wide_min())) return fail(SP_ERR_INTERNAL, "multi_mul: the expanding form takes host scalars below the wide size");
  if (n == 0 || n > 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multi_mul: 1 .. 4096 scalars");
  if (raw_blocks && n < multi_mul_wide_min()) return fail(SP_ERR_INTERNAL, "multi_mul: raw blocks need the wide kernel");
  int erc = multi_mul_ensure(c, lane);
  if (erc) return erc;
  if (raw_blocks) {
    if (!last || raw_blocks > n - 1 || 64 * (n - 1) > 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS * sizeof(fe_t)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multi_mul: raw blocks do not fit");
    memcpy((char*)c->h_mm[lane] + 256, scalars, 64 * raw_blocks);
    if (raw_blocks < n - 1) memset((char*)c->h_mm[lane] + 256 + 64 * raw_blocks, 0, 64 * (n - 1 - raw_blocks));  // from_uniform(0) == 0
    c->mm_host_bytes[lane] = 64 * raw_blocks;
  } else if (expand) {
    memcpy((char*)c->h_mm[lane] + 256, scalars, (n / 2 + 2) * sizeof(fe_t));
    c->mm_host_bytes[lane] = (n / 2 + 2) * sizeof(fe_t);
  } else if (!d_scalars) {
    memcpy((char*)c->h_mm[lane] + 256, scalars, n * sizeof(fe_t));
    c->mm_host_bytes[lane] = n * sizeof(fe_t);
  }
  if (++c->mm_seq[lane] == 0) ++c->mm_seq[lane];
  const unsigned seq = c->mm_seq[lane];
  hipStream_t st = lane ? c->stream2 : c->stream;
  const fe_t* src = d_scalars ? d_scalars : reinterpret_cast<const fe_t*>((char*)c->d_mm[lane] + 256);
  const fe_t lastv = last ? *last : fe_zero();
  const size_t wide_min = multi_mul_wide_min();
  const unsigned nblk = (unsigned)((n + 3) / 4);
  const unsigned gblocks = (n < wide_min && nblk >= MM_GROUP_MIN_BLOCKS && mm_groups_enabled()) ? (nblk + MM_GROUPS - 1) / MM_GROUPS : 0u;
  c->timed_on(st, "multi_mul", 32ull * n, [&] {
    if (n >= wide_min)
      hipLaunchKernelGGL(spk::k_multi_mul_wide, dim3((unsigned)((n * 32 + spk::MULTI_MUL_WIDE_ITEMS - 1) / spk::MULTI_MUL_WIDE_ITEMS)), dim3(512), 0, st, src, n, d_tables,
                         reinterpret_cast<xyzz_t*>((char*)c->d_mm_work[lane] + 256), reinterpret_cast<unsigned*>(c->d_mm_work[lane]), reinterpret_cast<unsigned*>(c->d_mm[lane]), seq,
                         lastv, last ? 1 : 0, raw_blocks ? 1 : 0);
    else
      hipLaunchKernelGGL(spk::k_multi_mul_coop, dim3(nblk), dim3(512), 0, st, src, n, d_tables, reinterpret_cast<xyzz_t*>((char*)c->d_mm_work[lane] + 256),
                         reinterpret_cast<unsigned*>(c->d_mm_work[lane]),
                         reinterpret_cast<unsigned*>((char*)c->d_mm[lane] + (gblocks ? MM_GROUP_SLOTS_OFF : 0)), seq, lastv, expand ? 2 : (last ? 1 : 0), gblocks);
  });
  c->mm_groups[lane] = gblocks ? (nblk + gblocks - 1) / gblocks : 0;
  if (seq_out) *seq_out = seq;
  return SP_OK;
}
int multi_mul_collect(sp_ctx* c, int lane, unsigned seq, jac_t* out, bool yield) {
  const unsigned groups = c->mm_groups[lane];
  if (groups) {  // one slot per group, added here (the top of the tree: ~0.5 us an addition on the host against ~4.8 us a level on the device)
    jac_t acc = jac_identity();
    bool done[MM_GROUPS] = {};
    unsigned remaining = groups;
    bool synced = false;
    for (long spins = 0; remaining; ++spins) {
      for (unsigned g = 0; g < groups; ++g) __builtin_prefetch((const char*)c->h_mm[lane] + MM_GROUP_SLOTS_OFF + 128 * g, 0, 3);
      for (unsigned g = 0; g < groups; ++g) {
        if (done[g]) continue;
        volatile const unsigned* slot = reinterpret_cast<volatile const unsigned*>((const char*)c->h_mm[lane] + MM_GROUP_SLOTS_OFF + 128 * g);
        if (slot[24] != seq) continue;
        std::atomic_thread_fence(std::memory_order_acquire);
        unsigned w[24], a = seq, b = seq * spk::MULTI_MUL_SLOT_K;
        for (int i = 0; i < 24; ++i) {
          w[i] = slot[i];
          a += w[i];
          b += (unsigned)(i + 1) * w[i];
        }
        if (!(slot[24] == seq && slot[25] == a && slot[26] == b && slot[27] == seq)) continue;
        jac_t p;
        memcpy(&p, w, sizeof(jac_t));
        acc = jac_add(acc, p);
        done[g] = true;
        --remaining;
      }
      if (!remaining) break;
      if (spins > 400000) {
        sp::slow_note("multi_mul_collect (groups)", spins);
        if (synced) return fail(SP_ERR_INTERNAL, "multi_mul: the kernel did not deliver its result slots");
        SP_HIP(sp::stream_sync(lane ? c->stream2 : c->stream));  // e.g. under a profiler
        synced = true;
        spins = 0;
      }
      if (yield && spins > 20000) std::this_thread::sleep_for(std::chrono::microseconds(20));
      else sp::relax();
    }
    *out = acc;
    if (c->mm_host_bytes[lane]) {
      explicit_bzero((char*)c->h_mm[lane] + 256, c->mm_host_bytes[lane]);
      c->mm_host_bytes[lane] = 0;
    }
    return SP_OK;
  }
  volatile const unsigned* slot = reinterpret_cast<volatile const unsigned*>(c->h_mm[lane]);
  unsigned w[24];
  bool synced = false;
  for (long spins = 0;; ++spins) {
    if (slot[24] == seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      unsigned a = seq, b = seq * spk::MULTI_MUL_SLOT_K;
      for (int i = 0; i < 24; ++i) {
        w[i] = slot[i];
        a += w[i];
        b += (unsigned)(i + 1) * w[i];
      }
      if (slot[24] == seq && slot[25] == a && slot[26] == b && slot[27] == seq) break;
    }
    if (spins > 400000) {
      sp::slow_note("multi_mul_collect", spins);
      if (synced) return fail(SP_ERR_INTERNAL, "multi_mul: the kernel did not deliver its result slot");
      SP_HIP(sp::stream_sync(lane ? c->stream2 : c->stream));  // e.g. under a profiler
      synced = true;
      spins = 0;
    }
    // a helper-thread caller: the kernel takes ~130 us; past that it is late because the chip is shared, and the poll yields its CPU
    if (yield && spins > 20000) std::this_thread::sleep_for(std::chrono::microseconds(20));
    else sp::relax();
  }
  memcpy(out, w, sizeof(jac_t));
  if (c->mm_host_bytes[lane]) {  // the kernel has finished with the scalars (randomness blocks, blinds) it read from the mapped page
    explicit_bzero((char*)c->h_mm[lane] + 256, c->mm_host_bytes[lane]);
    c->mm_host_bytes[lane] = 0;
  }
  return SP_OK;
}
// window tables of the whole key (num_cols bases, then h), built on first use: every MSM over the key — the IPA mask commitment delta, comm_LZ —
// becomes one table walk (1.07 GB for the 2048-wide key; 288 GB of HBM make the trade free)
int ck_key_tables(sp_ctx* c, const sp_ck* ck) {
  {
    const char* e = getenv("SPARTAN_KEY_TABLES");  // "0": keep the bucket MSMs (A/B runs, tests of the fallback)
    if (e && e[0] == '0') return 1;
  }
  for (;;) {  // claim the build, find it done, or wait for the builder outside the lock (group_common.hpp)
    {
      std::lock_guard<std::mutex> lk(ck->lazy_mu);
      if (ck->d_keytables) return SP_OK;
      if (ck->keytables_failed) return 1;
      if (ck->num_cols + 1 > 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS) return 1;
      if (!ck->keytables_building) {
        ck->keytables_building = true;
        break;
      }
    }
    relax();
  }
  aff_t* tables = nullptr;
  int rc = SP_OK;
  {
    std::vector<aff_t> pts(ck->num_cols + 1);
    hipError_t e = hipMemcpy(pts.data(), ck->d_bases, ck->num_cols * sizeof(aff_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      rc = fail(SP_ERR_NO_DEVICE, std::string("key tables: ") + hipGetErrorString(e));
    } else {
      pts[ck->num_cols] = ck->h;
      if (build_window_tables(c, pts.data(), pts.size(), &tables)) rc = 1;  // out of memory: the bucket MSM stays
    }
  }
  std::lock_guard<std::mutex> lk(ck->lazy_mu);
  ck->keytables_building = false;
  if (rc == 1) ck->keytables_failed = true;
  if (rc) return rc;
  ck->d_keytables = tables;
  return SP_OK;
}
}  // namespace sp
extern "C" {
// _begin launches on the auxiliary stream, _finish polls (one multiplication in flight per context and lane).
int sp_fbtables_multi_mul_begin(sp_ctx* c, const sp_fbtables* t, const uint64_t* scalars, size_t n) {
  if (n != t->n) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_multi_mul: one scalar per table");
  if (int r = sp_fbtables_ready(t, 1); r < 0) return r;  // (tables of sp_fbtables_create_async whose build has not ended: waited for here)
  return sp::multi_mul_launch(c, 1, t->d_tables, scalars, n, nullptr, nullptr, nullptr, 0);
}
// the same walk with its scalars one level short of eq(r, .): P = eq(r_1 .. r_(k-1), .) (ceil(nfixed / 2) entries), S0 | S1 (h's scalar is S0 + r_k (S1 - S0))
// and r_k - the kernel forms the last level itself (k_multi_mul_coop EXPAND). n = nfixed + 1 tables; below the wide kernel's size only.
int sp_fbtables_multi_mul_begin_eq(sp_ctx* c, const sp_fbtables* t, const uint64_t* P, size_t nfixed, const uint64_t S01[8], const uint64_t r_last[4]) {
  if (!t || !P || !S01 || !r_last) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_multi_mul_begin_eq: null argument");
  const size_t n = nfixed + 1, np = n / 2;
  if (n != t->n || nfixed == 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_multi_mul_begin_eq: one table per fixed row and one of h");
  if (n >= sp::multi_mul_wide_min()) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_fbtables_multi_mul_begin_eq: at most 1022 fixed rows");
  if (int r = sp_fbtables_ready(t, 1); r < 0) return r;
  std::vector<fe_t> buf(np + 2);
  memcpy(buf.data(), P, np * sizeof(fe_t));
  memcpy(buf.data() + np, S01, 2 * sizeof(fe_t));
  fe_t rl;
  memcpy(&rl, r_last, 32);
  return sp::multi_mul_launch(c, 1, t->d_tables, reinterpret_cast<const uint64_t*>(buf.data()), n, nullptr, nullptr, &rl, 0, true);
}
int sp_fbtables_multi_mul_finish(sp_ctx* c, uint64_t out_aff[8]) {
  if (!c->h_mm[1] || c->mm_seq[1] == 0) return fail(SP_ERR_INTERNAL, "sp_fbtables_multi_mul_finish: nothing in flight");
  jac_t sum;
  int rc = sp::multi_mul_collect(c, 1, c->mm_seq[1], &sum, true);
  if (rc) return rc;
  store_aff(out_aff, jac_to_affine(sum));
  return SP_OK;
}
int sp_fbtables_multi_mul(sp_ctx* c, const sp_fbtables* t, const uint64_t* scalars, size_t n, uint64_t out_aff[8]) {
  int rc = sp_fbtables_multi_mul_begin(c, t, scalars, n);
  return rc ? rc : sp_fbtables_multi_mul_finish(c, out_aff);
}

// ---- HyraxPCS::prove (src/provider/pcs/hyrax_pc.rs:387-478) + InnerProductArgumentLinear::prove (src/provider/pcs/ipa.rs:125-170) ----------------------
// ONE call in the place where the reference's trait method sits, so that a caller that follows src/spartan.rs:423-437 statement by statement gets the
// overlap below the ABI that the C++ driver used to arrange above it:
//   * the 64-byte-per-row transcript encoding of `comm` and the Keccak blocks of absorb(b"poly_com", comm) run on the context's helper thread,
//   * delta = <d, ck> + r_delta h (ipa.rs:147) and comm_LZ = <LZ, ck> + r_LZ h (hyrax_pc.rs:454-455) are table walks over the window tables of the
//     whole key (ck_key_tables: one launch each, no digits / sort / buckets / host Horner) on two streams side by side; LZ = L^T poly
//     (bind_with_delayed, :38-54) stays in device memory in front of its walk and travels to the host only for z_vec,
//   * the host meanwhile draws d_vec from the randomness stream (the wide reductions of E::Scalar::random), forms r_LZ = <L, blind>, <R, d> (tensor
//     form: n + sqrt(n) products) and beta.
// Transcript order and every value are the reference's. out = delta (8) | beta (8) | z_vec (4 * cols) | z_delta (4) | z_beta (4) words.
static void hp_point_bytes(const aff_t& a, uint8_t out[64]) {  // x BE || y BE (src/provider/traits.rs:288-305)
  sp::fe_to_be_bytes<B>(a.x, out);
  sp::fe_to_be_bytes<B>(a.y, out + 32);
}
}  // extern "C"
// ---- PCS::prove announced ahead ----------------------------------------------------------------------------------------------------------------
// src/spartan.rs calls PCS::prove last (:425-435), but three of its inputs exist long before: the commitment and its blinds when
// r1cs_instance_and_witness returns (:238-245), the IPA's randomness whenever the caller draws it (the reference draws inside
// InnerProductArgumentLinear::prove, ipa.rs:139-149: independent of everything), and the ROW half of the evaluation point when the inner sum-check has
// drawn it — rounds before its end. A caller (the shim's r1cs_instance_and_witness wrapper; prove_reference_order here) that announces the opening
// lets the library start under the sum-checks what sp_hyrax_prove would otherwise start behind them: the commitment's transcript encoding and its
// Keccak blocks + the mask vector's wide reductions (helper thread), delta's table walk (auxiliary stream, at once), and — when sp_sumcheck_quad on the
// same context reports the row challenges — L^T W and comm_LZ's walk. sp_hyrax_prove checks that what it is given is what was announced and then only
// collects; anything that does not match is dropped and computed as before. No protocol value changes: same group elements, same transcript bytes.
struct sp_pcs_ahead {
  const sp_ck* ck = nullptr;
  const sp_table* poly = nullptr;
  size_t n = 0, npt = 0, nvr = 0, cols = 0, num_rows = 0;
  std::vector<aff_t> comm;
  std::vector<fe_t> blind, dvec, row_pt;
  std::vector<uint8_t> rng;
  fe_t r_delta;
  sp::Keccak256State hashed;  // "poly_com" || commitment bytes hashed into a fresh sponge (valid when the transcript is fresh at the absorb: checked)
  bool worker_busy = false, delta_launched = false, delta_collected = false;
  // written by the helper thread's jobs (the launches behind the last row challenge, the announcement's own job) and read by the sum-check's thread between
  // its rounds without waiting for the worker: atomics, sequentially consistent - these are a handful of accesses per prove
  std::atomic<bool> lz_launched{false}, failed{false};
  // <R, d> (ipa.rs:148) with R = eq(point[nvr..]) = left (x) right: T[b] = sum_a left[a] d[a * nright + b] once the first half of R's variables is drawn
  // col_pt collects the column challenges as they arrive; col_pt_T is the snapshot T was built from (frozen when the job is submitted): sp_hyrax_prove
  // compares ITS point with col_pt_T, so a later sum-check of the same length on this context — which overwrites col_pt — cannot make a stale T pass
  std::vector<fe_t> col_pt, col_pt_T, T;
  size_t hb = 0;
  bool T_submitted = false, T_ready = false;
  unsigned seq_delta = 0, seq_lz = 0;
  jac_t delta_j;
  fe_t r_LZ;
  // r_LZ = <eq(r_rows, .), blinds> is the blinds' multilinear extension at r_rows: the vector is folded by its top variable with every row challenge
  // (2^(nvr-k) products behind challenge k, under the device's next round), so that the last challenge leaves one product (hyrax_pc.rs:446-455)
  std::vector<fe_t> bfold;
  size_t rows_folded = 0;
  // With FixedBaseMul tables of the first `nfixed` commitment rows and of h (sp_hyrax_prove_announce_tables; the other rows are blind_i h by the caller's
  // word) comm_LZ = sum_{i < nfixed} L_i comm[i] + (sum_{i >= nfixed} L_i blind_i) h is ONE walk over those tables that needs L alone, not L^T W: it goes
  // out right behind the last row challenge instead of behind the 17-50 us matrix-vector product. eq(r_rows, .) is expanded a level per challenge
  // (P, new variable = index LSB, eq.rs:66-76); zfold = the zero rows' blinds folded like bfold gives h's scalar.
  const aff_t* row_tables = nullptr;
  size_t nfixed = 0;
  std::vector<fe_t> P, zfold, zfold_prev;
  bool lz_submitted = false;  // the launches behind the last row challenge are a job of the helper thread (the sum-check's thread goes on with its rounds)
  // z_vec = r LZ + d on the device (ipa.rs:160-163): the mask vector is uploaded behind delta's walk, the scaled sum lands in mapped memory
  fe_t* d_out = nullptr;      // [LZ (num_cols) | - | d (cols)] in the auxiliary lane's WS_ROWMAT_OUT
  std::atomic<bool> dvec_uploaded{false};
  std::atomic<unsigned> z_armed{0};      // sequence number of the k_scale_add_wait queued behind L^T W (0: none), released by sp_hyrax_prove's challenge or aborted
  bool z_fired = false;
};
namespace sp {
static void pcs_ahead_drain(sp_ctx* c) {  // no device job of a dropped announcement may outlive it (its mapped result slot is reused)
  sp_pcs_ahead* S = c->pcs_ahead;
  if (!S) return;
  if (S->worker_busy && c->pcs_worker) c->pcs_worker->wait();
  if (S->z_armed && !S->z_fired) {
    scale_add_abort(c, S->z_armed);
    S->z_fired = true;
  }
  jac_t sink;
  if (S->delta_launched && !S->delta_collected) (void)multi_mul_collect(c, 1, S->seq_delta, &sink, false);
  if (S->lz_launched) {
    (void)multi_mul_collect(c, 1, S->seq_lz, &sink, false);
    if (c->pcs_ev) (void)event_sync(c->pcs_ev);
  }
}
bool pcs_ahead_wants(const sp_ctx* c, size_t rounds) { return c->pcs_ahead && !c->pcs_ahead->failed && rounds == c->pcs_ahead->npt + 1; }
// The announcement holds zero-knowledge material for the span of both sum-checks (the commitment's blinds, the IPA's randomness blocks and the mask
// vector drawn from them, the partial sums of <R, d>, r_delta, r_LZ): wiped before the memory goes back to the allocator.
template <class T>
static void wipe_vec(std::vector<T>& v) {
  if (!v.empty()) explicit_bzero(v.data(), v.size() * sizeof(T));
}
void pcs_ahead_free(sp_ctx* c) {
  if (!c || !c->pcs_ahead) return;
  pcs_ahead_drain(c);
  sp_pcs_ahead* S = c->pcs_ahead;
  c->pcs_ahead = nullptr;
  if (S->dvec_uploaded && S->d_out && c->h_pinned_vec) {
    // a dropped or retracted announcement had staged the mask vector (pinned block) and uploaded it (auxiliary stream): neither copy outlives it.
    // (a consumed one has wiped both already - sp_hyrax_prove - and wiping zeros again costs a 64 KiB memset off everybody's path)
    (void)sp::stream_sync(c->stream2);
    explicit_bzero(reinterpret_cast<fe_t*>(c->h_pinned_vec) + c->h_pinned_vec_cols, S->cols * sizeof(fe_t));
    (void)hipMemsetAsync(S->d_out + S->ck->num_cols + 1, 0, S->cols * sizeof(fe_t), c->stream2);
  }
  auto wipe = [S] {
    wipe_vec(S->blind);
    wipe_vec(S->dvec);
    wipe_vec(S->rng);
    wipe_vec(S->T);
    wipe_vec(S->bfold);
    wipe_vec(S->zfold);
    wipe_vec(S->zfold_prev);
    explicit_bzero(&S->r_delta, sizeof(fe_t));
    explicit_bzero(&S->r_LZ, sizeof(fe_t));
    delete S;
  };
  // ~250 KB at config 2: 15-20 us of stores, on the context's helper thread when there is one (nothing else of the announcement is alive: drained above)
  if (c->pcs_worker) c->pcs_worker->submit(wipe);
  else wipe();
}
// the inner sum-check's challenges: round 0 binds the variable that separates W from (1, X); rounds 1 .. nvr are the opening's row variables
void pcs_ahead_on_challenge(void* ctx, size_t round, const uint64_t r[4]) {
  sp_ctx* c = (sp_ctx*)ctx;
  sp_pcs_ahead* S = c->pcs_ahead;
  if (!S || S->failed || S->nvr == 0 || round == 0) return;
  if (round > S->nvr) {  // a column variable: after the first half of them the helper thread forms the partial sums of <R, d>
    const size_t k = round - S->nvr - 1;
    if (k < S->hb) memcpy(&S->col_pt[k], r, 32);
    if (k + 1 == S->hb && !S->T_submitted) {
      if (S->worker_busy) c->pcs_worker->wait();
      S->worker_busy = true;
      S->T_submitted = true;
      S->col_pt_T = S->col_pt;
      c->pcs_worker->submit([S] {
        const size_t nleft = (size_t)1 << S->hb, nright = S->cols >> S->hb;
        std::vector<fe_t> left(nleft);
        eq_table_host(S->col_pt_T.data(), S->hb, left.data());
        S->T.assign(nright, fe_zero());
        for (size_t a = 0; a < nleft; ++a)
          for (size_t b = 0; b < nright; ++b) S->T[b] = fe_add<spk::SF>(S->T[b], fe_mul<spk::SF>(left[a], S->dvec[a * nright + b]));
        S->T_ready = true;
      });
    }
    return;
  }
  if (S->lz_submitted || S->lz_launched) return;  // (lz_submitted first: while it is false no helper job is writing lz_launched)
  if (round != S->rows_folded + 1) {  // (challenges arrive in order, once each; anything else and the plain call computes everything itself)
    S->failed = true;
    return;
  }
  memcpy(&S->row_pt[round - 1], r, 32);
  if (round == 1 && c->pcs_worker) c->pcs_worker->keep_hot(700);  // the jobs below and the opening's own are claimed without a wake-up
  {
    const fe_t rk = S->row_pt[round - 1];
    const size_t h = S->bfold.size() / 2;
    for (size_t j = 0; j < h; ++j) S->bfold[j] = fe_add<spk::SF>(S->bfold[j], fe_mul<spk::SF>(rk, fe_sub<spk::SF>(S->bfold[j + h], S->bfold[j])));
    explicit_bzero(S->bfold.data() + h, h * sizeof(fe_t));
    S->bfold.resize(h);
    S->rows_folded = round;
    if (S->row_tables) {
      if (h == 1) S->zfold_prev = S->zfold;  // (S0, S1): h's scalar is S0 + r (S1 - S0), which the walk's kernel can form itself
      for (size_t j = 0; j < h; ++j) S->zfold[j] = fe_add<spk::SF>(S->zfold[j], fe_mul<spk::SF>(rk, fe_sub<spk::SF>(S->zfold[j + h], S->zfold[j])));
      explicit_bzero(S->zfold.data() + h, h * sizeof(fe_t));
      S->zfold.resize(h);
      if (round < S->nvr) {  // (the last level is the helper job's: 2^(nvr-1) products)
        std::vector<fe_t> Q(2 * S->P.size());
        for (size_t i = 0; i < S->P.size(); ++i) {
          const fe_t hi = fe_mul<spk::SF>(S->P[i], rk);
          Q[2 * i + 1] = hi;
          Q[2 * i] = fe_sub<spk::SF>(S->P[i], hi);
        }
        S->P.swap(Q);
      }
    }
  }
  if (round != S->nvr) return;
  S->r_LZ = S->bfold[0];
  if (S->worker_busy) {  // the announcement's helper job (it launched delta's walk on this lane)
    c->pcs_worker->wait();
    S->worker_busy = false;
  }
  if (S->failed || S->nvr > 10) {
    S->failed = true;
    return;
  }
  // Behind the last row challenge: delta's walk has had the whole outer sum-check and is collected (its lane's result slot is needed again), then L^T W and
  // comm_LZ's walk go out on the same lane - half a dozen runtime calls, ~20 us, on the helper thread: this thread's next round is not held up by them.
  S->lz_submitted = true;
  S->worker_busy = true;
  c->pcs_worker->submit([S, c] {
    if (hipSetDevice(c->device) != hipSuccess) {
      S->failed = true;
      return;
    }
    if (S->delta_launched && !S->delta_collected) {
      if (multi_mul_collect(c, 1, S->seq_delta, &S->delta_j, false)) {
        S->failed = true;
        return;
      }
      S->delta_collected = true;
    }
    const size_t num_rows = S->num_rows, cols = S->cols, num_cols = S->ck->num_cols, nvr = S->nvr;
    const size_t splits = num_rows < 64 ? num_rows : 64;
    fe_t* dL = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_L, num_rows * sizeof(fe_t), 1);
    fe_t* part = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_PART, splits * cols * sizeof(fe_t), 1);
    fe_t* dout = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_OUT, (num_cols + 1 + cols) * sizeof(fe_t), 1);
    if (!dL || !part || !dout || dout != S->d_out) {  // (sized by the announcement: a buffer that moved since would have lost the uploaded mask)
      S->failed = true;
      return;
    }
    hipStream_t st = c->stream2;
    bool walk_out = false;
    if (S->row_tables) {  // comm_LZ from the rows' own tables: the walk first, L^T W (needed for z_vec only) behind it
      const fe_t rl = S->row_pt[nvr - 1];
      int mrc;
      if (S->nfixed + 1 < multi_mul_wide_min() && S->zfold_prev.size() == 2) {  // the last level of eq(r_rows, .) on the device (k_multi_mul_coop EXPAND)
        const size_t np = (S->nfixed + 1) / 2;
        std::vector<fe_t> buf(np + 2);
        memcpy(buf.data(), S->P.data(), np * sizeof(fe_t));
        buf[np] = S->zfold_prev[0];
        buf[np + 1] = S->zfold_prev[1];
        mrc = multi_mul_launch(c, 1, S->row_tables, reinterpret_cast<const uint64_t*>(buf.data()), S->nfixed + 1, &S->seq_lz, nullptr, &rl, 0, true);
        wipe_vec(buf);
      } else {
        std::vector<fe_t> sc(S->nfixed + 1);
        for (size_t h2 = 0; h2 < S->P.size(); ++h2) {
          const fe_t hi = fe_mul<spk::SF>(S->P[h2], rl), lo = fe_sub<spk::SF>(S->P[h2], hi);
          if (2 * h2 < S->nfixed) sc[2 * h2] = lo;
          if (2 * h2 + 1 < S->nfixed) sc[2 * h2 + 1] = hi;
        }
        sc[S->nfixed] = S->zfold[0];
        mrc = multi_mul_launch(c, 1, S->row_tables, reinterpret_cast<const uint64_t*>(sc.data()), S->nfixed + 1, &S->seq_lz, nullptr, nullptr, 0);
        wipe_vec(sc);
      }
      if (mrc) {
        S->failed = true;
        return;
      }
      walk_out = true;
    }
    spk::EqTensorArgs a;
    const size_t hb = nvr / 2, lb = nvr - hb;
    eq_table_host(S->row_pt.data(), hb, a.left);
    eq_table_host(S->row_pt.data() + hb, lb, a.right);
    a.lo_bits = (int)lb;
    a.n = (unsigned)num_rows;
    hipLaunchKernelGGL(spk::k_eq_tensor<false>, dim3((unsigned)((num_rows + 255) / 256)), dim3(256), 0, st, a, dL);
    if (cols < num_cols && hipMemsetAsync(dout + cols, 0, (num_cols - cols) * sizeof(fe_t), st) != hipSuccess) {
      S->failed = true;
      S->lz_launched = walk_out;  // (still drained)
      return;
    }
    c->timed_on(st, "rowmat_vec", 32ull * (num_rows * cols + num_rows + cols), [&] { launch_rowmat_vec(st, S->poly->d, num_rows, cols, dL, part, splits, dout); });
    if ((!walk_out && multi_mul_launch(c, 1, S->ck->d_keytables, nullptr, num_cols + 1, &S->seq_lz, dout, &S->r_LZ, 0)) || hipEventRecord(c->pcs_ev, st) != hipSuccess) {
      S->failed = true;  // (a launched walk is still drained by pcs_ahead_drain: lz_launched stays false only if the launch itself failed)
      S->lz_launched = walk_out;
      return;
    }
    // z_vec's kernel waits for the IPA's challenge on the second auxiliary stream, behind this one's work so far. That stream is non-blocking: a waiting
    // kernel on a blocking stream would hold up every legacy default-stream operation of the process (a synchronous hipMemcpy between the sum-check and
    // the opening) until its watchdog.
    if (S->dvec_uploaded && c->stream3 && hipStreamWaitEvent(c->stream3, c->pcs_ev, 0) == hipSuccess)
      S->z_armed = scale_add_arm(c, c->stream3, dout, dout + num_cols + 1, cols);
    S->lz_launched = true;
  });
}
}  // namespace sp

extern "C" int sp_hyrax_prove_retract(sp_ctx* c) {
  if (!c) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_hyrax_prove_retract: null context");
  sp::pcs_ahead_free(c);
  return SP_OK;
}
static int announce_impl(sp_ctx* c, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n, const uint64_t* blinds, const uint8_t* rng,
                         size_t rng_blocks, const sp_fbtables* row_tables, size_t nfixed);
extern "C" int sp_hyrax_prove_announce(sp_ctx* c, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n, const uint64_t* blinds,
                                       const uint8_t* rng, size_t rng_blocks) {
  return announce_impl(c, ck, comm_rows_aff, rows, poly, n, blinds, rng, rng_blocks, nullptr, 0);
}
extern "C" int sp_hyrax_prove_announce_tables(sp_ctx* c, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n, const uint64_t* blinds,
                                              const uint8_t* rng, size_t rng_blocks, const sp_fbtables* row_tables, size_t nfixed) {
  if (!row_tables || row_tables->n != nfixed + 1 || nfixed > rows)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_hyrax_prove_announce_tables: one table per fixed row and one of h");
  return announce_impl(c, ck, comm_rows_aff, rows, poly, n, blinds, rng, rng_blocks, row_tables, nfixed);
}
static int announce_impl(sp_ctx* c, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n, const uint64_t* blinds, const uint8_t* rng,
                         size_t rng_blocks, const sp_fbtables* row_tables, size_t nfixed) {
  if (!c || !ck || !comm_rows_aff || !poly || !blinds || !rng) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_hyrax_prove_announce: null argument");
  sp::pcs_ahead_free(c);  // at most one announcement per context; an unconsumed one is dropped
  size_t npt = 0;
  while (((size_t)1 << npt) < n) ++npt;
  const size_t num_cols = ck->num_cols, num_rows = (n + num_cols - 1) / num_cols;
  if (n != ((size_t)1 << npt) || n > poly->cap || rows != num_rows || (num_rows & (num_rows - 1)) || num_rows < 2) return SP_OK;  // nothing to start ahead: the plain call decides
  size_t nvr = 0;
  while (((size_t)1 << nvr) < num_rows) ++nvr;
  const size_t cols = n / num_rows;
  if (rng_blocks < cols + 2 || nvr > 10) return SP_OK;
  SP_HIP(hipSetDevice(c->device));
  if (sp::ck_key_tables(c, ck) != 0) return SP_OK;  // no window tables of the key: the bucket MSMs have nothing to gain from an early start of this kind
  const bool delta_raw = num_cols + 1 >= sp::multi_mul_wide_min() && 64 * num_cols <= 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS * sizeof(fe_t);
  if (!delta_raw) return SP_OK;
  if (!c->pcs_ev) SP_HIP(hipEventCreateWithFlags(&c->pcs_ev, hipEventDisableTiming));
  if (c->h_pcs_bytes < cols * sizeof(fe_t)) {
    if (c->h_pcs) hipHostFree(c->h_pcs);
    c->h_pcs = nullptr;
    c->h_pcs_bytes = 0;
    SP_HIP(hipHostMalloc(&c->h_pcs, cols * sizeof(fe_t)));
    c->h_pcs_bytes = cols * sizeof(fe_t);
  }
  sp_pcs_ahead* S = new sp_pcs_ahead();
  S->ck = ck;
  S->poly = poly;
  S->n = n, S->npt = npt, S->nvr = nvr, S->cols = cols, S->num_rows = num_rows;
  S->comm.assign(reinterpret_cast<const aff_t*>(comm_rows_aff), reinterpret_cast<const aff_t*>(comm_rows_aff) + rows);
  S->blind.assign(reinterpret_cast<const fe_t*>(blinds), reinterpret_cast<const fe_t*>(blinds) + rows);
  S->rng.assign(rng, rng + 64 * (cols + 2));
  S->row_pt.resize(nvr);
  S->hb = (npt - nvr) / 2;
  S->col_pt.resize(S->hb ? S->hb : 1);
  S->dvec.resize(cols);
  S->r_delta = fe_from_uniform<spk::SF>(S->rng.data() + 64 * cols);
  S->bfold = S->blind;
  // (tables still being built - sp_fbtables_create_async a moment ago - are not waited for: this opening takes the walk over the key instead)
  if (row_tables && nfixed >= 1 && nfixed + 1 <= 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS && sp_fbtables_ready(row_tables, 0) == 1) {
    S->row_tables = row_tables->d_tables;
    S->nfixed = nfixed;
    S->zfold = S->blind;
    for (size_t i = 0; i < nfixed; ++i) S->zfold[i] = fe_zero();
    S->P.assign(1, fe_one<spk::SF>());
  }
  {  // the buffers of the launches behind the last row challenge and of z_vec, sized here: the helper thread's workspace() calls are then plain look-ups
    const size_t splits = num_rows < 64 ? num_rows : 64;
    int rc_pin = ensure_pinned_vec(c, cols);
    fe_t* dL = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_L, num_rows * sizeof(fe_t), 1);
    fe_t* part = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_PART, splits * cols * sizeof(fe_t), 1);
    S->d_out = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_OUT, (num_cols + 1 + cols) * sizeof(fe_t), 1);
    if (rc_pin || !dL || !part || !S->d_out) {
      delete S;
      return SP_OK;  // the plain call will report what is wrong
    }
  }
  c->pcs_ahead = S;
  // helper thread: the commitment's transcript bytes into a fresh sponge, then the mask vector's wide reductions
  if (!c->pcs_worker) c->pcs_worker = new sp::Worker();
  S->worker_busy = true;
  c->pcs_worker->submit([S, c, num_cols] {
    // delta = <d, ck> + r_delta h first: its walk (auxiliary stream) reduces the randomness blocks itself
    if (hipSetDevice(c->device) != hipSuccess ||
        sp::multi_mul_launch(c, 1, S->ck->d_keytables, reinterpret_cast<const uint64_t*>(S->rng.data()), num_cols + 1, &S->seq_delta, nullptr, &S->r_delta, S->cols))
      S->failed = true;
    else
      S->delta_launched = true;
    static const char* b = "poly_commitment_begin";  // HyraxCommitment::to_transcript_bytes (hyrax_pc.rs:714-729)
    static const char* e = "poly_commitment_end";
    S->hashed.init();
    S->hashed.update(reinterpret_cast<const uint8_t*>("poly_com"), 8);
    S->hashed.update(reinterpret_cast<const uint8_t*>(b), strlen(b));
    uint8_t buf[64 * 16];
    for (size_t i = 0; i < S->comm.size(); i += 16) {
      const size_t m = S->comm.size() - i < 16 ? S->comm.size() - i : 16;
      for (size_t k = 0; k < m; ++k) hp_point_bytes(S->comm[i + k], buf + 64 * k);
      S->hashed.update(buf, 64 * m);
    }
    S->hashed.update(reinterpret_cast<const uint8_t*>(e), strlen(e));
    for (size_t i = 0; i < S->cols; ++i) S->dvec[i] = fe_from_uniform<spk::SF>(S->rng.data() + 64 * i);
    // the mask vector to the device, behind delta's walk on the auxiliary stream (needed when the IPA's challenge is drawn: z_vec = r LZ + d)
    fe_t* stage = reinterpret_cast<fe_t*>(c->h_pinned_vec) + c->h_pinned_vec_cols;
    memcpy(stage, S->dvec.data(), S->cols * sizeof(fe_t));
    if (!S->failed && hipMemcpyAsync(S->d_out + num_cols + 1, stage, S->cols * sizeof(fe_t), hipMemcpyHostToDevice, c->stream2) == hipSuccess) S->dvec_uploaded = true;
  });
  return SP_OK;
}
extern "C" {

int sp_hyrax_prove(sp_ctx* c, const sp_ck* ck, const sp_ck* ck_eval, sp_transcript* tr, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n,
                   const uint64_t* blinds, const uint64_t* point, size_t npt, const uint64_t comm_eval_aff[8], const uint64_t blind_eval[4], const uint8_t* rng,
                   size_t rng_blocks, uint64_t* out) {
  typedef spk::SF SF;
  if (!c || !ck || !ck_eval || !tr || !comm_rows_aff || !poly || !blinds || (!point && npt) || !comm_eval_aff || !blind_eval || !rng || !out)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_hyrax_prove: null argument");
  if (npt > 40 || n != ((size_t)1 << npt) || n > poly->cap)
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "Hyrax prove: Expected 2^point.len() elements in poly");  // hyrax_pc.rs:400-408
  const size_t num_cols = ck->num_cols, num_rows = (n + num_cols - 1) / num_cols;
  if (num_rows & (num_rows - 1)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "Hyrax prove: the row count must be a power of two");
  if (rows != num_rows) return fail(SP_ERR_INVALID_INPUT_LENGTH, "Hyrax prove: one commitment row and one blind per matrix row");
  size_t nvr = 0;
  while (((size_t)1 << nvr) < num_rows) ++nvr;
  const size_t cols = n / num_rows;  // |R| = |LZ| = |d|
  if (!ck_eval->d_cktables || ck_eval->num_cols < 1) return fail(SP_ERR_INVALID_INPUT_LENGTH, "Hyrax prove: ck_eval must be a narrow key with tables");
  SP_HIP(hipSetDevice(c->device));
  static const bool laps = getenv("SPARTAN_HOST_LAPS") != nullptr;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_lap = laps ? now_us() : 0;
  auto lap = [&](const char* what) {
    if (!laps) return;
    const double t = now_us();
    fprintf(stderr, "  hyrax_prove lap %-24s %8.1f us\n", what, t - t_lap);
    t_lap = t;
  };
  const aff_t* comm = reinterpret_cast<const aff_t*>(comm_rows_aff);
  const fe_t* blind = reinterpret_cast<const fe_t*>(blinds);
  const fe_t* pt = reinterpret_cast<const fe_t*>(point);
  if (rng_blocks < cols + 2) return fail(SP_ERR_INVALID_INPUT_LENGTH, "Hyrax prove: the randomness stream holds fewer than cols + 2 blocks");
  // E::Scalar::random draws (ipa.rs:139-149): d_vec, then the blinds of delta and beta, each the wide reduction of 64 uniform bytes. The two blinds
  // now, the mask vector further down while the device already works on LZ.
  std::vector<fe_t> dvec_own;
  const fe_t r_delta = fe_from_uniform<SF>(rng + 64 * cols), r_beta = fe_from_uniform<SF>(rng + 64 * (cols + 1));
  fe_t b_eval;
  memcpy(&b_eval, blind_eval, 32);
  aff_t comm_eval;
  memcpy(&comm_eval, comm_eval_aff, sizeof(aff_t));

  // (0) was this opening announced (sp_hyrax_prove_announce)? Then its commitment hashing, mask vector, delta and — if the inner sum-check reported the row
  // challenges — L^T W and comm_LZ are done or under way; anything that differs from what was announced drops the announcement.
  sp_pcs_ahead* S = c->pcs_ahead;
  bool ahead = S && !S->failed && S->ck == ck && S->poly == poly && S->n == n && S->comm.size() == rows && memcmp(S->comm.data(), comm, rows * sizeof(aff_t)) == 0 &&
               memcmp(S->blind.data(), blind, rows * sizeof(fe_t)) == 0 && memcmp(S->rng.data(), rng, 64 * (cols + 2)) == 0;
  if (S && !ahead) {
    sp::pcs_ahead_free(c);
    S = nullptr;
  }
  struct AheadDone {  // consumed or not, the announcement ends with this call
    sp_ctx* c;
    ~AheadDone() { sp::pcs_ahead_free(c); }
  } ahead_done{c};
  // (1) helper thread: transcript.absorb(b"poly_com", comm) (hyrax_pc.rs:410) into a copy of the running hasher, installed at the join below
  tr->join();
  if (!c->pcs_worker) c->pcs_worker = new sp::Worker();
  sp::Keccak256State hashed = tr->t.h;
  bool hashed_ahead = false;
  if (ahead) {  // the announced hashing started from a fresh sponge: usable exactly when nothing has been absorbed since the last squeeze
    c->pcs_worker->wait();
    S->worker_busy = false;
    if (S->failed || !S->delta_launched) {  // the helper could not launch delta's walk: as if nothing had been announced
      sp::pcs_ahead_free(c);
      S = nullptr;
      ahead = false;
    }
  }
  if (ahead) {
    bool fresh = tr->t.h.fill == 0;
    for (int i = 0; i < 25 && fresh; ++i) fresh = tr->t.h.a[i] == 0;
    if (fresh) {
      hashed = S->hashed;
      hashed_ahead = true;
    }
  }
  if (!hashed_ahead) {
    sp::Keccak256State* hp = &hashed;
    c->pcs_worker->submit([hp, comm, rows] {
      static const char* b = "poly_commitment_begin";  // HyraxCommitment::to_transcript_bytes (hyrax_pc.rs:714-729)
      static const char* e = "poly_commitment_end";
      hp->update(reinterpret_cast<const uint8_t*>("poly_com"), 8);
      hp->update(reinterpret_cast<const uint8_t*>(b), strlen(b));
      uint8_t buf[64 * 16];
      for (size_t i = 0; i < rows; i += 16) {
        const size_t m = rows - i < 16 ? rows - i : 16;
        for (size_t k = 0; k < m; ++k) hp_point_bytes(comm[i + k], buf + 64 * k);
        hp->update(buf, 64 * m);
      }
      hp->update(reinterpret_cast<const uint8_t*>(e), strlen(e));
    });
  }
  // beta = ck_c <R, d> + h r_beta (ipa.rs:149) is two table walks on the host, ~11 us each. With the helper thread polling (an announced opening keeps it
  // so) and not busy with the commitment's hashing, h's walk runs beside this thread's compare / <R, d> and ck_c's beside its collection of the two device
  // walks; otherwise both are this thread's (sp_hyrax_commit_small). The jobs write here: declared in front of `join`, which waits on every exit path.
  jac_t beta_h = jac_identity(), beta_c = jac_identity();
  const bool beta_beside = hashed_ahead && ck_eval->n_tables >= 2 && ck_eval->num_cols >= 1 && c->pcs_worker->hot();
  struct Join {  // the hashing job reads `comm` and writes `hashed`: it must have finished on every exit path (not so the z_vec helper job further down)
    sp::Worker* w;
    bool joined = false;
    ~Join() {
      if (!joined) w->wait();
    }
  } join{c->pcs_worker};

  if (beta_beside) {
    jac_t* dst = &beta_h;
    const fe_t rb = r_beta;
    c->pcs_worker->submit([ck_eval, dst, rb] { *dst = ck_mul_host(ck_eval, ck_eval->n_tables - 1, rb); });
  }
  lap("submit hashing");
  // (2) device: delta's walk on the auxiliary stream, LZ and comm_LZ's walk on the main stream
  const int kt = (nvr == 0 && cols > num_cols) ? 1 : sp::ck_key_tables(c, ck);
  if (kt < 0) return kt;
  const bool walk = kt == 0;
  std::vector<fe_t> LZ;
  fe_t r_LZ;
  aff_t comm_LZ, delta;
  unsigned seq_delta = 0, seq_lz = 0;
  sp_msm_job* delta_job = nullptr;
  struct JobGuard {  // an error exit must not leave the asynchronous MSM of the fallback path outstanding
    sp_ctx* c;
    sp_msm_job*& j;
    ~JobGuard() {
      uint64_t sink[8];
      if (j) sp_msm_job_finish(c, j, sink);
      j = nullptr;
    }
  } job_guard{c, delta_job};
  int rc;
  // delta = <d, ck> + r_delta h first: it needs nothing but the randomness stream, whose blocks the kernel reduces itself
  const bool delta_raw = walk && num_cols + 1 >= sp::multi_mul_wide_min() && 64 * num_cols <= 4 * (size_t)spk::MULTI_MUL_MAX_BLOCKS * sizeof(fe_t);
  if (ahead && !walk) {  // (cannot happen: the announcement needs the key's window tables)
    sp::pcs_ahead_free(c);
    S = nullptr;
    ahead = false;
  }
  if (ahead) seq_delta = S->seq_delta;
  else if (delta_raw && (rc = sp::multi_mul_launch(c, 1, ck->d_keytables, reinterpret_cast<const uint64_t*>(rng), num_cols + 1, &seq_delta, nullptr, &r_delta, cols))) return rc;
  lap("delta launch");
  std::vector<fe_t> L;
  bool lz_ahead = false;
  if (nvr == 0) {  // a single row: the commitment is the row itself (hyrax_pc.rs:417-423)
    comm_LZ = comm[0];
    LZ.resize(cols);
    if ((rc = sp_table_read(c, poly, 0, n, reinterpret_cast<uint64_t*>(LZ.data())))) return rc;
    r_LZ = blind[0];
  } else if (ahead && S->lz_launched && memcmp(S->row_pt.data(), pt, nvr * sizeof(fe_t)) == 0) {
    lz_ahead = true;  // L^T W and comm_LZ's walk have been running on the auxiliary stream since the inner sum-check drew the row challenges
    r_LZ = S->r_LZ;
    seq_lz = S->seq_lz;
  } else {
    L.resize((size_t)1 << nvr);
    eq_table_host(pt, nvr, L.data());
    LZ.resize(cols);
    if (ahead && S->lz_launched) {  // started for other row challenges than the point given now: drain it, its lane is needed
      if (S->z_armed && !S->z_fired) {
        scale_add_abort(c, S->z_armed);
        S->z_fired = true;
      }
      jac_t sink;
      (void)sp::multi_mul_collect(c, 1, S->seq_lz, &sink, false);
      (void)sp::event_sync(c->pcs_ev);
      S->lz_launched = false;
    }
    r_LZ = fe_zero();
    for (size_t i = 0; i < num_rows; ++i) r_LZ = fe_add<SF>(r_LZ, fe_mul<SF>(L[i], blind[i]));
    if (walk) {
      if (!c->pcs_ev) SP_HIP(hipEventCreateWithFlags(&c->pcs_ev, hipEventDisableTiming));
      if (c->h_pcs_bytes < cols * sizeof(fe_t)) {
        if (c->h_pcs) hipHostFree(c->h_pcs);
        c->h_pcs = nullptr;
        c->h_pcs_bytes = 0;
        SP_HIP(hipHostMalloc(&c->h_pcs, cols * sizeof(fe_t)));
        c->h_pcs_bytes = cols * sizeof(fe_t);
      }
      const size_t splits = num_rows < 64 ? num_rows : 64;
      fe_t* dL = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_L, num_rows * sizeof(fe_t));
      fe_t* part = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_PART, splits * cols * sizeof(fe_t));
      fe_t* dout = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_OUT, (num_cols + 1) * sizeof(fe_t));
      if (!dL || !part || !dout) return SP_ERR_NO_DEVICE;
      if (nvr <= 10) {  // L = left (x) right from two half tables passed by value
        spk::EqTensorArgs a;
        const size_t hb = nvr / 2, lb = nvr - hb;
        eq_table_host(pt, hb, a.left);
        eq_table_host(pt + hb, lb, a.right);
        a.lo_bits = (int)lb;
        a.n = (unsigned)num_rows;
        hipLaunchKernelGGL(spk::k_eq_tensor<false>, dim3((unsigned)((num_rows + 255) / 256)), dim3(256), 0, c->stream, a, dL);
      } else {
        SP_HIP(hipMemcpyAsync(dL, L.data(), num_rows * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));  // (L outlives the stream work: joined below)
      }
      if (cols < num_cols) SP_HIP(hipMemsetAsync(dout + cols, 0, (num_cols - cols) * sizeof(fe_t), c->stream));
      c->timed("rowmat_vec", 32ull * (num_rows * cols + num_rows + cols), [&] { launch_rowmat_vec(c->stream, poly->d, num_rows, cols, dL, part, splits, dout); });
      if ((rc = sp::multi_mul_launch(c, 0, ck->d_keytables, nullptr, num_cols + 1, &seq_lz, dout, &r_LZ, 0))) return rc;
      SP_HIP(hipMemcpyAsync(c->h_pcs, dout, cols * sizeof(fe_t), hipMemcpyDeviceToHost, c->stream));  // behind the walk: LZ is needed on the host only for z_vec
      SP_HIP(hipEventRecord(c->pcs_ev, c->stream));
    } else {
      if ((rc = sp_rowmat_vec(c, poly, num_rows, cols, reinterpret_cast<const uint64_t*>(L.data()), reinterpret_cast<uint64_t*>(LZ.data())))) return rc;
    }
  }
  lap("LZ + comm_LZ launch");
  // d_vec: 2048 wide reductions at config 2, ~65 us of host work beside the device's (delta's walk reduces its own copy of the blocks)
  if (!ahead) {  // (an announcement's helper job has drawn it)
    dvec_own.resize(cols);
    for (size_t i = 0; i < cols; ++i) dvec_own[i] = fe_from_uniform<SF>(rng + 64 * i);
  }
  const std::vector<fe_t>& dvec = ahead ? S->dvec : dvec_own;
  lap("d_vec draw");
  if (!walk && (rc = sp_msm_ck_begin(c, ck, reinterpret_cast<const uint64_t*>(dvec.data()), cols, &delta_job))) return rc;
  if (walk && !delta_raw && !ahead) {  // narrow keys: the walk from the drawn scalars
    std::vector<fe_t> sc(num_cols + 1, fe_zero());
    memcpy(sc.data(), dvec.data(), cols * sizeof(fe_t));
    sc[num_cols] = r_delta;
    if ((rc = sp::multi_mul_launch(c, 1, ck->d_keytables, reinterpret_cast<const uint64_t*>(sc.data()), num_cols + 1, &seq_delta, nullptr, nullptr, 0))) return rc;
  }
  // (3) host work under the device's: <R, d> with R = eq(point[nvr..]) = left (x) right (ipa.rs:148), beta = ck_c * <R, d> + h_c * r_beta (:149)
  fe_t ip = fe_zero();
  if (ahead && S->T_ready && S->hb == (npt - nvr) / 2 && S->col_pt_T.size() >= S->hb && memcmp(S->col_pt_T.data(), pt + nvr, S->hb * sizeof(fe_t)) == 0) {
    const size_t k = npt - nvr;
    std::vector<fe_t> right((size_t)1 << (k - S->hb));
    eq_table_host(pt + nvr + S->hb, k - S->hb, right.data());
    for (size_t b2 = 0; b2 < right.size(); ++b2) ip = fe_add<SF>(ip, fe_mul<SF>(right[b2], S->T[b2]));
  } else {
    const size_t k = npt - nvr, hb = k / 2;
    std::vector<fe_t> left((size_t)1 << hb), right((size_t)1 << (k - hb));
    eq_table_host(pt + nvr, hb, left.data());
    eq_table_host(pt + nvr + hb, k - hb, right.data());
    for (size_t a = 0; a < left.size(); ++a) {
      fe_t inner = fe_zero();
      for (size_t b2 = 0; b2 < right.size(); ++b2) inner = fe_add<SF>(inner, fe_mul<SF>(right[b2], dvec[a * right.size() + b2]));
      ip = fe_add<SF>(ip, fe_mul<SF>(left[a], inner));
    }
  }
  lap("<R, d>");
  aff_t beta;
  if (beta_beside) {
    c->pcs_worker->wait(1);  // h's walk (an unclaimed job is taken back and run here)
    jac_t* dst = &beta_c;
    const fe_t ipv = ip;
    c->pcs_worker->submit([ck_eval, dst, ipv] { *dst = ck_mul_host(ck_eval, 0, ipv); });
  } else if ((rc = sp_hyrax_commit_small(c, ck_eval, reinterpret_cast<const uint64_t*>(&ip), 1, reinterpret_cast<const uint64_t*>(&r_beta), reinterpret_cast<uint64_t*>(&beta)))) {
    return rc;
  }
  lap("beta");
  // (4) joins
  if (walk) {
    jac_t dj;
    if (ahead && S->delta_collected) {
      dj = S->delta_j;
    } else {
      if ((rc = sp::multi_mul_collect(c, 1, seq_delta, &dj, false))) return rc;
      if (ahead) S->delta_collected = true;
    }
    delta = jac_to_affine(dj);
    if (nvr != 0) {
      jac_t lj;
      if ((rc = sp::multi_mul_collect(c, lz_ahead ? 1 : 0, seq_lz, &lj, false))) return rc;
      if (lz_ahead) S->lz_launched = false;
      comm_LZ = jac_to_affine(lj);
      if (!(lz_ahead && S->dvec_uploaded)) {  // (an announced opening forms z_vec on the device: LZ never comes to the host)
        if (lz_ahead) {
          LZ.resize(cols);
          SP_HIP(hipMemcpyAsync(c->h_pcs, S->d_out, cols * sizeof(fe_t), hipMemcpyDeviceToHost, c->stream2));
          SP_HIP(hipEventRecord(c->pcs_ev, c->stream2));
        }
        SP_HIP(sp::event_sync(c->pcs_ev));
        memcpy(LZ.data(), c->h_pcs, cols * sizeof(fe_t));
      }
    }
  } else {
    if (nvr != 0 && (rc = sp_msm_ck(c, ck, reinterpret_cast<const uint64_t*>(LZ.data()), cols, reinterpret_cast<const uint64_t*>(&r_LZ), reinterpret_cast<uint64_t*>(&comm_LZ))))
      return rc;
    sp_msm_job* j = delta_job;
    delta_job = nullptr;
    if ((rc = sp_msm_ck_finish(c, ck, j, reinterpret_cast<const uint64_t*>(&r_delta), reinterpret_cast<uint64_t*>(&delta)))) return rc;
  }
  if (beta_beside) {
    c->pcs_worker->wait(1);
    beta = jac_to_affine(jac_add(beta_c, beta_h));
  }
  lap("join walks");
  if (!hashed_ahead) c->pcs_worker->wait();
  join.joined = true;
  lap("join hashing");
  tr->t.h = hashed;
  // (5) InnerProductArgumentLinear::prove, transcript part (ipa.rs:132-158)
  {
    static const char* ds = "inner product argument (linear)";
    tr->t.dom_sep(reinterpret_cast<const uint8_t*>(ds), strlen(ds));
    uint8_t b[128];
    hp_point_bytes(comm_LZ, b);
    hp_point_bytes(comm_eval, b + 64);
    tr->t.absorb(reinterpret_cast<const uint8_t*>("U"), 1, b, 128);
    hp_point_bytes(delta, b);
    tr->t.absorb(reinterpret_cast<const uint8_t*>("delta"), 5, b, 64);
    hp_point_bytes(beta, b);
    tr->t.absorb(reinterpret_cast<const uint8_t*>("beta"), 4, b, 64);
  }
  fe_t rr;
  if (!tr->t.squeeze<SF>(reinterpret_cast<const uint8_t*>("r"), 1, &rr)) return fail(SP_ERR_INTERNAL_TRANSCRIPT, "transcript round counter overflow");
  // (6) z_vec = r * LZ + d, z_delta = r * r_LZ + r_delta, z_beta = r * blind_eval + r_beta (ipa.rs:160-168)
  memcpy(out, &delta, sizeof(aff_t));
  memcpy(out + 8, &beta, sizeof(aff_t));
  fe_t* zv = reinterpret_cast<fe_t*>(out + 16);
  if (lz_ahead && S->dvec_uploaded) {
    // on the device, behind comm_LZ's walk on the auxiliary stream (collected above): one launch, the sum lands in mapped memory (2048 products on this
    // thread and its helper were ~32 us)
    if (S->z_armed && !S->z_fired) {
      S->z_fired = true;
      if ((rc = scale_add_fire(c, c->stream3, S->z_armed, S->d_out, S->d_out + num_cols + 1, rr, cols, out + 16, "hyrax_prove z_vec"))) return rc;
    } else if ((rc = scale_add_to_host(c, c->stream2, S->d_out, S->d_out + num_cols + 1, rr, cols, out + 16, "hyrax_prove z_vec"))) {
      return rc;
    }
    explicit_bzero(reinterpret_cast<fe_t*>(c->h_pinned_vec) + c->h_pinned_vec_cols, cols * sizeof(fe_t));  // the mask's staging copy
    (void)hipMemsetAsync(S->d_out + num_cols + 1, 0, cols * sizeof(fe_t), c->stream2);                       // and its device copy
    S->dvec_uploaded = false;  // (both copies are gone: pcs_ahead_free has nothing left to wipe)
  } else {
    // shared with the helper thread chunk by chunk: whoever is awake takes the next 128 elements. The owner never waits for the helper to WAKE (a sleeping
    // thread can take milliseconds when the process is at its CPU quota), only for chunks the helper has actually claimed.
    struct Share {
      std::atomic<size_t> next{0}, done{0};
    };
    auto sh = std::make_shared<Share>();
    const fe_t* lz = LZ.data();
    const fe_t* dv = dvec.data();
    constexpr size_t CH = 128;
    const size_t nch = (cols + CH - 1) / CH;
    auto work = [sh, zv, lz, dv, rr, cols, nch] {
      for (;;) {
        const size_t ci = sh->next.fetch_add(1, std::memory_order_acq_rel);
        if (ci >= nch) return;
        const size_t hi = (ci + 1) * CH < cols ? (ci + 1) * CH : cols;
        for (size_t i = ci * CH; i < hi; ++i) zv[i] = fe_add<SF>(fe_mul<SF>(rr, lz[i]), dv[i]);
        sh->done.fetch_add(1, std::memory_order_acq_rel);
      }
    };
    if (nch > 2) c->pcs_worker->submit(work);
    work();
    while (sh->done.load(std::memory_order_acquire) < nch) sp::relax();
  }
  zv[cols] = fe_add<SF>(fe_mul<SF>(rr, r_LZ), r_delta);
  zv[cols + 1] = fe_add<SF>(fe_mul<SF>(rr, b_eval), r_beta);
  lap("transcript + z_vec");
  return SP_OK;
}

// PCS::commit of a HOST vector on a narrow key (hyrax_pc.rs:221-260: every row a FixedBaseMul::multi_mul + h * blind), the latency form: the (row, column)
// scalars and the row blinds go to the device through the mapped page as ONE launch of the cooperative table walk over the 16-bit windows (four tree
// levels, ~26-30 us for up to 640 scalars: the chain of dependent additions is all there is), every scalar's point comes back as (X, Y, ZZ, ZZZ) in a
// self-validating slot, and a row's cols + 1 points are added by the polling host threads, a row each (~33 additions, 10 us). Against the device-table
// form (upload, three device-to-device copies, two kernels - the second a shuffle tree of plain additions - a copy back, a synchronise): 232 -> ~75 us for
// the 16 rows of NovaNIFS's cross term at config 3.
struct RowsHostJob {
  sp_ctx* c;
  int lane;
  unsigned seq;
  size_t per, rows;
  std::vector<jac_t> out;
  std::atomic<int> rc{SP_OK};
};
static void rows_host_part(void* arg, unsigned part, unsigned np) {
  RowsHostJob& J = *static_cast<RowsHostJob*>(arg);
  std::vector<xyzz_t> xs(J.per);
  for (size_t r = J.rows * part / np; r < J.rows * (part + 1) / np; ++r) {
    // the slots of row r: scalars r * per .. (r + 1) * per - 1
    unsigned* out = reinterpret_cast<unsigned*>(xs.data());
    const int T = spk::FB_SLOT_TAG;
    for (size_t i = 0; i < J.per; ++i) {
      volatile const unsigned* slot = reinterpret_cast<volatile const unsigned*>((char*)J.c->h_fbm[J.lane] + FB_SLOT_BYTES * (r * J.per + i));
      unsigned w[32];
      long spins = 0;
      for (;; ++spins) {
        if (slot[T] == J.seq) {
          std::atomic_thread_fence(std::memory_order_acquire);
          unsigned a = J.seq, b = J.seq * 0x9E3779B1u;
          for (int k = 0; k < 32; ++k) {
            w[k] = slot[k];
            a += w[k];
            b += (unsigned)(k + 1) * w[k];
          }
          if (slot[T] == J.seq && slot[T + 1] == a && slot[T + 2] == b && slot[T + 3] == J.seq) break;
        }
        if (spins > 40000000) {  // seconds: the kernel never delivered (the caller synchronises and reports)
          J.rc.store(SP_ERR_INTERNAL, std::memory_order_relaxed);
          return;
        }
        __builtin_ia32_pause();
      }
      memcpy(out + 32 * i, w, 128);
    }
    xyzz_t acc = xyzz_identity();
    for (size_t i = 0; i < J.per; ++i) acc = xyzz_add(acc, xs[i]);
    J.out[r] = xyzz_to_jac(acc);
  }
}
int sp_hyrax_commit_rows_host(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t* blinds, uint64_t* out_rows_aff) {
  if (!ck->d_cktables) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_rows_host: the key is wider than 64");
  const size_t cols = ck->num_cols, rows = (n + cols - 1) / cols, per = cols + 1;
  if (rows == 0) return SP_OK;
  if (!scalars || !blinds || !out_rows_aff) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_rows_host: null argument");
  const aff_t* t16 = fb_window16_enabled() ? tables16_of(ck->d_cktables) : nullptr;
  if (rows * per > FB_MAPPED_CAP || !fb_mapped_enabled() || !t16 || c->fb_async_busy) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_rows_host: too many rows for the mapped form (stage the vector and call sp_hyrax_commit)");
  std::vector<fe_t> sc(rows * per, fe_zero());
  for (size_t r = 0; r < rows; ++r) {
    const size_t lo = r * cols, len = lo + cols <= n ? cols : n - lo;
    memcpy(&sc[r * per], scalars + 4 * lo, len * sizeof(fe_t));
    memcpy(&sc[r * per + cols], blinds + 4 * r, sizeof(fe_t));
  }
  // table of scalar i = i % per: the key's columns, then h - the layout of the per-base table set (ntables = per)
  const int lane = c->hooks_host_only ? 1 : 0;  // (see sp_hyrax_commit_small)
  int rc = fb_mapped_launch(c, lane, ck->d_cktables, per, reinterpret_cast<const uint64_t*>(sc.data()), rows * per, true);
  if (rc) return rc;
  RowsHostJob J{c, lane, c->fbm_seq[lane], per, rows};
  J.out.assign(rows, jac_identity());
  sp::WalkPool& pool = sp::WalkPool::get();
  pool.keep_hot(2000);
  const unsigned np = (unsigned)std::min<size_t>(rows, (size_t)pool.walkers() + 1);
  pool.run(np ? np : 1u, rows_host_part, &J);
  if (J.rc.load() != SP_OK) {
    (void)sp::stream_sync(lane ? c->stream2 : c->stream);
    return fail(SP_ERR_INTERNAL, "commit_rows_host: the table walk did not deliver its result slots");
  }
  std::vector<aff_t> a(rows);
  normalize_batch(J.out, a.data());
  memcpy(out_rows_aff, a.data(), rows * sizeof(aff_t));
  return SP_OK;
}

// sum_i scalars[i] * ck[i] + blind_term, blind_term = h * blind computed by the caller beforehand (sp_fixed_base_mul_h[_begin]: the blind comes from the
// randomness stream and is known long before the scalars - the commitment of eval_W behind the inner sum-check, src/spartan.rs:423-437): the host walk of
// h's table (32 additions, ~11 us) is off the path. Identity = both coordinates zero.
int sp_hyrax_commit_small_with_term(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t blind_term_aff[8], uint64_t out_aff[8]) {
  if (!ck->d_cktables || n > ck->num_cols) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_small: key wider than 64 or too many scalars");
  if (n > 6) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_small_with_term: at most 6 scalars (the host-walk case)");
  aff_t term;
  memcpy(&term, blind_term_aff, sizeof(aff_t));
  jac_t acc = (fe_is_zero(term.x) && fe_is_zero(term.y)) ? jac_identity() : jac_from_affine(term);
  for (size_t i = 0; i < n; ++i) {
    fe_t sc;
    memcpy(&sc, scalars + 4 * i, 32);
    acc = jac_add(acc, ck_mul_host(ck, i, sc));
  }
  (void)c;
  store_aff(out_aff, jac_to_affine(acc));
  return SP_OK;
}
int sp_hyrax_commit_small(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t blind[4], uint64_t out_aff[8]) {
  if (!ck->d_cktables || n > ck->num_cols) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_small: key wider than 64 or too many scalars");
  // FixedBaseMul::multi_mul (msm.rs:727-773) + h_table.mul(blind). A handful of scalars: host (single lookup chains). More (a round of the ZK
  // verifier circuit commits up to 32 values, 41 times per NeutronNova proof): every scalar's table walk on its own device lanes, one launch.
  size_t nonzero = 0;
  for (size_t i = 0; i < n; ++i) nonzero += (scalars[4 * i] | scalars[4 * i + 1] | scalars[4 * i + 2] | scalars[4 * i + 3]) != 0;
  if (nonzero > 6) {
    const size_t cols = ck->num_cols;
    std::vector<fe_t> sc(cols + 1, fe_zero());
    memcpy(sc.data(), scalars, n * sizeof(fe_t));
    memcpy(&sc[cols], blind, sizeof(fe_t));
    if (cols + 1 <= FB_MAPPED_MAX && fb_mapped_enabled()) {  // the walks come back as (X, Y, ZZ, ZZZ) and are added in that form
      std::vector<xyzz_t> xs(cols + 1);
      // on the auxiliary stream while the caller has promised host-only round hooks (sp_ctx_round_hooks_host_only): a batched sum-check may have a launch
      // waiting for this very hook's challenge on the main stream, and a walk queued behind it would wait for the mailbox watchdog
      const int lane = c->hooks_host_only && !c->fb_async_busy ? 1 : 0;
      int rc = fb_mapped_launch(c, lane, ck->d_cktables, cols + 1, reinterpret_cast<const uint64_t*>(sc.data()), cols + 1, true);
      if (rc || (rc = fb_mapped_collect(c, lane, cols + 1, xs.data(), c->fbm_seq[lane], 32))) return rc;
      xyzz_t acc = xyzz_identity();
      for (const xyzz_t& p : xs) acc = xyzz_add(acc, p);
      store_aff(out_aff, jac_to_affine(xyzz_to_jac(acc)));
      return SP_OK;
    }
    std::vector<jac_t> pts;
    int rc = fixed_base_rows(c, ck->d_cktables, cols + 1, reinterpret_cast<const uint64_t*>(sc.data()), cols + 1, pts);
    if (rc) return rc;
    jac_t acc = jac_identity();
    for (const jac_t& p : pts) acc = jac_add(acc, p);
    store_aff(out_aff, jac_to_affine(acc));
    return SP_OK;
  }
  std::vector<jac_t> parts;
  fe_t b;
  memcpy(&b, blind, 32);
  // h's walk beside the scalars' when the context's helper thread is polling for work (an announced opening keeps it so: the commitment of eval_W and
  // beta, ~11 us a walk); a job the helper has not claimed within a microsecond is taken back and run here
  jac_t hpart;
  const bool beside = n >= 1 && c && c->pcs_worker && c->pcs_worker->hot();
  if (beside) {
    jac_t* hp = &hpart;
    const fe_t* bp = &b;
    c->pcs_worker->submit([ck, hp, bp] { *hp = ck_mul_host(ck, ck->n_tables - 1, *bp); });
  }
  for (size_t i = 0; i < n; ++i) {
    fe_t sc;
    memcpy(&sc, scalars + 4 * i, 32);
    parts.push_back(ck_mul_host(ck, i, sc));
  }
  if (beside) {
    c->pcs_worker->wait(1);
    parts.push_back(hpart);
  } else {
    parts.push_back(ck_mul_host(ck, ck->n_tables - 1, b));
  }
  jac_t acc = jac_identity();
  for (const jac_t& p : parts) acc = jac_add(acc, p);
  store_aff(out_aff, jac_to_affine(acc));
  return SP_OK;
}

// ---- a narrow commitment in two calls: the terms known early, then the rest (walk_pool.hpp) -------------------------------------------------------------
// commit(row, blind) = sum_i row[i] ck[i] + blind h (hyrax_pc.rs:221-260 on a key with per-base tables, msm.rs:727-773) is linear in the row: the terms
// that do not depend on the newest prover message (the Horner steps of the previous round's polynomial at its challenge, the blind) are walked while the
// device computes that message; what the transcript then waits for is the walk of the message's own few scalars.
struct sp_split_commit {
  const sp_ck* ck = nullptr;
  sp::WalkPool::Batch* early = nullptr;  // posted at begin (nullptr: the pool had no free slot - the terms are kept and walked at finish)
  std::vector<const aff_t*> kept;
  bool any_early = false;
  ~sp_split_commit() {
    if (!kept.empty()) explicit_bzero(kept.data(), kept.size() * sizeof(kept[0]));  // (entry pointers = digits of the scalars, the blind's among them)
  }
};
static int split_entries(const sp_ck* ck, const uint32_t* cols, const uint64_t* scalars, size_t n, const uint64_t* blind, std::vector<const aff_t*>& ents) {
  auto walk = [&](size_t table, const uint64_t* sc4) -> int {
    fe_t s;
    memcpy(&s, sc4, 32);
    if (fe_is_zero(s)) return SP_OK;
    const aff_t* t16 = ck->host_table16(table);
    if (!t16) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_split: no host table for this column (the key keeps its first 16 columns and h)");
    const fe_t c = fe_to_canonical<S>(s);
    for (int j = 0; j < 16; ++j) {
      const unsigned digit = (c.v[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      if (digit) ents.push_back(t16 + (size_t)j * 65535 + digit - 1);
    }
    return SP_OK;
  };
  int rc;
  for (size_t i = 0; i < n; ++i) {
    if (cols[i] >= ck->num_cols) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_split: column outside the key");
    if ((rc = walk(cols[i], scalars + 4 * i))) return rc;
  }
  if (blind && (rc = walk(ck->n_tables - 1, blind))) return rc;
  return SP_OK;
}
static unsigned split_parts(size_t n_ents) {  // ~12 entries a part: the owner adds the parts' sums one after the other (0.4 us each)
  const unsigned w = (unsigned)sp::WalkPool::get().walkers() + 1;
  unsigned p = (unsigned)((n_ents + 11) / 12);
  if (p > w) p = w;
  return p ? p : 1;
}
int sp_walkers(void) { return sp::WalkPool::get().walkers(); }
int sp_walkers_keep_hot(uint64_t microseconds) {
  sp::WalkPool::get().keep_hot((long)microseconds);
  return SP_OK;
}
int sp_host_parallel_for(unsigned nparts, sp_part_fn fn, void* arg) {
  if (!fn) return fail(SP_ERR_INVALID_INPUT_LENGTH, "parallel_for: no function");
  sp::WalkPool::get().run(nparts ? nparts : 1u, fn, arg);
  return SP_OK;
}
int sp_hyrax_commit_split_available(const sp_ck* ck, size_t cols_used) {
  return ck && ck->d_cktables && ck->h_tables16 && ck->h16_bases >= cols_used && cols_used <= ck->num_cols && sp::WalkPool::get().walkers() > 0;
}
int sp_hyrax_commit_split_begin(sp_ctx* c, const sp_ck* ck, const uint32_t* cols, const uint64_t* scalars, size_t n, const uint64_t* blind, sp_split_commit** out) {
  (void)c;
  if (!ck->d_cktables || !ck->h_tables16) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_split: the key has no host-side window tables");
  std::unique_ptr<sp_split_commit> job(new sp_split_commit());
  job->ck = ck;
  int rc = split_entries(ck, cols, scalars, n, blind, job->kept);
  if (rc) return rc;
  sp::WalkPool& pool = sp::WalkPool::get();
  pool.keep_hot(2000);
  if (!job->kept.empty() && job->kept.size() <= (size_t)sp::WalkPool::MAX_ENTS && (job->early = pool.acquire())) {
    memcpy(job->early->ents, job->kept.data(), job->kept.size() * sizeof(const aff_t*));
    job->early->n_ents = (unsigned)job->kept.size();
    pool.post(job->early, split_parts(job->kept.size()));
    job->kept.clear();
  }
  *out = job.release();
  return SP_OK;
}
int sp_hyrax_commit_split_finish(sp_ctx* c, sp_split_commit* job_, const uint32_t* cols, const uint64_t* scalars, size_t n, uint64_t out_aff[8]) {
  (void)c;
  std::unique_ptr<sp_split_commit> job(job_);
  sp::WalkPool& pool = sp::WalkPool::get();
  std::vector<const aff_t*>& ents = job->kept;  // (the early terms too when they could not be posted)
  int rc = split_entries(job->ck, cols, scalars, n, nullptr, ents);
  if (rc) {
    if (job->early) (void)pool.finish(job->early);
    return rc;
  }
  pool.keep_hot(2000);
  xyzz_t acc = xyzz_identity();
  sp::WalkPool::Batch* late = nullptr;
  if (ents.size() > 8 && ents.size() <= (size_t)sp::WalkPool::MAX_ENTS && (late = pool.acquire())) {
    memcpy(late->ents, ents.data(), ents.size() * sizeof(const aff_t*));
    late->n_ents = (unsigned)ents.size();
    pool.post(late, split_parts(ents.size()));
    acc = pool.finish(late);
  } else {
    for (const aff_t* e : ents) __builtin_prefetch(e, 0, 0);
    for (const aff_t* e : ents) acc = xyzz_add_mixed(acc, *e);
  }
  if (job->early) acc = xyzz_add(acc, pool.finish(job->early));
  store_aff(out_aff, jac_to_affine(xyzz_to_jac(acc)));
  return SP_OK;
}
void sp_hyrax_commit_split_drop(sp_split_commit* job) {
  if (!job) return;
  if (job->early) (void)sp::WalkPool::get().finish(job->early);
  delete job;
}


}  // extern "C"
