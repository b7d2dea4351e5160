// libspartan_hip.so — general Pippenger MSM over caller-supplied bases (kernels_pippenger.hpp): DlogGroupExt::vartime_multiscalar_mul /
// vartime_multiscalar_mul_small for n well beyond the Hyrax row width (src/provider/msm.rs:187-222, :367-409). Own translation unit: the latency
// kernels of capi_group.hip are built with a max-ILP scheduling flag that would leave the bucket kernel here at one wave per SIMD.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "group_common.hpp"
#include "kernels_pippenger.hpp"

using sp::fail;

namespace sp {

// Window width. The reference takes c = ceil(ln n) (msm.rs:201-205); here the trade is the same — additions per point fall with c (ceil(257 / c)
// windows), buckets per window and the window kernel's depth grow with it — but the constants are this kernel's: chosen by measurement on MI355X
// (profiles/r04_msm_big.txt).
int pippenger_window(size_t n) {
  if (n < ((size_t)1 << 17)) return 8;   // 2^16: 0.68 ms with 8 bits, 1.10 with 10, 1.26 with 12
  if (n < ((size_t)1 << 19)) return 12;  // 2^18: 2.06 ms with 12, 2.15 with 10 or 13
  if (n < ((size_t)1 << 23)) return 13;  // 2^20: 3.83 ms with 13, 3.93 with 12, 4.23 with 14; 2^22: 10.4 ms with 13 or 14, 11.5 with 12
  return 14;
}

// waves per SIMD k_pip_bucket_tasks is compiled for: 2 = 256 registers a lane, no spills; 3 = 168 registers, 54-66 spilled (tools/spill_report.py). SPARTAN_PIP_MINW
// picks (A/B: profiles/r06_spills.md)
#ifndef PIP_MINW_DEFAULT
#define PIP_MINW_DEFAULT 3
#endif
static int pip_minw() {
  static const int v = [] {
    const char* e = getenv("SPARTAN_PIP_MINW");
    return e && e[0] == '3' ? 3 : (e && e[0] == '2' ? 2 : PIP_MINW_DEFAULT);
  }();
  return v;
}
template <int C>
static int run_pippenger(sp_ctx* c, const fe_t* d_canon, const aff_t* d_bases, size_t n, bool full_width, jac_t* result) {
  hipStream_t st = c->stream;
  const int W = spk::pip_windows(C, full_width ? 256 : 64);
  const unsigned E = 1u << (C - 1);
  const size_t total = (size_t)W * E;
  short* digits = (short*)c->workspace(sp_ctx::WS_MSM_DIGITS, (size_t)W * n * sizeof(short));
  unsigned* order = (unsigned*)c->workspace(sp_ctx::WS_MSM_ORDER, (size_t)W * n * 4);
  // counts | cursor | start in one buffer
  unsigned* meta = (unsigned*)c->workspace(sp_ctx::WS_MSM_START, (total * 2 + (size_t)W * (E + 1)) * 4);
  xyzz_t* buckets = (xyzz_t*)c->workspace(sp_ctx::WS_MSM_BUCKETS, total * sizeof(xyzz_t));
  jac_t* wsum = (jac_t*)c->workspace(sp_ctx::WS_MSM_WSUM, (size_t)W * C * sizeof(jac_t));
  if (!digits || !order || !meta || !buckets || !wsum) return SP_ERR_NO_DEVICE;
  unsigned *counts = meta, *cursor = meta + total, *start = meta + 2 * total;
  SP_HIP(hipMemsetAsync(counts, 0, total * 4, st));
  const unsigned chunks = (unsigned)((n + spk::PIP_CHUNK - 1) / spk::PIP_CHUNK);
  c->timed("msm_big_sort", 32ull * n, [&] {
    if (full_width) hipLaunchKernelGGL((spk::k_pip_digits<C, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_canon, (unsigned)n, digits);
    else hipLaunchKernelGGL((spk::k_pip_digits<C, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_canon, (unsigned)n, digits);
    hipLaunchKernelGGL((spk::k_pip_hist<C>), dim3(chunks, W), dim3(256), 0, st, digits, (unsigned)n, counts);
    hipLaunchKernelGGL((spk::k_pip_scan<C>), dim3(W), dim3(256), 0, st, counts, start, cursor);
    hipLaunchKernelGGL((spk::k_pip_scatter<C>), dim3(chunks, W), dim3(256), 0, st, digits, (unsigned)n, cursor, order);
  });
  // Tasks: a bucket's list in pieces of at most `chunk` entries — twice the mean bucket size (at least 256), so a typical bucket is one task and only
  // the crowded ones (the top window's, repeated scalars) are cut up; LPB lanes per task with ~64 entries each, so the LPB-lane shuffle tree
  // (log2 LPB full additions) stays a few per cent of the task's mixed additions.
  const size_t avg = n / E + 1;
  unsigned chunk = 256;
  while (chunk < 2 * avg) chunk <<= 1;
  int lpb = 1;
  while (lpb < 64 && (size_t)lpb * 64 < chunk / 2) lpb <<= 1;
  const size_t max_tasks = total + (size_t)W * n / chunk + 1;  // non-empty buckets + full chunks
  while (lpb < 64 && max_tasks * lpb < ((size_t)1 << 17)) lpb <<= 1;  // ... but enough lanes to fill 256 CUs
  // task_first [total + 1] | task_bucket [max_tasks] | multi_list [total] | counts
  unsigned* tmeta = (unsigned*)c->workspace(sp_ctx::WS_MSM_TASKS, (total + 1 + max_tasks + total + 4) * 4);
  xyzz_t* partial = (xyzz_t*)c->workspace(sp_ctx::WS_MSM_PARTIAL, max_tasks * sizeof(xyzz_t));
  if (!tmeta || !partial) return SP_ERR_NO_DEVICE;
  unsigned *task_first = tmeta, *task_bucket = tmeta + total + 1, *multi_list = task_bucket + max_tasks;
  spk::PipTaskCounts* tcounts = reinterpret_cast<spk::PipTaskCounts*>(multi_list + total);
  SP_HIP(hipMemsetAsync(buckets, 0, total * sizeof(xyzz_t), st));
  const size_t lanes = max_tasks * lpb;
  const dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
  c->timed("msm_big_buckets", 96ull * n, [&] {
    hipLaunchKernelGGL(spk::k_pip_tasks_scan, dim3(1), dim3(1024), 0, st, start, E, total, chunk, task_first, tcounts);
    hipLaunchKernelGGL(spk::k_pip_tasks_fill, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, task_first, total, task_bucket, multi_list, tcounts);
#define PIP_TASKS(L, MW) \
  hipLaunchKernelGGL((spk::k_pip_bucket_tasks<L, MW>), grid, block, 0, st, d_bases, (unsigned)n, E, order, start, chunk, task_first, task_bucket, tcounts, buckets, partial)
    if (pip_minw() == 2) {
      switch (lpb) {
        case 1: PIP_TASKS(1, 2); break;
        case 2: PIP_TASKS(2, 2); break;
        case 4: PIP_TASKS(4, 2); break;
        case 8: PIP_TASKS(8, 2); break;
        case 16: PIP_TASKS(16, 2); break;
        case 32: PIP_TASKS(32, 2); break;
        default: PIP_TASKS(64, 2); break;
      }
    } else {
      switch (lpb) {
        case 1: PIP_TASKS(1, 3); break;
        case 2: PIP_TASKS(2, 3); break;
        case 4: PIP_TASKS(4, 3); break;
        case 8: PIP_TASKS(8, 3); break;
        case 16: PIP_TASKS(16, 3); break;
        case 32: PIP_TASKS(32, 3); break;
        default: PIP_TASKS(64, 3); break;
      }
    }
#undef PIP_TASKS
    const size_t max_multi = (size_t)W * n / chunk + 1;  // a bucket needs more than `chunk` entries to be cut
    hipLaunchKernelGGL(spk::k_pip_bucket_join, dim3((unsigned)((max_multi * 64 + 255) / 256)), dim3(256), 0, st, multi_list, task_first, tcounts, partial, buckets);
  });
  c->timed("msm_big_window", 0, [&] { hipLaunchKernelGGL(spk::k_pip_bitsums, dim3(C, W), dim3(256), 0, st, buckets, E, C, wsum); });
  std::vector<jac_t> ws((size_t)W * C);
  SP_HIP(hipMemcpyAsync(ws.data(), wsum, ws.size() * sizeof(jac_t), hipMemcpyDeviceToHost, st));
  SP_HIP(sp::stream_sync(st));
  // Horner over the windows and, inside a window, over the bits of the bucket weights, high to low (msm.rs:150-175 with the window sums bit-sliced):
  // acc = 2 acc + S[w][bit]
  jac_t acc = jac_identity();
  for (int w = W - 1; w >= 0; --w)
    for (int bit = C - 1; bit >= 0; --bit) {
      acc = jac_dbl(acc);
      acc = jac_add(acc, ws[(size_t)w * C + bit]);
    }
  *result = acc;
  return SP_OK;
}

// n canonical scalars and n affine bases already in HBM, on the context's main stream. full_width: scalars of the whole field (sign-folded here);
// otherwise values < 2^64 (msm_small's digit path, msm.rs:367-409). window = 0 picks pippenger_window(n).
int msm_pippenger(sp_ctx* c, const fe_t* d_canon, const aff_t* d_bases, size_t n, bool full_width, int window, jac_t* result) {
  *result = jac_identity();
  if (n == 0) return SP_OK;
  if (n >= ((size_t)1 << 31)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "msm: n too large");
  switch (window ? window : pippenger_window(n)) {
    case 8: return run_pippenger<8>(c, d_canon, d_bases, n, full_width, result);
    case 10: return run_pippenger<10>(c, d_canon, d_bases, n, full_width, result);
    case 12: return run_pippenger<12>(c, d_canon, d_bases, n, full_width, result);
    case 13: return run_pippenger<13>(c, d_canon, d_bases, n, full_width, result);
    case 14: return run_pippenger<14>(c, d_canon, d_bases, n, full_width, result);
    default: return fail(SP_ERR_INVALID_INPUT_LENGTH, "msm: window width must be 8, 10, 12, 13 or 14");
  }
}

}  // namespace sp

extern "C" int sp_msm_pippenger_window(size_t n) { return sp::pippenger_window(n); }
