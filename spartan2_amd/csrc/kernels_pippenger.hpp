// HIP kernels (gfx950) of the general Pippenger MSM for caller-supplied bases: DlogGroupExt::vartime_multiscalar_mul -> msm / cpu_msm_serial
// (src/provider/msm.rs:59-222) at sizes where one block per window (kernels_msm.hpp: the latency form of the 2048-wide Hyrax MSMs) would leave the
// chip idle. Same mathematics — signed C-bit digits (msm.rs:110-145), buckets per window, sum_k k * B_k by summation by parts (:150-175), window Horner —
// laid out for throughput:
//   k_pip_digits<C>    one lane per scalar: sign folding s -> min(s, order - s) (so the top window never carries), all signed digits in one pass,
//                      stored as int16 [window][point] with the point's effective negation folded into the digit's sign
//   k_pip_hist<C>      (chunk of points, window) blocks: LDS histogram of |digit|, flushed with one global atomic per non-empty bucket
//   k_pip_scan<C>      one block per window: exclusive prefix over the 2^(C-1) bucket counts
//   k_pip_scatter<C>   (chunk, window) blocks: LDS histogram again, ONE global atomic per non-empty bucket reserves the chunk's range in the bucket's
//                      list, LDS cursors place the chunk's points in it — a multi-block counting sort whose global atomics are per (block, bucket),
//                      not per point. The order inside a bucket depends on block scheduling; a bucket's SUM, a group element, does not.
//   k_pip_tasks_*      cut every bucket's list into tasks of at most `chunk` entries (bucket sizes are as uneven as the scalars make them)
//   k_pip_bucket_tasks<LPB>  LPB adjacent lanes per task stride through its entries with mixed XYZZ additions (madd-2008-s, 10 products; base gathers
//                      are random 64-byte reads, the next one issued before the addition that consumes the current one), shuffle tree over the LPB lanes
//   k_pip_bucket_join  one wave per bucket that was cut into several tasks: adds their partial sums
//   k_pip_bitsums      (bit, window) blocks: S_bit = plain tree sum of the buckets whose weight has that bit set; sum_k k B_k = sum_bit 2^bit S_bit
// The Horner over windows AND bits (256 doublings, one addition per bit) stays on the host side of the library: one CPU core is ~20x a GPU lane on a
// dependent chain.
// Bound: VALU (one mixed addition = 10 base-field products per (point, window) pair); algorithmic bytes per SURVEY 8(d): 96 B per (scalar, base) pair.
#pragma once
#include "curve.hpp"
#include "device_utils.hpp"

namespace spk {

typedef FqP PSF;  // scalar field

constexpr int pip_windows(int C, int bits) { return (bits + 1 + C - 1) / C; }  // bits = 256 (folded full-width scalars) or 64

// ---- digits --------------------------------------------------------------------------------------------------------------------------------
// digits[w * n + j] = signed digit of window w of scalar j, in [-(E-1), E], E = 2^(C-1); sign already includes the fold (s -> order - s <=> P -> -P).
// FOLD = full-width scalars (canonical limbs): folded values are < 2^255, so ceil(257 / C) windows hold every carry. !FOLD = values < 2^64.
template <int C, bool FOLD>
__global__ void __launch_bounds__(256) k_pip_digits(const fe_t* __restrict__ canon, unsigned n, short* __restrict__ digits) {
  constexpr int W = pip_windows(C, FOLD ? 256 : 64);
  constexpr int E = 1 << (C - 1);
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  fe_t c = canon[j];
  bool neg = false;
  if (FOLD) {
    fe_t d;
    uint32_t bw = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d.v[k] = sp_subb(PSF::P(k), c.v[k], bw);  // order - c
    uint32_t lt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) (void)sp_subb(d.v[k], c.v[k], lt);  // lt == 1 iff order - c < c
    if (lt && !fe_is_zero(c)) {
      c = d;
      neg = true;
    }
  }
  int carry = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const int pos = C * w, idx = pos >> 5, sh = pos & 31;
    unsigned raw = 0;
    if (idx < (FOLD ? 8 : 2)) {
      raw = c.v[idx] >> sh;
      if (sh + C > 32 && idx + 1 < (FOLD ? 8 : 2)) raw |= c.v[idx + 1] << (32 - sh);
      raw &= (1u << C) - 1u;
    }
    int r = (int)raw + carry;
    if (r > E) {
      r -= 1 << C;
      carry = 1;
    } else {
      carry = 0;
    }
    digits[(size_t)w * n + j] = (short)(neg ? -r : r);
  }
}

// ---- multi-block counting sort per window ------------------------------------------------------------------------------------------------
constexpr unsigned PIP_CHUNK = 8192;  // points per (chunk, window) block

template <int C>
__global__ void __launch_bounds__(256) k_pip_hist(const short* __restrict__ digits, unsigned n, unsigned* __restrict__ counts /* [W][E] */) {
  constexpr unsigned E = 1u << (C - 1);
  __shared__ unsigned h[E];
  const unsigned w = blockIdx.y, lo = blockIdx.x * PIP_CHUNK, hi = lo + PIP_CHUNK < n ? lo + PIP_CHUNK : n;
  for (unsigned k = threadIdx.x; k < E; k += 256) h[k] = 0;
  __syncthreads();
  const short* dg = digits + (size_t)w * n;
  for (unsigned j = lo + threadIdx.x; j < hi; j += 256) {
    const int d = dg[j];
    if (d) atomicAdd(&h[(unsigned)(d < 0 ? -d : d) - 1u], 1u);
  }
  __syncthreads();
  unsigned* g = counts + (size_t)w * E;
  for (unsigned k = threadIdx.x; k < E; k += 256)
    if (h[k]) atomicAdd(&g[k], h[k]);
}
// start[w][k] = first position of bucket k's list inside window w's segment of `order`; start[w][E] = the window's total; cursor = a working copy
template <int C>
__global__ void __launch_bounds__(256) k_pip_scan(const unsigned* __restrict__ counts, unsigned* __restrict__ start /* [W][E + 1] */, unsigned* __restrict__ cursor /* [W][E] */) {
  constexpr unsigned E = 1u << (C - 1), PER = (E + 255) / 256;
  __shared__ unsigned part[256];
  const unsigned w = blockIdx.x;
  const unsigned* c = counts + (size_t)w * E;
  unsigned local[PER], sum = 0;
#pragma unroll
  for (unsigned i = 0; i < PER; ++i) {
    const unsigned k = threadIdx.x * PER + i;
    local[i] = k < E ? c[k] : 0u;
    sum += local[i];
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (unsigned off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan over the 256 partial sums
    const unsigned v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = part[threadIdx.x] - sum;
#pragma unroll
  for (unsigned i = 0; i < PER; ++i) {
    const unsigned k = threadIdx.x * PER + i;
    if (k < E) {
      start[(size_t)w * (E + 1) + k] = run;
      cursor[(size_t)w * E + k] = run;
      run += local[i];
    }
  }
  if (threadIdx.x == 255) start[(size_t)w * (E + 1) + E] = part[255];
}
template <int C>
__global__ void __launch_bounds__(256) k_pip_scatter(const short* __restrict__ digits, unsigned n, unsigned* __restrict__ cursor /* [W][E] */,
                                                     unsigned* __restrict__ order /* [W][n] */) {
  constexpr unsigned E = 1u << (C - 1);
  __shared__ unsigned h[E];  // counts, then the chunk's base position per bucket, then a running cursor
  const unsigned w = blockIdx.y, lo = blockIdx.x * PIP_CHUNK, hi = lo + PIP_CHUNK < n ? lo + PIP_CHUNK : n;
  for (unsigned k = threadIdx.x; k < E; k += 256) h[k] = 0;
  __syncthreads();
  const short* dg = digits + (size_t)w * n;
  for (unsigned j = lo + threadIdx.x; j < hi; j += 256) {
    const int d = dg[j];
    if (d) atomicAdd(&h[(unsigned)(d < 0 ? -d : d) - 1u], 1u);
  }
  __syncthreads();
  unsigned* g = cursor + (size_t)w * E;
  for (unsigned k = threadIdx.x; k < E; k += 256) {
    const unsigned cnt = h[k];
    h[k] = cnt ? atomicAdd(&g[k], cnt) : 0u;
  }
  __syncthreads();
  unsigned* ord = order + (size_t)w * n;
  for (unsigned j = lo + threadIdx.x; j < hi; j += 256) {
    const int d = dg[j];
    if (d) {
      const unsigned pos = atomicAdd(&h[(unsigned)(d < 0 ? -d : d) - 1u], 1u);
      ord[pos] = j | (d < 0 ? 0x80000000u : 0u);
    }
  }
}

// ---- bucket sums ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ xyzz_t pip_shfl_down(const xyzz_t& a, int delta) {
  xyzz_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.x.v[i] = __shfl_down(a.x.v[i], delta, 64);
    r.y.v[i] = __shfl_down(a.y.v[i], delta, 64);
    r.zz.v[i] = __shfl_down(a.zz.v[i], delta, 64);
    r.zzz.v[i] = __shfl_down(a.zzz.v[i], delta, 64);
  }
  return r;
}
// Bucket sizes are as uneven as the scalars make them — the top window of a 255-bit scalar has only 2^(255 mod C) buckets in use, each holding
// n / 2^(255 mod C) points, and a caller's scalars may repeat — so the unit of work is not a bucket but a TASK: at most `chunk` consecutive entries of
// one bucket's list. k_pip_tasks_scan counts ceil(size / chunk) tasks per bucket and scans the counts (one block: the bucket count is <= 2^18),
// k_pip_tasks_fill writes the task -> bucket map and lists the buckets that got more than one task, k_pip_bucket_tasks<LPB> runs the tasks (LPB
// adjacent lanes each), k_pip_bucket_join adds the partial sums of the multi-task buckets (one wave per bucket). Grids are sized for the worst case
// (no device -> host round trip for the counts); surplus blocks read the real count and leave.
struct PipTaskCounts {
  unsigned tasks, multi;
};
__global__ void __launch_bounds__(1024) k_pip_tasks_scan(const unsigned* __restrict__ start, unsigned E, size_t total_buckets, unsigned chunk,
                                                         unsigned* __restrict__ task_first /* [total + 1] */, PipTaskCounts* __restrict__ counts) {
  __shared__ unsigned part[1024];
  const size_t per = (total_buckets + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < total_buckets ? lo + per : total_buckets;
  unsigned sum = 0;
  for (size_t b = lo; b < hi; ++b) {
    const size_t w = b / E, k = b % E;
    const unsigned sz = start[w * (E + 1) + k + 1] - start[w * (E + 1) + k];
    sum += (sz + chunk - 1) / chunk;
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (unsigned off = 1; off < 1024; off <<= 1) {
    const unsigned v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = part[threadIdx.x] - sum;
  for (size_t b = lo; b < hi; ++b) {
    const size_t w = b / E, k = b % E;
    const unsigned sz = start[w * (E + 1) + k + 1] - start[w * (E + 1) + k];
    task_first[b] = run;
    run += (sz + chunk - 1) / chunk;
  }
  if (threadIdx.x == 1023) {
    task_first[total_buckets] = part[1023];
    counts->tasks = part[1023];
    counts->multi = 0;
  }
}
__global__ void __launch_bounds__(256) k_pip_tasks_fill(const unsigned* __restrict__ task_first, size_t total_buckets, unsigned* __restrict__ task_bucket,
                                                        unsigned* __restrict__ multi_list, PipTaskCounts* __restrict__ counts) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total_buckets) return;
  const unsigned lo = task_first[b], hi = task_first[b + 1];
  for (unsigned t = lo; t < hi; ++t) task_bucket[t] = (unsigned)b;
  if (hi - lo > 1) multi_list[atomicAdd(&counts->multi, 1u)] = (unsigned)b;
}
// buckets[] must be zero (= the identity in XYZZ form) on entry: empty buckets are never written
template <int LPB, int MINW>
__global__ void __launch_bounds__(256, MINW) k_pip_bucket_tasks(const aff_t* __restrict__ bases, unsigned n, unsigned E, const unsigned* __restrict__ order,
                                                             const unsigned* __restrict__ start, unsigned chunk, const unsigned* __restrict__ task_first,
                                                             const unsigned* __restrict__ task_bucket, const PipTaskCounts* __restrict__ counts,
                                                             xyzz_t* __restrict__ buckets, xyzz_t* __restrict__ partial) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t task = gid / LPB;
  const unsigned sub = (unsigned)(gid % LPB);
  const bool live = task < counts->tasks;
  if (!live) return;  // (the LPB lanes of a task leave or stay together, so the shuffles below only ever pair lanes that stayed)
  xyzz_t acc = xyzz_identity();
  unsigned b = 0;
  bool single = false;
  if (live) {
    b = task_bucket[task];
    const unsigned t0 = task_first[b];
    single = task_first[b + 1] - t0 == 1;
    const size_t w = b / E, k = b % E;
    const unsigned blo = start[w * (E + 1) + k], bhi = start[w * (E + 1) + k + 1];
    const unsigned lo = blo + (unsigned)(task - t0) * chunk, hi = lo + chunk < bhi ? lo + chunk : bhi;
    const unsigned* ord = order + w * n;
    unsigned p = lo + sub;
    unsigned e_next = p < hi ? ord[p] : 0u;
    aff_t q_next;
    if (p < hi) q_next = bases[e_next & 0x7fffffffu];
#pragma unroll 1
    for (; p < hi; p += LPB) {
      const unsigned e = e_next;
      aff_t q = q_next;
      if (p + LPB < hi) {  // the next base's gather is in flight under this addition
        e_next = ord[p + LPB];
        q_next = bases[e_next & 0x7fffffffu];
      }
      if (e & 0x80000000u) q.y = fe_neg<B>(q.y);
      acc = xyzz_add_mixed(acc, q);
    }
  }
#pragma unroll
  for (int d = LPB / 2; d >= 1; d >>= 1) {
    const xyzz_t o = pip_shfl_down(acc, d);
    if (sub < (unsigned)d) acc = xyzz_add(acc, o);
  }
  if (live && sub == 0) (single ? buckets[b] : partial[task]) = acc;
}
__global__ void __launch_bounds__(256) k_pip_bucket_join(const unsigned* __restrict__ multi_list, const unsigned* __restrict__ task_first,
                                                         const PipTaskCounts* __restrict__ counts, const xyzz_t* __restrict__ partial, xyzz_t* __restrict__ buckets) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  if (wave >= counts->multi) return;
  const unsigned b = multi_list[wave], lo = task_first[b], hi = task_first[b + 1];
  xyzz_t acc = xyzz_identity();
  for (unsigned t = lo + lane; t < hi; t += 64) acc = xyzz_add(acc, partial[t]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const xyzz_t o = pip_shfl_down(acc, d);
    if (lane < (unsigned)d) acc = xyzz_add(acc, o);
  }
  if (lane == 0) buckets[b] = acc;
}

// ---- window sums: sum_{k >= 1} k * B_k with B_k = buckets[k - 1], bit-sliced --------------------------------------------------------------------
// sum_k k B_k = sum_bit 2^bit S_bit with S_bit = the plain sum of the buckets whose weight k has that bit set: C independent tree sums per window
// (grid = C x windows blocks, each a chain of E / 512 + 8 additions) instead of the running-sum recurrence's 2 E dependent additions (msm.rs:169-174) or
// a weighted tree with its doublings. The powers of two cost nothing extra: the host's Horner over the windows already doubles once per bit, it now
// adds one S per bit instead of one window sum per C bits (256 additions instead of 256 / C).
__global__ void __launch_bounds__(256) k_pip_bitsums(const xyzz_t* __restrict__ buckets, unsigned E, int C, jac_t* __restrict__ sums /* [W][C] */) {
  __shared__ xyzz_t sh[4];
  const unsigned bit = blockIdx.x, w = blockIdx.y;
  const xyzz_t* b = buckets + (size_t)w * E;
  // the j-th weight with `bit` set: insert a 1 at position `bit` into j (j < E / 2); weight E itself (only bit C - 1) is j = 0 of the top bit
  xyzz_t acc = xyzz_identity();
  if (bit == (unsigned)C - 1) {
    if (threadIdx.x == 0) acc = b[E - 1];
  } else {
    const unsigned lowmask = (1u << bit) - 1u;
    for (unsigned j = threadIdx.x; j < E / 2; j += 256) {
      const unsigned k = ((j & ~lowmask) << 1) | (1u << bit) | (j & lowmask);  // 1 <= k < E
      acc = xyzz_add(acc, b[k - 1]);
    }
  }
#pragma unroll 1
  for (int d = 32; d >= 1; d >>= 1) {
    const xyzz_t o = pip_shfl_down(acc, d);
    if ((threadIdx.x & 63) < (unsigned)d) acc = xyzz_add(acc, o);
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    acc = xyzz_add(xyzz_add(sh[0], sh[1]), xyzz_add(sh[2], sh[3]));
    sums[(size_t)w * C + bit] = xyzz_to_jac(acc);
  }
}

}  // namespace spk
