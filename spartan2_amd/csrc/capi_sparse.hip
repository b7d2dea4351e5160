// libspartan_hip.so — R1CS sparse kernels and entry points of include/spartan_hip.h.
//   k_spmv3     K5/K6  PrecomputedSparseMatrix::multiply_vec (src/r1cs/sparse.rs:194-233) for A, B, C in one launch, and
//                      multiply_vec_incremental_into (src/r1cs/mod.rs:1170-1211) = cached + filtered rows (sparse.rs:305-380)
//   k_polyabc*  K7     bind_and_prepare_poly_ABC / accumulate_rows (src/r1cs/mod.rs:1235-1398) as a column-major GATHER
//                      (no 256-bit atomics): short columns one lane each, long columns (the constant-1 column of the
//                      booleanity rows) one block each.
// Entries keep the reference's classes: +-1 and |k| in 2..7 as an int8 code (add / sub / double-add chains, sparse.rs:137-155),
// everything else as a full field coefficient.
#include <algorithm>
#include <cstring>
#include <vector>

#include "core.hpp"
#include "device_utils.hpp"

using sp::fail;
typedef FqP S;

namespace spk {

struct SplitDev {          // major-ordered sparse structure, entries split by coefficient class
  const unsigned* sptr;    // [n_major + 1] small-class offsets
  const unsigned* sidx;    // minor index of each small-class entry
  const signed char* scode;  // +-1 .. +-7
  const unsigned* gptr;    // [n_major + 1] general-class offsets
  const unsigned* gidx;
  const fe_t* gval;
};

__device__ __forceinline__ fe_t small_mul(int code, const fe_t& x) {  // sparse.rs:137-155
  const int a = code < 0 ? -code : code;
  fe_t r;
  switch (a) {
    case 1: r = x; break;
    case 2: r = fe_dbl<S>(x); break;
    case 3: r = fe_add<S>(fe_dbl<S>(x), x); break;
    case 4: r = fe_dbl<S>(fe_dbl<S>(x)); break;
    case 5: r = fe_add<S>(fe_dbl<S>(fe_dbl<S>(x)), x); break;
    case 6: { fe_t d = fe_dbl<S>(x); r = fe_add<S>(fe_dbl<S>(d), d); break; }
    default: { fe_t d = fe_dbl<S>(x); r = fe_add<S>(fe_add<S>(fe_dbl<S>(d), d), x); break; }
  }
  return r;
}
__device__ __forceinline__ fe_t acc_small(const fe_t& acc, int code, const fe_t& x) {
  if (code == 1) return fe_add<S>(acc, x);
  if (code == -1) return fe_sub<S>(acc, x);
  fe_t m = small_mul(code, x);
  return code < 0 ? fe_sub<S>(acc, m) : fe_add<S>(acc, m);
}
// sum over the entries of one major index, strided (first, step) so a block can share a long list
__device__ __forceinline__ fe_t gather_major(const SplitDev& m, size_t major, const fe_t* __restrict__ x, unsigned first, unsigned step) {
  fe_t acc = fe_zero();
  for (unsigned k = m.sptr[major] + first, e = m.sptr[major + 1]; k < e; k += step) acc = acc_small(acc, m.scode[k], x[m.sidx[k]]);
  for (unsigned k = m.gptr[major] + first, e = m.gptr[major + 1]; k < e; k += step) acc = fe_add<S>(acc, fe_mul<S>(m.gval[k], x[m.gidx[k]]));
  return acc;
}

// The same sum with FOUR entries in flight: index / code loads first, then the four gathers, then the four accumulations. One lane walking a
// column otherwise pays two dependent memory latencies (index, then x[index]) per entry, and the column kernels are bound by exactly that chain
// (k_polyabc_short: 82 us for 5 M entries at config 2, nowhere near a bandwidth limit).
__device__ __forceinline__ fe_t gather_major_x4(const SplitDev& m, size_t major, const fe_t* __restrict__ x) {
  fe_t acc = fe_zero();
  unsigned k = m.sptr[major];
  const unsigned e = m.sptr[major + 1];
  for (; k + 4 <= e; k += 4) {
    const unsigned i0 = m.sidx[k], i1 = m.sidx[k + 1], i2 = m.sidx[k + 2], i3 = m.sidx[k + 3];
    const int c0 = m.scode[k], c1 = m.scode[k + 1], c2 = m.scode[k + 2], c3 = m.scode[k + 3];
    const fe_t x0 = x[i0], x1 = x[i1], x2 = x[i2], x3 = x[i3];
    acc = acc_small(acc, c0, x0);
    acc = acc_small(acc, c1, x1);
    acc = acc_small(acc, c2, x2);
    acc = acc_small(acc, c3, x3);
  }
  if (k + 2 <= e) {
    const unsigned i0 = m.sidx[k], i1 = m.sidx[k + 1];
    const int c0 = m.scode[k], c1 = m.scode[k + 1];
    const fe_t x0 = x[i0], x1 = x[i1];
    acc = acc_small(acc, c0, x0);
    acc = acc_small(acc, c1, x1);
    k += 2;
  }
  if (k < e) acc = acc_small(acc, m.scode[k], x[m.sidx[k]]);
  for (unsigned g = m.gptr[major], ge = m.gptr[major + 1]; g < ge; ++g) acc = fe_add<S>(acc, fe_mul<S>(m.gval[g], x[m.gidx[g]]));
  return acc;
}

struct Spmv3Args {
  SplitDev m[3];
  const fe_t* base[3];  // cached products to add (nullptr for a plain multiply_vec)
  fe_t* out[3];
};
// One lane per row, four entries in flight (gather_major_x4). A row longer than SPMV_LONG_ROW entries - the 32-bit additions of a SHA-256 round: ~6000 rows of
// 33 .. 225 entries among 880 K of one to three at config 2 - would keep its wave waiting on one lane's chain of dependent (index, then element) loads
// (the per-wave longest rows of A sum to 958 K such steps against 2 M entries in all: 0.75 ms for the cached product of prep_prove): those are walked by
// the whole wave, an entry per lane, and added with a shuffle tree; the owner lane keeps the sum.
// LONG_ROWS = false (no row of any of the three structures is longer than SPMV_LONG_ROW - the filtered rows of the incremental product inside a prove): the
// plain lane-per-row walk at 64 registers; the cooperative form needs 116 and halves the waves per SIMD of what is then a streaming kernel (62 vs 42 us).
constexpr unsigned SPMV_LONG_ROW = 24;
template <bool LONG_ROWS>
__global__ void __launch_bounds__(256) k_spmv3(Spmv3Args a, const fe_t* __restrict__ z, size_t nrows) {
  const int which = blockIdx.y;
  const SplitDev m = a.m[which];
  const fe_t* base = a.base[which];
  fe_t* out = a.out[which];
  if (!LONG_ROWS) {
    for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += (size_t)gridDim.x * blockDim.x) {
      fe_t acc = gather_major(m, row, z, 0, 1);
      if (base) acc = fe_add<S>(acc, base[row]);
      out[row] = acc;
    }
    return;
  }
  const unsigned lane = threadIdx.x & 63u;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t wbase = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); wbase < nrows; wbase += stride) {  // (uniform over a wave)
    const size_t row = wbase + lane;
    const bool valid = row < nrows;
    bool is_long = false;
    fe_t acc = fe_zero();
    if (valid) {
      is_long = (m.sptr[row + 1] - m.sptr[row]) + (m.gptr[row + 1] - m.gptr[row]) > SPMV_LONG_ROW;
      if (!is_long) acc = gather_major_x4(m, row, z);
    }
    unsigned long long pending = __ballot(is_long);
    while (pending) {
      const int owner = __ffsll((long long)pending) - 1;
      pending &= pending - 1;
      const size_t r = wbase + (size_t)owner;
      const fe_t part = wave_sum(gather_major(m, r, z, lane, 64));
      if ((int)lane == owner) acc = part;
    }
    if (valid) {
      if (base) acc = fe_add<S>(acc, base[row]);
      out[row] = acc;
    }
  }
}

// The tau-independent halves of the outer sum-check's FIRST evaluation (evaluation_points_cubic_with_three_inputs, src/sumcheck.rs:1041-1105), formed
// right behind the matrix-vector product while both run in the shadow of commit_zeros: P0[i] = A0 B0 - C0, P1[i] = (A1 - A0)(B1 - B0) for the pair
// (i, i + N/2) the top variable joins (src/polys/multilinear.rs:101). The first evaluation on the critical path then reads 2 x 16 MiB instead of
// 84 MiB. (A fused form — one thread computing both rows of all three products — was measured: its six row walks per thread cost more than this
// second streaming pass, 116 us against 50 + 25 us.)
__global__ void __launch_bounds__(256) k_round0_products(const fe_t* __restrict__ A, const fe_t* __restrict__ B, const fe_t* __restrict__ C, size_t half,
                                                         fe_t* __restrict__ p0, fe_t* __restrict__ p1) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    const fe_t a0 = A[i], a1 = A[i + half], b0 = B[i], b1 = B[i + half], c0 = C[i];
    p0[i] = fe_sub<S>(fe_mul<S>(a0, b0), c0);
    p1[i] = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  }
}

struct PolyAbcArgs {
  SplitDev m[3];  // column-major A, B, C
  fe_t r, r2;
};
constexpr unsigned LONG_COLUMN = 512;
__device__ __forceinline__ unsigned col_len(const PolyAbcArgs& a, size_t col) {
  unsigned n = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) n += (a.m[i].sptr[col + 1] - a.m[i].sptr[col]) + (a.m[i].gptr[col + 1] - a.m[i].gptr[col]);
  return n;
}
// Long columns (the constant-1 column has ~one entry per booleanity row): NB blocks share a column, each striding
// over its entry lists; the column's last block to arrive adds the NB partial triples and applies (1, r, r^2).
constexpr unsigned LONG_NB_MAX = 128;
__device__ __forceinline__ unsigned long_nb(unsigned len) {
  unsigned nb = (len + 2047) / 2048;
  return nb < 1 ? 1 : (nb > LONG_NB_MAX ? LONG_NB_MAX : nb);
}
// accumulate_rows (src/r1cs/mod.rs:1324-1398) as a column-major gather in ONE launch: blocks [0, LONG_NB_MAX * n_long) are the long columns' (they are
// dispatched first and run under the short columns' blocks, off the path of the inner sum-check's first round), the next `short_blocks` walk the short
// columns — `order` lists them by decreasing entry count and, within one count, by their (A, B, C) entry counts, so the 64 columns of a wave run the same
// trip counts —, the last blocks of the grid write the output's zero tail. The last block of a long column to arrive (one counter per column, <= 128
// arrivals, partial triples stored write-through and read back past this XCD's L2) adds the column's partials and applies (1, r, r^2).
__global__ void __launch_bounds__(256) k_polyabc_short_and_long(PolyAbcArgs a, const fe_t* __restrict__ rx, const unsigned* __restrict__ order, size_t n_short,
                                                                fe_t* __restrict__ out, const unsigned* __restrict__ long_cols, unsigned n_long,
                                                                fe_t* __restrict__ partials, unsigned* __restrict__ tickets, unsigned short_blocks,
                                                                size_t zero_from, size_t zero_n, int permuted) {
  __shared__ fe_t smem[3 * 4];
  __shared__ unsigned s_last;
  const unsigned long_blocks = LONG_NB_MAX * n_long;
  if (blockIdx.x >= long_blocks + short_blocks) {
    // the zero tail of the output (out_len > num_cols: 32 MB at config 2, a 7 us fill launch in front of this kernel until round 3): the last blocks
    // of the grid stream it out under the column walks
    uint4* z = reinterpret_cast<uint4*>(out + zero_from);
    const size_t n16 = zero_n * 2, nb = gridDim.x - long_blocks - short_blocks;
    for (size_t i = (size_t)(blockIdx.x - long_blocks - short_blocks) * blockDim.x + threadIdx.x; i < n16; i += nb * blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  if (blockIdx.x < long_blocks) {
    const unsigned by = blockIdx.x / LONG_NB_MAX, bx = blockIdx.x % LONG_NB_MAX;
    const size_t col = long_cols[by];                       // the column's number: where its sum goes
    const size_t at = permuted ? n_short + by : col;        // ... and where the structure keeps it (walk order: the long columns behind the short ones)
    const unsigned nb = long_nb(col_len(a, at));
    if (bx >= nb) return;
    fe_t acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = gather_major(a.m[i], at, rx, bx * blockDim.x + threadIdx.x, nb * blockDim.x);
    block_sum<3>(acc, smem);
    fe_t* trip = partials + ((size_t)by * LONG_NB_MAX) * 3;
    if (threadIdx.x == 0) {
      unsigned* dst = reinterpret_cast<unsigned*>(trip + (size_t)bx * 3);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int w = 0; w < 8; ++w) __hip_atomic_store(dst + 8 * i + w, acc[i].v[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // release / acquire at agent scope on the ticket (a few hundred long-column blocks at most, all of them under the short columns' walk: the
      // L2 write-back a release costs does not matter here as it did in a thousand-block streaming launch)
      s_last = __hip_atomic_fetch_add(tickets + by, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nb - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      acc[i] = fe_zero();
      if (threadIdx.x < nb) {
        const unsigned* src = reinterpret_cast<const unsigned*>(trip + (size_t)threadIdx.x * 3 + i);
#pragma unroll
        for (int w = 0; w < 8; ++w) acc[i].v[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();  // smem reuse
    block_sum<3>(acc, smem);
    if (threadIdx.x == 0) {
      __hip_atomic_store(tickets + by, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      out[col] = fe_add<S>(fe_add<S>(acc[0], fe_mul<S>(a.r, acc[1])), fe_mul<S>(a.r2, acc[2]));
    }
    return;
  }
  const size_t nblk = short_blocks;
  for (size_t i = (size_t)(blockIdx.x - long_blocks) * blockDim.x + threadIdx.x; i < n_short; i += nblk * blockDim.x) {
    const size_t col = order[i], at = permuted ? i : col;
    fe_t sa = gather_major_x4(a.m[0], at, rx), sb = gather_major_x4(a.m[1], at, rx), sc = gather_major_x4(a.m[2], at, rx);
    if (!fe_is_zero(sb)) sa = fe_add<S>(sa, fe_mul<S>(a.r, sb));
    if (!fe_is_zero(sc)) sa = fe_add<S>(sa, fe_mul<S>(a.r2, sc));
    out[col] = sa;
  }
}
}  // namespace spk

// ---- host side: classification and upload ---------------------------------------------------------------------------
namespace {

struct SplitHost {
  std::vector<unsigned> sptr, sidx, gptr, gidx;
  std::vector<signed char> scode;
  std::vector<fe_t> gval;
};
struct SplitOnDevice {
  unsigned *sptr = nullptr, *sidx = nullptr, *gptr = nullptr, *gidx = nullptr;
  signed char* scode = nullptr;
  fe_t* gval = nullptr;
  unsigned max_len = 0;  // longest major (entries of both classes): picks k_spmv3's form
  spk::SplitDev view() const { return spk::SplitDev{sptr, sidx, scode, gptr, gidx, gval}; }
  void release() {
    hipFree(sptr);
    hipFree(sidx);
    hipFree(gptr);
    hipFree(gidx);
    hipFree(scode);
    hipFree(gval);
  }
};

struct Classifier {  // from_sparse (sparse.rs:49-134)
  fe_t pos[8], neg[8];
  Classifier() {
    for (int k = 1; k <= 7; ++k) {
      pos[k] = fe_from_u64<S>(k);
      neg[k] = fe_neg<S>(pos[k]);
    }
  }
  int code(const fe_t& v) const {
    for (int k = 1; k <= 7; ++k) {
      if (fe_eq(v, pos[k])) return k;
      if (fe_eq(v, neg[k])) return -k;
    }
    return 0;
  }
};

template <class T>
int upload(T** dst, const std::vector<T>& src) {
  size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
  hipError_t e = hipMalloc((void**)dst, bytes);
  if (e != hipSuccess) return fail(SP_ERR_NO_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  if (!src.empty()) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(SP_ERR_NO_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
  return SP_OK;
}
int upload_split(const SplitHost& h, SplitOnDevice* d) {
  int rc;
  if ((rc = upload(&d->sptr, h.sptr))) return rc;
  if ((rc = upload(&d->sidx, h.sidx))) return rc;
  if ((rc = upload(&d->scode, h.scode))) return rc;
  if ((rc = upload(&d->gptr, h.gptr))) return rc;
  if ((rc = upload(&d->gidx, h.gidx))) return rc;
  d->max_len = 0;
  for (size_t i = 0; i + 1 < h.sptr.size(); ++i) {
    const unsigned l = (h.sptr[i + 1] - h.sptr[i]) + (h.gptr[i + 1] - h.gptr[i]);
    if (l > d->max_len) d->max_len = l;
  }
  return upload(&d->gval, h.gval);
}

struct Entry {
  unsigned major, minor;
  int code;
  fe_t val;
};
// build a major-ordered split structure from (major, minor) entries already grouped by major
SplitHost build_split(size_t n_major, const std::vector<Entry>& es) {
  SplitHost h;
  h.sptr.assign(n_major + 1, 0);
  h.gptr.assign(n_major + 1, 0);
  for (const Entry& e : es) (e.code ? h.sptr : h.gptr)[e.major + 1]++;
  for (size_t i = 0; i < n_major; ++i) {
    h.sptr[i + 1] += h.sptr[i];
    h.gptr[i + 1] += h.gptr[i];
  }
  h.sidx.resize(h.sptr[n_major]);
  h.scode.resize(h.sptr[n_major]);
  h.gidx.resize(h.gptr[n_major]);
  h.gval.resize(h.gptr[n_major]);
  std::vector<unsigned> sc(h.sptr.begin(), h.sptr.end() - 1), gc(h.gptr.begin(), h.gptr.end() - 1);
  for (const Entry& e : es) {
    if (e.code) {
      unsigned p = sc[e.major]++;
      h.sidx[p] = e.minor;
      h.scode[p] = (signed char)e.code;
    } else {
      unsigned p = gc[e.major]++;
      h.gidx[p] = e.minor;
      h.gval[p] = e.val;
    }
  }
  return h;
}

}  // namespace

struct sp_shape {
  sp_dims dims;
  size_t num_vars = 0, num_cols = 0;
  SplitOnDevice row[3];       // full CSR (multiply_vec)
  SplitOnDevice filtered[3];  // FilteredSpmv rows: col >= num_shared + num_precommitted, row < num_cons_unpadded
  SplitOnDevice col[3];       // column-major, rows < num_cons_unpadded (accumulate_rows)
  unsigned* d_long_cols = nullptr;
  unsigned* d_short_order = nullptr;  // short columns by decreasing entry count (k_polyabc_short_and_long)
  size_t n_short = 0;
  size_t n_long_cols = 0;
  bool col_permuted = false;  // col[] is stored in walk order: position i = d_short_order[i] for i < n_short, then the long columns
  uint64_t nnz[3] = {0, 0, 0}, nnz_filtered[3] = {0, 0, 0};
};

extern "C" {

int sp_shape_from_csr(sp_ctx* c, const sp_csr* A, const sp_csr* Bm, const sp_csr* C, const sp_dims* dims, sp_shape** out) {
  (void)c;
  sp_shape* s = new sp_shape();
  s->dims = *dims;
  s->num_vars = dims->num_shared + dims->num_precommitted + dims->num_rest;
  s->num_cols = s->num_vars + 1 + dims->num_public + dims->num_challenges;
  const size_t nrows = dims->num_cons, col_min = dims->num_shared + dims->num_precommitted, nr_used = dims->num_cons_unpadded;
  const sp_csr* M[3] = {A, Bm, C};
  Classifier cls;
  std::vector<unsigned> col_count(s->num_cols, 0);
  std::vector<SplitHost> col_host(3);
  for (int m = 0; m < 3; ++m) {
    const size_t nnz = M[m]->indptr[nrows];
    s->nnz[m] = nnz;
    std::vector<Entry> all, filt, bycol;
    all.reserve(nnz);
    for (size_t r = 0; r < nrows; ++r) {
      for (uint64_t k = M[m]->indptr[r]; k < M[m]->indptr[r + 1]; ++k) {
        Entry e;
        e.major = (unsigned)r;
        e.minor = M[m]->indices[k];
        if (e.minor >= s->num_cols) {
          delete s;
          return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_shape_from_csr: column index out of range");
        }
        memcpy(&e.val, M[m]->data + 4 * k, 32);
        e.code = cls.code(e.val);
        all.push_back(e);
        if (r < nr_used && e.minor >= col_min) filt.push_back(e);
      }
    }
    s->nnz_filtered[m] = filt.size();
    // column-major copy of the rows accumulate_rows visits
    std::vector<unsigned> cptr(s->num_cols + 1, 0);
    for (const Entry& e : all)
      if (e.major < nr_used) cptr[e.minor + 1]++;
    for (size_t i = 0; i < s->num_cols; ++i) cptr[i + 1] += cptr[i];
    bycol.resize(cptr[s->num_cols]);
    {
      std::vector<unsigned> cur(cptr.begin(), cptr.end() - 1);
      for (const Entry& e : all)
        if (e.major < nr_used) {
          Entry t = e;
          t.major = e.minor;
          t.minor = e.major;
          bycol[cur[e.minor]++] = t;
        }
    }
    for (size_t i = 0; i < s->num_cols; ++i) col_count[i] += cptr[i + 1] - cptr[i];
    int rc;
    col_host[m] = build_split(s->num_cols, bycol);
    if ((rc = upload_split(build_split(nrows, all), &s->row[m])) || (rc = upload_split(build_split(nrows, filt), &s->filtered[m]))) {
      delete s;
      return rc;
    }
  }
  std::vector<unsigned> long_cols;
  for (size_t i = 0; i < s->num_cols; ++i)
    if (col_count[i] >= spk::LONG_COLUMN) long_cols.push_back((unsigned)i);
  s->n_long_cols = long_cols.size();
  int rc = upload(&s->d_long_cols, long_cols);
  if (rc) {
    sp_shape_free(s);
    return rc;
  }
  {
    std::vector<unsigned> order;
    order.reserve(s->num_cols);
    for (size_t i = 0; i < s->num_cols; ++i)
      if (col_count[i] < spk::LONG_COLUMN) order.push_back((unsigned)i);
    // Sorted by decreasing entry count; within one total the columns are grouped by their (A, B, C) entry counts, so the lanes of a wave run the same
    // trip counts in each of the three matrices. (Sorting inside windows of neighbouring columns instead — for locality of the pointer loads and of the
    // gathers from evals_rx — measured the same 115-120 us: locality is not what bounds the kernel.)
    auto cnt6 = [&](unsigned col, unsigned out6[6]) {  // small / general entry counts of A, B, C
      for (int m = 0; m < 3; ++m) {
        out6[2 * m] = col_host[m].sptr[col + 1] - col_host[m].sptr[col];
        out6[2 * m + 1] = col_host[m].gptr[col + 1] - col_host[m].gptr[col];
      }
    };
    auto by_len = [&](unsigned x, unsigned y) {
      if (col_count[x] != col_count[y]) return col_count[x] > col_count[y];
      unsigned a[6], b[6];
      cnt6(x, a);
      cnt6(y, b);
      for (int k = 0; k < 5; ++k)
        if (a[k] != b[k]) return a[k] > b[k];
      return false;
    };
    static const int order_mode = [] {
      const char* e = getenv("SPARTAN_POLYABC_ORDER");  // "natural": column order as it is; "window": sorted inside windows of 4096 neighbouring columns (A/B runs)
      return !e ? 0 : (e[0] == 'n' ? 1 : (e[0] == 'w' ? 2 : 0));
    }();
    if (order_mode == 0) std::stable_sort(order.begin(), order.end(), by_len);
    else if (order_mode == 2)
    {
      const char* w = getenv("SPARTAN_POLYABC_WINDOW");
      const size_t win = w && atol(w) > 0 ? (size_t)atol(w) : 4096;
      for (size_t lo = 0; lo < order.size(); lo += win) std::stable_sort(order.begin() + lo, order.begin() + std::min(order.size(), lo + win), by_len);
    }
    s->n_short = order.size();
    if ((rc = upload(&s->d_short_order, order))) {
      sp_shape_free(s);
      return rc;
    }
    // The column-major structure is STORED in the order it is walked - the short columns in `order`, then the long ones (round 6): lane i of the walk reads
    // entry i of the pointer arrays and the index lists of neighbouring lanes are neighbours in memory, instead of a dozen scattered 4-byte loads per
    // column through the permutation. (Config 4's synthetic instance - 3.8 M columns of 1-3 entries - walked in natural order took 1.02 ms of inner
    // sum-check against 1.45 sorted, and the SHA-256 instances the other way round, 120 against 76 us: with the storage permuted the sorted walk has both.)
    // SPARTAN_POLYABC_LAYOUT=natural keeps the columns where their numbers put them (A/B runs).
    static const bool permuted = [] {
      const char* e = getenv("SPARTAN_POLYABC_LAYOUT");
      return !(e && e[0] == 'n');
    }();
    s->col_permuted = permuted;
    std::vector<unsigned> seq;
    if (permuted) {
      seq = order;
      seq.insert(seq.end(), long_cols.begin(), long_cols.end());
    }
    for (int m = 0; m < 3; ++m) {
      if (!permuted) {
        if ((rc = upload_split(col_host[m], &s->col[m]))) {
          sp_shape_free(s);
          return rc;
        }
        continue;
      }
      const SplitHost& h = col_host[m];
      SplitHost q;
      q.sptr.assign(seq.size() + 1, 0);
      q.gptr.assign(seq.size() + 1, 0);
      q.sidx.reserve(h.sidx.size());
      q.scode.reserve(h.scode.size());
      q.gidx.reserve(h.gidx.size());
      q.gval.reserve(h.gval.size());
      for (size_t pos = 0; pos < seq.size(); ++pos) {
        const unsigned col = seq[pos];
        q.sidx.insert(q.sidx.end(), h.sidx.begin() + h.sptr[col], h.sidx.begin() + h.sptr[col + 1]);
        q.scode.insert(q.scode.end(), h.scode.begin() + h.sptr[col], h.scode.begin() + h.sptr[col + 1]);
        q.gidx.insert(q.gidx.end(), h.gidx.begin() + h.gptr[col], h.gidx.begin() + h.gptr[col + 1]);
        q.gval.insert(q.gval.end(), h.gval.begin() + h.gptr[col], h.gval.begin() + h.gptr[col + 1]);
        q.sptr[pos + 1] = (unsigned)q.sidx.size();
        q.gptr[pos + 1] = (unsigned)q.gidx.size();
      }
      if ((rc = upload_split(q, &s->col[m]))) {
        sp_shape_free(s);
        return rc;
      }
    }
  }
  SP_HIP(hipDeviceSynchronize());
  *out = s;
  return SP_OK;
}
void sp_shape_free(sp_shape* s) {
  if (!s) return;
  for (int m = 0; m < 3; ++m) {
    s->row[m].release();
    s->filtered[m].release();
    s->col[m].release();
  }
  hipFree(s->d_long_cols);
  hipFree(s->d_short_order);
  delete s;
}

int sp_shape_info(const sp_shape* s, uint64_t out[8]) {
  if (!s || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_shape_info: null argument");
  for (int m = 0; m < 3; ++m) {
    out[m] = s->nnz[m];
    out[3 + m] = s->nnz_filtered[m];
  }
  out[6] = s->n_long_cols;
  out[7] = s->n_short;
  return SP_OK;
}

static int spmv3(sp_ctx* c, const sp_shape* s, const SplitOnDevice* mats, const uint64_t* nnz, const sp_table* z, const sp_table* const* base,
                 sp_table** outs, const char* what) {
  const size_t nrows = s->dims.num_cons;
  if (z->len != s->num_cols) return fail(SP_ERR_INVALID_WITNESS_LENGTH, "multiply_vec: z has the wrong length");
  spk::Spmv3Args a;
  for (int m = 0; m < 3; ++m) {
    if (outs[m]->cap < nrows) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec: output table too short");
    if (base && base[m]->cap < nrows) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_incremental: cached table too short");
    a.m[m] = mats[m].view();
    a.base[m] = base ? base[m]->d : nullptr;
    a.out[m] = outs[m]->d;
    outs[m]->len = nrows;
    outs[m]->lo_eff = outs[m]->hi_eff = (size_t)-1;
  }
  size_t blocks = (nrows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  // sum_nnz (4 + 32) [+32 per general coefficient, ignored] + 3*32*N outputs (+ cached reads when incremental)
  uint64_t bytes = 36ull * (nnz[0] + nnz[1] + nnz[2]) + 96ull * nrows * (base ? 2 : 1);
  const bool long_rows = mats[0].max_len > spk::SPMV_LONG_ROW || mats[1].max_len > spk::SPMV_LONG_ROW || mats[2].max_len > spk::SPMV_LONG_ROW;
  c->timed(what, bytes, [&] {
    if (long_rows) hipLaunchKernelGGL(spk::k_spmv3<true>, dim3((unsigned)blocks, 3), dim3(256), 0, c->stream, a, z->d, nrows);
    else hipLaunchKernelGGL(spk::k_spmv3<false>, dim3((unsigned)blocks, 3), dim3(256), 0, c->stream, a, z->d, nrows);
  });
  return SP_OK;
}

int sp_multiply_vec(sp_ctx* c, const sp_shape* s, const sp_table* z, sp_table* az, sp_table* bz, sp_table* cz) {
  sp_table* outs[3] = {az, bz, cz};
  return spmv3(c, s, s->row, s->nnz, z, nullptr, outs, "spmv");
}
int sp_multiply_vec_batched(sp_ctx* c, const sp_shape* s, const sp_table* const* zs, size_t count, sp_table* const* az, sp_table* const* bz, sp_table* const* cz) {
  if (count && (!zs || !az || !bz || !cz)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_batched: null argument");
  // one launch per vector: the matrices (a few MB of indices) stay in the L2s between launches, which is what the reference's single pass over the
  // matrix for all vectors buys on a CPU
  for (size_t k = 0; k < count; ++k) {
    sp_table* outs[3] = {az[k], bz[k], cz[k]};
    int rc = spmv3(c, s, s->row, s->nnz, zs[k], nullptr, outs, "spmv");
    if (rc) return rc;
  }
  return SP_OK;
}
int sp_multiply_vec_incremental(sp_ctx* c, const sp_shape* s, const sp_table* z, const sp_table* caz, const sp_table* cbz, const sp_table* ccz,
                                sp_table* az, sp_table* bz, sp_table* cz) {
  sp_table* outs[3] = {az, bz, cz};
  const sp_table* base[3] = {caz, cbz, ccz};
  return spmv3(c, s, s->filtered, s->nnz_filtered, z, base, outs, "spmv_incremental");
}

int sp_multiply_vec_incremental_round0(sp_ctx* c, const sp_shape* s, const sp_table* z, const sp_table* caz, const sp_table* cbz, const sp_table* ccz, sp_table* az,
                                       sp_table* bz, sp_table* cz, sp_table* p0, sp_table* p1) {
  const size_t nrows = s->dims.num_cons, half = nrows / 2;
  if (nrows < 2) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_incremental_round0: needs at least two rows");
  if (p0->cap < half || p1->cap < half) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_incremental_round0: product tables too short");
  int rc = sp_multiply_vec_incremental(c, s, z, caz, cbz, ccz, az, bz, cz);
  if (rc) return rc;
  p0->len = p1->len = half;
  p0->lo_eff = p0->hi_eff = p1->lo_eff = p1->hi_eff = (size_t)-1;
  size_t blocks = (half + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  c->timed("round0_products", 224ull * half, [&] { hipLaunchKernelGGL(spk::k_round0_products, dim3((unsigned)blocks), dim3(256), 0, c->stream, az->d, bz->d, cz->d, half, p0->d, p1->d); });
  return SP_OK;
}

int sp_poly_abc(sp_ctx* c, const sp_shape* s, const sp_table* rx, const uint64_t r_[4], size_t out_len, sp_table* out) {
  if (rx->len != s->dims.num_cons) return fail(SP_ERR_INVALID_INPUT_LENGTH, "poly_ABC: rx must have num_cons elements");
  if (out_len < s->num_cols || out->cap < out_len) return fail(SP_ERR_INVALID_INPUT_LENGTH, "poly_ABC: output too short");
  spk::PolyAbcArgs a;
  for (int m = 0; m < 3; ++m) a.m[m] = s->col[m].view();
  memcpy(&a.r, r_, 32);
  a.r2 = fe_mul<S>(a.r, a.r);
  // the long columns' partial triples and arrival counters belong to the CONTEXT (a shape is shared by every context that proves with its key): the
  // counters are zeroed when the buffer is (re)allocated and each column's last block leaves its counter at zero again
  const size_t ticket_bytes = (s->n_long_cols + 1) * sizeof(unsigned);
  const bool fresh = c->ws_bytes[sp_ctx::WS_POLYABC_TICKETS] < ticket_bytes;
  fe_t* partials = (fe_t*)c->workspace(sp_ctx::WS_POLYABC_PARTIALS, (s->n_long_cols + 1) * spk::LONG_NB_MAX * 3 * sizeof(fe_t));
  unsigned* tickets = (unsigned*)c->workspace(sp_ctx::WS_POLYABC_TICKETS, ticket_bytes);
  if (!partials || !tickets) return SP_ERR_NO_DEVICE;
  if (fresh) SP_HIP(hipMemsetAsync(tickets, 0, c->ws_bytes[sp_ctx::WS_POLYABC_TICKETS], c->stream));
  size_t blocks = (s->n_short + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  const uint64_t bytes = 36ull * (s->nnz[0] + s->nnz[1] + s->nnz[2]) + 32ull * out_len;
  const size_t zero_n = out_len - s->num_cols;
  size_t zblocks = (2 * zero_n + 256 * 16 - 1) / (256 * 16);  // 16 stores of 16 bytes per thread
  if (zblocks > 2048) zblocks = 2048;
  c->timed("poly_abc", bytes, [&] {
    hipLaunchKernelGGL(spk::k_polyabc_short_and_long, dim3((unsigned)(blocks + spk::LONG_NB_MAX * s->n_long_cols + zblocks)), dim3(256), 0, c->stream, a, rx->d, s->d_short_order,
                       s->n_short, out->d, s->d_long_cols, (unsigned)s->n_long_cols, partials, tickets, (unsigned)blocks, (size_t)s->num_cols, zero_n, s->col_permuted ? 1 : 0);
  });
  return SP_OK;
}

}  // extern "C"
