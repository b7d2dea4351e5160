// libspartan_hip.so — group / MSM / Hyrax entry points of include/spartan_hip.h.
// Device: digit sort, bucket accumulation, per-window weighted reduction, binary row sums, fixed-base lookups,
// row-matrix product. Host (inside the library): window Horner, adding the blind term, batch normalisation.
#include <cstring>
#include <vector>

#include "core.hpp"
#include "kernels_msm.cuh"

using sp::fail;
typedef FqP S;

struct sp_ck {
  sp_ctx* ctx = nullptr;
  size_t num_cols = 0;
  aff_t* d_bases = nullptr;
  aff_t h;
  aff_t* d_htable = nullptr;  // 32 * 255 affine multiples of h
  aff_t* d_cktables = nullptr;  // num_cols <= 64: one 32*255 table per base (hyrax_pc.rs:81-96 ck_tables)
};

namespace {

struct DevBuf {  // RAII device allocation
  void* p = nullptr;
  ~DevBuf() {
    if (p) hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(SP_ERR_NO_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    return SP_OK;
  }
  template <class T>
  T* as() {
    return (T*)p;
  }
};

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

// Pippenger on the device for n canonical scalars already in HBM; returns the Jacobian sum on the host.
int msm_device(sp_ctx* c, const fe_t* d_canon, const aff_t* d_bases, size_t n, int windows, jac_t* result) {
  *result = jac_identity();
  if (n == 0) return SP_OK;
  if (n >= (1u << 31)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "msm: n too large");
  unsigned* order = (unsigned*)c->workspace(sp_ctx::WS_MSM_ORDER, (size_t)windows * n * 4);
  unsigned* start = (unsigned*)c->workspace(sp_ctx::WS_MSM_START, (size_t)windows * (spk::MSM_BUCKETS + 1) * 4);
  jac_t* buckets = (jac_t*)c->workspace(sp_ctx::WS_MSM_BUCKETS, (size_t)windows * spk::MSM_BUCKETS * sizeof(jac_t));
  jac_t* wsum = (jac_t*)c->workspace(sp_ctx::WS_MSM_WSUM, (size_t)windows * sizeof(jac_t));
  if (!order || !start || !buckets || !wsum) return SP_ERR_NO_DEVICE;
  c->timed("msm_sort", 32ull * n, [&] {
    hipLaunchKernelGGL(spk::k_msm_sort, dim3(windows), dim3(256), 0, c->stream, d_canon, (unsigned)n, order, start);
  });
  unsigned lanes = (unsigned)windows * spk::MSM_BUCKETS * spk::MSM_LANES_PER_BUCKET;
  c->timed("msm_bucket_sum", 96ull * n, [&] {
    hipLaunchKernelGGL(spk::k_msm_bucket_sum, dim3((lanes + 255) / 256), dim3(256), 0, c->stream, d_bases, (unsigned)n, order, start, windows, buckets);
  });
  c->timed("msm_window_reduce", 0, [&] {
    hipLaunchKernelGGL(spk::k_msm_window_reduce, dim3(windows), dim3(spk::MSM_BUCKETS), 0, c->stream, buckets, wsum);
  });
  std::vector<jac_t> w(windows);
  SP_HIP(hipMemcpyAsync(w.data(), wsum, windows * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(hipStreamSynchronize(c->stream));
  // Horner over windows, high to low (msm.rs:150-175): acc = 2^8 acc + W_w
  jac_t acc = jac_identity();
  for (int i = windows - 1; i >= 0; --i) {
    for (int k = 0; k < spk::MSM_C; ++k) acc = jac_dbl(acc);
    acc = jac_add(acc, w[i]);
  }
  *result = acc;
  return SP_OK;
}

int upload_canonical(sp_ctx* c, const uint64_t* scalars, size_t n, fe_t** canon_out) {
  fe_t* raw = (fe_t*)c->workspace(sp_ctx::WS_SCALARS_RAW, n * sizeof(fe_t));
  fe_t* canon = (fe_t*)c->workspace(sp_ctx::WS_SCALARS_CANON, n * sizeof(fe_t));
  if (!raw || !canon) return SP_ERR_NO_DEVICE;
  if (n) {
    SP_HIP(hipMemcpyAsync(raw, scalars, n * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(spk::k_to_canonical, dim3((unsigned)blocks), dim3(256), 0, c->stream, raw, n, canon);
  }
  SP_HIP(hipStreamSynchronize(c->stream));  // `scalars` is a borrowed host buffer
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
  if ((rc = msm_device(c, canon, dbases, n, spk::MSM_MAX_WINDOWS, &r))) return rc;
  store_aff(out_aff, jac_to_affine(r));
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
  if ((rc = msm_device(c, canon.as<fe_t>(), dbases.as<aff_t>(), n, 9, &r))) return rc;
  store_aff(out_aff, jac_to_affine(r));
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
  DevBuf tj;
  int rc;
  if ((rc = tj.alloc(ntab * per * sizeof(jac_t)))) return rc;
  aff_t* tables;
  SP_HIP(hipMalloc((void**)&tables, ntab * per * sizeof(aff_t)));
  for (size_t t = 0; t < ntab; ++t) {
    aff_t base = (t + 1 == ntab) ? k->h : load_aff(ck_aff + 8 * t);
    hipLaunchKernelGGL(spk::k_fixed_base_table, dim3(1), dim3(64), 0, c->stream, base, tj.as<jac_t>() + t * per);
  }
  hipLaunchKernelGGL(spk::k_jac_to_affine, dim3((unsigned)((ntab * per + 255) / 256)), dim3(256), 0, c->stream, tj.as<jac_t>(), ntab * per, tables);
  SP_HIP(hipStreamSynchronize(c->stream));
  if (ntab > 1) {
    k->d_cktables = tables;
    k->d_htable = tables + (ntab - 1) * per;
  } else {
    k->d_htable = tables;
  }
  *out = k;
  return SP_OK;
}
void sp_ck_free(sp_ck* k) {
  if (!k) return;
  if (k->d_bases) hipFree(k->d_bases);
  if (k->d_cktables) hipFree(k->d_cktables);
  else if (k->d_htable) hipFree(k->d_htable);
  delete k;
}

// table[i % ntables] * scalars[i] on the device (Jacobian results in host memory); ntables == 1 for h
static int fixed_base_rows(sp_ctx* c, const aff_t* d_tables, size_t ntables, const uint64_t* scalars, size_t n, std::vector<jac_t>& out) {
  out.assign(n, jac_identity());
  if (n == 0) return SP_OK;
  fe_t* ds = (fe_t*)c->workspace(sp_ctx::WS_FB_SCALARS, n * sizeof(fe_t));
  jac_t* dout = (jac_t*)c->workspace(sp_ctx::WS_FB_OUT, n * sizeof(jac_t));
  if (!ds || !dout) return SP_ERR_NO_DEVICE;
  SP_HIP(hipMemcpyAsync(ds, scalars, n * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  size_t threads = n * 32;
  c->timed("fixed_base", 32ull * n, [&] {
    hipLaunchKernelGGL(spk::k_fixed_base_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, ds, n, d_tables, ntables, dout);
  });
  SP_HIP(hipMemcpyAsync(out.data(), dout, n * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(hipStreamSynchronize(c->stream));
  return SP_OK;
}

int sp_fixed_base_mul_h(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, uint64_t* out_aff) {
  std::vector<jac_t> pts;
  int rc = fixed_base_rows(c, ck->d_htable, 1, scalars, n, pts);
  if (rc) return rc;
  std::vector<aff_t> a(n);
  normalize_batch(pts, a.data());
  if (n) memcpy(out_aff, a.data(), n * sizeof(aff_t));
  return SP_OK;
}

int sp_hyrax_commit(sp_ctx* c, const sp_ck* ck, const sp_table* v, size_t off, size_t n, const uint64_t* blinds, int /*is_small: auto-detected*/,
                    uint64_t* out_rows_aff) {
  if (off + n > v->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "hyrax commit: range exceeds the table");
  const size_t cols = ck->num_cols, rows = (n + cols - 1) / cols;
  if (rows == 0) return SP_OK;
  int rc;
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
  SP_HIP(hipStreamSynchronize(c->stream));
  for (size_t r = 0; r < rows; ++r) {
    if (hflags[r] > 1u) {  // msm_10 / msm_small_rest / full msm rows (msm.rs:367-409, :187-222): digit path
      size_t lo = r * cols, len = (lo + cols <= n) ? cols : n - lo;
      if ((rc = msm_device(c, canon + lo, ck->d_bases, len, (hflags[r] & 4u) ? spk::MSM_MAX_WINDOWS : 9, &msm_rows[r]))) return rc;
    }
  }
  std::vector<jac_t> hb;
  if ((rc = fixed_base_rows(c, ck->d_htable, 1, blinds, rows, hb))) return rc;
  for (size_t r = 0; r < rows; ++r) msm_rows[r] = jac_add(msm_rows[r], hb[r]);
  std::vector<aff_t> a(rows);
  normalize_batch(msm_rows, a.data());
  memcpy(out_rows_aff, a.data(), rows * sizeof(aff_t));
  return SP_OK;
}

int sp_rowmat_vec(sp_ctx* c, const sp_table* poly, size_t rows, size_t cols, const uint64_t* L, uint64_t* out) {
  if (rows * cols > poly->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "bind_with_delayed: poly shorter than rows*cols");
  if (rows == 0 || cols == 0) return SP_OK;
  size_t splits = rows < 64 ? rows : 64;
  fe_t* dL = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_L, rows * sizeof(fe_t));
  fe_t* part = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_PART, splits * cols * sizeof(fe_t));
  fe_t* dout = (fe_t*)c->workspace(sp_ctx::WS_ROWMAT_OUT, cols * sizeof(fe_t));
  if (!dL || !part || !dout) return SP_ERR_NO_DEVICE;
  SP_HIP(hipMemcpyAsync(dL, L, rows * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  c->timed("rowmat_vec", 32ull * (rows * cols + rows + cols), [&] {
    hipLaunchKernelGGL(spk::k_rowmat_vec, dim3((unsigned)((cols + 63) / 64), (unsigned)splits), dim3(256), 0, c->stream, poly->d, rows, cols,
                       dL, part);
    hipLaunchKernelGGL(spk::k_sum_columns, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, c->stream, part, splits, cols, dout);
  });
  SP_HIP(hipMemcpyAsync(out, dout, cols * sizeof(fe_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(hipStreamSynchronize(c->stream));
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
    std::vector<jac_t> hb;
    if ((rc = fixed_base_rows(c, ck->d_htable, 1, blind, 1, hb))) return rc;
    r = jac_add(r, hb[0]);
  }
  store_aff(out_aff, jac_to_affine(r));
  return SP_OK;
}

int sp_hyrax_commit_small(sp_ctx* c, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t blind[4], uint64_t out_aff[8]) {
  if (!ck->d_cktables || n > ck->num_cols) return fail(SP_ERR_INVALID_INPUT_LENGTH, "commit_small: key wider than 64 or too many scalars");
  // one launch: n lookups in the per-base tables + 1 in the h table (tables are contiguous: bases then h)
  std::vector<uint64_t> sc((n + 1) * 4);
  memcpy(sc.data(), scalars, n * 32);
  memcpy(sc.data() + 4 * n, blind, 32);
  std::vector<jac_t> parts;
  // table index for element i is (ck->num_cols - n + i) ... simpler: run the bases and h separately when n < num_cols
  int rc;
  if (n == ck->num_cols) {
    if ((rc = fixed_base_rows(c, ck->d_cktables, ck->num_cols + 1, sc.data(), n + 1, parts))) return rc;
  } else {
    std::vector<jac_t> a, b;
    if ((rc = fixed_base_rows(c, ck->d_cktables, ck->num_cols + 1, sc.data(), n, a))) return rc;
    if ((rc = fixed_base_rows(c, ck->d_htable, 1, blind, 1, b))) return rc;
    parts = a;
    parts.push_back(b[0]);
  }
  jac_t acc = jac_identity();
  for (const jac_t& p : parts) acc = jac_add(acc, p);
  store_aff(out_aff, jac_to_affine(acc));
  return SP_OK;
}

}  // extern "C"
