// Shared by the group-side translation units of libspartan_hip.so (capi_group.hip, capi_comb.hip).
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "core.hpp"
#include "curve.hpp"

struct sp_ck {
  sp_ctx* ctx = nullptr;
  size_t num_cols = 0;
  aff_t* d_bases = nullptr;
  aff_t h;
  aff_t* d_htable = nullptr;  // 32 * 255 affine multiples of h
  aff_t* d_cktables = nullptr;  // num_cols <= 64: one 32*255 table per base (hyrax_pc.rs:81-96 ck_tables)
  aff_t* d_tables16 = nullptr;  // the same table set with 16-bit windows (16 x 65535 entries per base) for the latency paths of <= 128 scalars
  std::vector<aff_t> h_tables;  // host copy of all tables (bases..., h): single multiplications are latency-bound -> host
  size_t n_tables = 0;
  const aff_t* host_table(size_t t) const { return h_tables.data() + t * 32 * 255; }
  const aff_t* host_htable() const { return host_table(n_tables - 1); }
  // keys of <= 2 tables (the width-1 key of eval_W / beta, h of the wide key): a host copy of the 16-bit-window tables too (16 x 65535 entries a table,
  // 64 MiB each) - a single multiplication on the host is then 16 mixed additions instead of 32 (capi_group.hip ck_mul_host)
  // ONE copy per distinct base set and process (capi_group.hip host_tables16_of: eight contexts on one key share it), allocated without value-initialising;
  // SPARTAN_HOST_T16=0 keeps none (the single multiplications then walk the 8-bit tables: 32 mixed additions instead of 16)
  std::shared_ptr<aff_t[]> h_tables16;
  size_t h16_bases = 0;  // leading bases that have a host copy (h's table follows them): all of a key of <= 2 tables, the first 16 of a narrow key
  const aff_t* host_table16(size_t t) const {
    if (!h_tables16) return nullptr;
    if (t + 1 == n_tables) return h_tables16.get() + h16_bases * ((size_t)16 * 65535);
    return t < h16_bases ? h_tables16.get() + t * ((size_t)16 * 65535) : nullptr;
  }
  // fixed-base comb table of the whole key (kernels_msm.hpp k_comb_*), built on first use by a commitment of many non-small rows
  mutable aff_t* d_comb = nullptr;
  mutable int comb_c = 0, comb_windows = 0;
  mutable bool comb_failed = false, comb_building = false;
  // 8-bit window tables of every base and of h (capi_group.hip ck_key_tables), built on first use by sp_hyrax_prove
  mutable aff_t* d_keytables = nullptr;
  mutable bool keytables_failed = false, keytables_building = false;
  // The two lazily built table sets above are written through a `const sp_ck*` that several contexts / helper threads may share. lazy_mu guards the
  // pointers and flags ONLY: the builder claims the build under the lock (`*_building`), builds with the lock released, and publishes under the lock
  // again; everybody else waits through sp::relax() without holding it. A build synchronises with its stream, and on a thread that carries several
  // proofs that wait is the wait hook (a switch to another proof, which may ask for the same key): no library mutex is ever held across it.
  mutable std::mutex lazy_mu;
};


struct DevBuf {  // RAII device allocation
  void* p = nullptr;
  ~DevBuf() {
    if (p) hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return sp::fail(SP_ERR_NO_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    return SP_OK;
  }
  template <class T>
  T* as() {
    return (T*)p;
  }
};


namespace sp {
// capi_bulk.hip (default scheduler): FixedBaseMul::precompute, Curve::batch_normalize and bind_with_delayed launchers
static const size_t WT_CHUNK = 1024;  // bases per pass of launch_window_tables
void launch_fixed_base_table(hipStream_t st, const aff_t& base, jac_t* table_jac);
void launch_fixed_base_tables(hipStream_t st, const aff_t* d_bases, size_t n, jac_t* table_jac);
void launch_fixed_base_tables16(hipStream_t st, const aff_t* d_bases, size_t n, jac_t* table_jac);
void launch_jac_to_affine(hipStream_t st, const jac_t* in, size_t n, aff_t* out, fe_t* pre = nullptr);
size_t window_tables_scratch(size_t nb);
void launch_window_tables(hipStream_t st, const aff_t* d_points, size_t n, char* scratch, aff_t* tables);
void launch_rowmat_vec(hipStream_t st, const fe_t* poly, size_t rows, size_t cols, const fe_t* dL, fe_t* part, size_t splits, fe_t* dout);
// capi_group.hip: one-launch table-walk MSM (k_multi_mul_coop) on lane 0 (main stream) or 1 (auxiliary stream), and the window tables of a key
int multi_mul_ensure(sp_ctx* c, int lane);
size_t multi_mul_wide_min();
int multi_mul_launch(sp_ctx* c, int lane, const aff_t* d_tables, const uint64_t* scalars, size_t n, unsigned* seq_out, const fe_t* d_scalars, const fe_t* last,
                     size_t raw_blocks, bool expand = false);
int multi_mul_collect(sp_ctx* c, int lane, unsigned seq, jac_t* out, bool yield);
int ck_key_tables(sp_ctx* c, const sp_ck* ck);  // 0 = ready, 1 = not available (take the bucket MSM), < 0 = error
// capi_pippenger.hip: the general (multi-block) Pippenger for caller-supplied bases; window = 0 -> pippenger_window(n)
int pippenger_window(size_t n);
int msm_pippenger(sp_ctx* c, const fe_t* d_canon, const aff_t* d_bases, size_t n, bool full_width, int window, jac_t* result);
static const size_t PIPPENGER_MIN = 4096;  // below: the one-block-per-window latency form (kernels_msm.hpp), sized for the 2048-wide Hyrax MSMs
// capi_comb.hip: the fixed-base comb table of a key (built on first use) and row commitments over it
size_t comb_min_rows();
int comb_ensure(sp_ctx* c, const sp_ck* ck);  // 0 = table ready, 1 = not available (take the bucket path), < 0 = error
int comb_rows(sp_ctx* c, const sp_ck* ck, const fe_t* canon, size_t cols, size_t n, const std::vector<unsigned>& sel, int nbits, std::vector<jac_t>& out);
}  // namespace sp
