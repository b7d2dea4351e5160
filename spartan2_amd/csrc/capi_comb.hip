// libspartan_hip.so — fixed-base comb tables of the commitment key: PCS::commit of many non-small rows over one key (hyrax_pc.rs:230-300 with the
// per-base tables of :81-96 / msm.rs:653-773 extended from <= 64 bases to the whole 2048-base key, which 288 GB of HBM makes affordable).
#include <cstdlib>
#include <cstring>

#include "group_common.hpp"
#include "kernels_comb.hpp"

using sp::fail;

#ifndef COMB_MINW_DEFAULT
#define COMB_MINW_DEFAULT 3
#endif
namespace sp {

// ---- fixed-base comb path for many rows over the key (kernels_comb.hpp) ---------------------------------------------------------------------------
// SPARTAN_COMB_BITS = signed window width C (8, 10, 12, 13 or 14; 0 disables the path): the table takes ceil(257 / C) * num_cols * 2^(C-1) * 64 bytes
// (2048 bases: C = 12: 5.9 GB, C = 13 (default): 10.7 GB, C = 14: 20 GB; measured sweep in profiles/r02_comb_bits_sweep.txt). A commit must have 256 digit-path rows before the table is built (comb_min_rows).
static int comb_bits() {
  static const int v = [] {
    const char* e = getenv("SPARTAN_COMB_BITS");
    int b = e ? atoi(e) : 13;
    if (b != 0 && b != 8 && b != 10 && b != 12 && b != 13 && b != 14) b = 13;
    return b;
  }();
  return v;
}
size_t comb_min_rows() { return 256; }
static int comb_build(sp_ctx* c, const sp_ck* ck, aff_t** out_tab, int* out_c, int* out_windows) {  // 0 = built, 1 = not available, < 0 = error
  const int C = comb_bits(), windows = (257 + C - 1) / C;
  const unsigned E = 1u << (C - 1);
  const size_t ncols = ck->num_cols, per_window = ncols * E, total = per_window * windows;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total * sizeof(aff_t) + ((size_t)4 << 30)) return 1;
  aff_t* tab = nullptr;
  if (hipMalloc((void**)&tab, total * sizeof(aff_t)) != hipSuccess) return 1;
  // Jacobian staging for a group of windows (<= 8 GiB), then Montgomery's trick per 8 points
  size_t gw = ((size_t)8 << 30) / (per_window * sizeof(jac_t));
  if (gw < 1) gw = 1;
  if (gw > (size_t)windows) gw = windows;
  while (gw > 1 && free_b < total * sizeof(aff_t) + gw * per_window * sizeof(jac_t) + ((size_t)2 << 30)) --gw;
  jac_t* stage = nullptr;
  if (hipMalloc((void**)&stage, gw * per_window * sizeof(jac_t)) != hipSuccess) {
    hipFree(tab);
    return 1;
  }
  const unsigned segs = E >= 64 ? 8 : 1;
  for (int w0 = 0; w0 < windows; w0 += (int)gw) {
    const int nw = windows - w0 < (int)gw ? windows - w0 : (int)gw;
    const size_t lanes = (size_t)nw * ncols * segs, pts = (size_t)nw * per_window;
    hipLaunchKernelGGL(spk::k_comb_multiples, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, c->stream, ck->d_bases, (unsigned)ncols, C, w0, nw, E, segs, stage);
    hipLaunchKernelGGL(spk::k_comb_normalize, dim3((unsigned)((pts / 8 + 255) / 256 + 1)), dim3(256), 0, c->stream, stage, pts, tab + (size_t)w0 * per_window);
  }
  hipError_t e = sp::stream_sync(c->stream);
  hipFree(stage);
  if (e != hipSuccess) {
    hipFree(tab);
    return fail(SP_ERR_NO_DEVICE, std::string("comb table build: ") + hipGetErrorString(e));
  }
  *out_tab = tab;
  *out_c = C;
  *out_windows = windows;
  return SP_OK;
}
int comb_ensure(sp_ctx* c, const sp_ck* ck) {
  for (;;) {  // claim the build, find it done, or wait for the builder outside the lock (group_common.hpp: no mutex is held across a stream wait)
    {
      std::lock_guard<std::mutex> lk(ck->lazy_mu);
      if (ck->d_comb) return SP_OK;
      if (ck->comb_failed || comb_bits() == 0) return 1;  // not available: the caller takes the bucket path
      if (!ck->comb_building) {
        ck->comb_building = true;
        break;
      }
    }
    sp::relax();
  }
  aff_t* tab = nullptr;
  int C = 0, windows = 0;
  const int rc = comb_build(c, ck, &tab, &C, &windows);
  std::lock_guard<std::mutex> lk(ck->lazy_mu);
  ck->comb_building = false;
  if (rc) {
    ck->comb_failed = true;
    return rc;
  }
  ck->comb_c = C;
  ck->comb_windows = windows;
  ck->d_comb = tab;
  return SP_OK;
}
// waves per SIMD k_comb_rows is compiled for: 2 = 256 registers a lane, no spills; 3 = 168 registers, 67 spilled (tools/spill_report.py). SPARTAN_COMB_MINW picks
// (A/B: profiles/r06_spills.md)
static int comb_minw() {
  static const int v = [] {
    const char* e = getenv("SPARTAN_COMB_MINW");
    return e && e[0] == '3' ? 3 : (e && e[0] == '2' ? 2 : COMB_MINW_DEFAULT);
  }();
  return v;
}
// rows `sel` of canon (values < 2^nbits) against the comb table -> out[sel[i]]
int comb_rows(sp_ctx* c, const sp_ck* ck, const fe_t* canon, size_t cols, size_t n, const std::vector<unsigned>& sel, int nbits, std::vector<jac_t>& out) {
  const int C = ck->comb_c;
  int windows = (nbits + 1 + C - 1) / C;
  if (windows > ck->comb_windows) windows = ck->comb_windows;
  DevBuf dsel, drows;
  int rc;
  if ((rc = dsel.alloc(sel.size() * 4)) || (rc = drows.alloc(sel.size() * sizeof(jac_t)))) return rc;
  SP_HIP(hipMemcpyAsync(dsel.p, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, c->stream));
  const dim3 grid((unsigned)sel.size()), block(256);
  // 96 B per (scalar, base) pair (+ one 64-byte table entry per window actually gathered)
  c->timed("msm_rows_comb", 96ull * sel.size() * cols, [&] {
#define COMB_ROWS(CC, MW) \
  hipLaunchKernelGGL((spk::k_comb_rows<CC, MW>), grid, block, 0, c->stream, canon, dsel.as<unsigned>(), cols, n, ck->d_comb, windows, drows.as<jac_t>())
    if (comb_minw() == 2) {
      switch (C) {
        case 8: COMB_ROWS(8, 2); break;
        case 10: COMB_ROWS(10, 2); break;
        case 12: COMB_ROWS(12, 2); break;
        case 13: COMB_ROWS(13, 2); break;
        default: COMB_ROWS(14, 2); break;
      }
    } else {
      switch (C) {
        case 8: COMB_ROWS(8, 3); break;
        case 10: COMB_ROWS(10, 3); break;
        case 12: COMB_ROWS(12, 3); break;
        case 13: COMB_ROWS(13, 3); break;
        default: COMB_ROWS(14, 3); break;
      }
    }
#undef COMB_ROWS
  });
  std::vector<jac_t> res(sel.size());
  SP_HIP(hipMemcpyAsync(res.data(), drows.p, sel.size() * sizeof(jac_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  for (size_t i = 0; i < sel.size(); ++i) out[sel[i]] = res[i];
  return SP_OK;
}


}  // namespace sp
