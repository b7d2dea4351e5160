// libspartan_hip.so - launchers of the throughput-shaped group kernels (kernels_bulk.hpp): this translation unit is compiled with the default scheduler,
// capi_group.hip (which calls these through group_common.hpp) with max-ilp.
#include <hip/hip_runtime.h>

#include "group_common.hpp"
#include "kernels_bulk.hpp"

namespace sp {
void launch_fixed_base_table(hipStream_t st, const aff_t& base, jac_t* table_jac) { hipLaunchKernelGGL(spk::k_fixed_base_table, dim3(1), dim3(64), 0, st, base, table_jac); }
void launch_fixed_base_tables(hipStream_t st, const aff_t* d_bases, size_t n, jac_t* table_jac) {
  hipLaunchKernelGGL(spk::k_fixed_base_tables, dim3((unsigned)n), dim3(64), 0, st, d_bases, n, table_jac);
}
void launch_fixed_base_tables16(hipStream_t st, const aff_t* d_bases, size_t n, jac_t* table_jac) {
  hipLaunchKernelGGL(spk::k_fixed_base_tables16, dim3((unsigned)(n * 16)), dim3(256), 0, st, d_bases, n, table_jac);
}
// Curve::batch_normalize of n device points. With `pre` (n elements of scratch): Montgomery's trick, one inversion per 32 points; without: one per point
void launch_jac_to_affine(hipStream_t st, const jac_t* in, size_t n, aff_t* out, fe_t* pre) {
  if (!pre) {
    hipLaunchKernelGGL(spk::k_jac_to_affine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, out);
    return;
  }
  const unsigned K = n >= ((size_t)1 << 20) ? 32u : 16u;
  const size_t T = (n + K - 1) / K;
  hipLaunchKernelGGL(spk::k_jac_to_affine_batch, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, st, in, n, K, out, pre);
}
// scratch of one chunk of nb bases: ladder (Jacobian, affine, running products) + the fill's Jacobian entries and running products
size_t window_tables_scratch(size_t nb) {
  return 128 * nb * (sizeof(jac_t) + sizeof(aff_t) + sizeof(fe_t)) + (size_t)255 * 32 * nb * (sizeof(jac_t) + sizeof(fe_t)) + 256;
}
// one 32 x 255 table of affine multiples per point (FixedBaseMul::precompute, msm.rs:653-689, for every point): queued on `st`, nothing waited for.
// d_points: n affine points in device memory; scratch: window_tables_scratch(min(n, WT_CHUNK)) bytes that nothing else on another stream uses
void launch_window_tables(hipStream_t st, const aff_t* d_points, size_t n, char* scratch, aff_t* tables) {
  const size_t per = 32 * 255;
  for (size_t lo = 0; lo < n; lo += WT_CHUNK) {
    const size_t nb = n - lo < WT_CHUNK ? n - lo : WT_CHUNK;
    jac_t* lad_j = (jac_t*)scratch;
    aff_t* lad_a = (aff_t*)(lad_j + 128 * nb);
    fe_t* lad_p = (fe_t*)(lad_a + 128 * nb);
    jac_t* J = (jac_t*)(lad_p + 128 * nb);
    fe_t* pre = (fe_t*)(J + (size_t)255 * 32 * nb);
    hipLaunchKernelGGL(spk::k_fb_ladder, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, d_points + lo, nb, lad_j);
    const unsigned K = 16;
    const size_t T = (128 * nb + K - 1) / K;
    hipLaunchKernelGGL(spk::k_jac_to_affine_batch, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, st, lad_j, 128 * nb, K, lad_a, lad_p);
    hipLaunchKernelGGL(spk::k_fb_fill, dim3((unsigned)((8 * 32 * nb + 63) / 64)), dim3(64), 0, st, lad_a, nb, J, pre, tables + lo * per);
  }
}
// bind_with_delayed (hyrax_pc.rs:38-54) on `st`: the one-launch streaming kernel for tall matrices, else the two-stage form
void launch_rowmat_vec(hipStream_t st, const fe_t* poly, size_t rows, size_t cols, const fe_t* dL, fe_t* part, size_t splits, fe_t* dout) {
  if (rows >= 128 && cols % spk::RMV_COLS == 0) {
    const size_t l_bytes = rows <= (size_t)spk::RMV_L_MAX ? rows * sizeof(fe_t) : 0;
    hipLaunchKernelGGL(spk::k_rowmat_vec_tall, dim3((unsigned)(cols / spk::RMV_COLS)), dim3(spk::RMV_THREADS), l_bytes, st, poly, rows, cols, dL, dout);
    return;
  }
  hipLaunchKernelGGL(spk::k_rowmat_vec, dim3((unsigned)((cols + 63) / 64), (unsigned)splits), dim3(256), 0, st, poly, rows, cols, dL, part);
  hipLaunchKernelGGL(spk::k_sum_columns, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, part, splits, cols, dout);
}
}  // namespace sp
