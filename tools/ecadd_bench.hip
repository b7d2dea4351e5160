// Latency of a chain of dependent Jacobian additions in one wave: the ordinary 16-product formula vs the 4-lane cooperative form
// (curve.hpp jac_add_coop4). Also checks that both give the same point.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ispartan2_amd/csrc tools/ecadd_bench.hip -o tools/ecadd_bench
#include <cstdio>
#include <vector>

#include "curve.hpp"

// EXPERIMENT (not used by the library): measured 16.3 us (ordinary) vs 10.5 us (cooperative) per dependent addition on MI355X — the
// selects / modular adds around the five product levels cost as much as four more products, so the gain did not justify replicating points
// over quads in the MSM kernels.
// ---- lane-cooperative Jacobian addition ------------------------------------------------------------------------------------------------
// A lone lane issues one base-field product in ~1 us (the 64-bit multiply-adds are quarter rate), so a Jacobian addition — 16 dependent-ish
// products in one instruction stream — costs ~15 us whenever the machine is latency-bound (bucket trees, window reductions, fixed-base
// trees: few points, long chains). Here FOUR adjacent lanes that hold the SAME two points share one addition: the 16 products are scheduled
// in 5 levels of at most 4, each lane computes one product per level and the results are exchanged with shuffles, so the chain is 5 products
// deep instead of 16. All four lanes return the same point. Rare cases (identity operand, P = +-Q) take the ordinary formula.
__device__ __forceinline__ fe_t coop4_pick(const fe_t& a0, const fe_t& a1, const fe_t& a2, const fe_t& a3, int sub) {
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = sub == 0 ? a0.v[i] : (sub == 1 ? a1.v[i] : (sub == 2 ? a2.v[i] : a3.v[i]));
  return r;
}
template <int K>
__device__ __forceinline__ fe_t coop4_from(const fe_t& v) {  // value of `v` held by lane K of this quad (DPP quad_perm broadcast: VALU rate, no LDS crossbar)
  fe_t r;
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[i], ctrl, 0xf, 0xf, true);
  return r;
}
__device__ __forceinline__ jac_t jac_add_coop4(const jac_t& p, const jac_t& q) {
  if (jac_is_identity(p)) return q;
  if (jac_is_identity(q)) return p;
  const int sub = __lane_id() & 3;
  const fe_t zero = fe_zero();
  // level 1: z1z1 = Z1^2 | z2z2 = Z2^2 | zz = (Z1+Z2)^2 | a = Y1 Z2
  const fe_t zs = fe_add<B>(p.z, q.z);
  fe_t m = fe_mul<B>(coop4_pick(p.z, q.z, zs, p.y, sub), coop4_pick(p.z, q.z, zs, q.z, sub));
  const fe_t z1z1 = coop4_from<0>(m), z2z2 = coop4_from<1>(m), zz = coop4_from<2>(m), a = coop4_from<3>(m);
  // level 2: u1 = X1 z2z2 | u2 = X2 z1z1 | s1 = a z2z2 | b = Y2 Z1
  m = fe_mul<B>(coop4_pick(p.x, q.x, a, q.y, sub), coop4_pick(z2z2, z1z1, z2z2, p.z, sub));
  const fe_t u1 = coop4_from<0>(m), u2 = coop4_from<1>(m), s1 = coop4_from<2>(m), b = coop4_from<3>(m);
  const fe_t h = fe_sub<B>(u2, u1);
  // level 3: s2 = b z1z1 | i = (2h)^2 | z3 = (zz - z1z1 - z2z2) h | -
  const fe_t h2 = fe_dbl<B>(h), zc = fe_sub<B>(fe_sub<B>(zz, z1z1), z2z2);
  m = fe_mul<B>(coop4_pick(b, h2, zc, zero, sub), coop4_pick(z1z1, h2, h, zero, sub));
  const fe_t s2 = coop4_from<0>(m), i = coop4_from<1>(m), z3 = coop4_from<2>(m);
  const fe_t rr = fe_dbl<B>(fe_sub<B>(s2, s1));
  if (fe_is_zero(h)) {  // P = Q or P = -Q: uniform across the group (same inputs)
    if (fe_is_zero(rr)) return jac_dbl(p);
    return jac_identity();
  }
  // level 4: j = h i | v = u1 i | rr^2 | -
  m = fe_mul<B>(coop4_pick(h, u1, rr, zero, sub), coop4_pick(i, i, rr, zero, sub));
  const fe_t j = coop4_from<0>(m), v = coop4_from<1>(m), rr2 = coop4_from<2>(m);
  jac_t r;
  r.x = fe_sub<B>(fe_sub<B>(rr2, j), fe_dbl<B>(v));
  // level 5: rr (v - x3) | s1 j
  m = fe_mul<B>(coop4_pick(rr, s1, zero, zero, sub), coop4_pick(fe_sub<B>(v, r.x), j, zero, zero, sub));
  const fe_t m1 = coop4_from<0>(m), m2 = coop4_from<1>(m);
  r.y = fe_sub<B>(m1, fe_dbl<B>(m2));
  r.z = z3;
  return r;
}


// per-lane data (so nothing is scalarised): lane group g = lane / grp walks the point list from offset g
__global__ void k_chain(const jac_t* pts, int n, int coop, jac_t* out) {
  const int g = threadIdx.x >> 2;  // the four lanes of a quad share their points in both variants (same work, comparable result)
  jac_t acc = pts[g % n];
  for (int i = 1; i < n; ++i) {
    const jac_t q = pts[(g + i) % n];
    acc = coop ? jac_add_coop4(acc, q) : jac_add(acc, q);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ void k_mults(jac_t g, int n, jac_t* out) {  // pts[i] = (i+1) * 3 * G, distinct points
  jac_t acc = jac_dbl(g);
  acc = jac_add(acc, g);
  jac_t step = acc;
  for (int i = 0; i < n; ++i) {
    out[i] = acc;
    acc = jac_add(jac_dbl(acc), step);
  }
}
int main() {
  const int n = 64;
  jac_t *pts, *out;
  hipMalloc(&pts, n * sizeof(jac_t));
  hipMalloc(&out, 4 * sizeof(jac_t));
  // G = (3, 0x5a6dd32df58708e64e97345cbe66600decd9d538a351bb3c30b4954925b1f02d)
  aff_t g;
  g.x = fe_from_u64<B>(3);
  {
    fe_t c;
    const unsigned w[8] = {0x25b1f02du, 0x30b49549u, 0xa351bb3cu, 0xecd9d538u, 0xbe66600du, 0x4e97345cu, 0xf58708e6u, 0x5a6dd32du};
    for (int i = 0; i < 8; ++i) c.v[i] = w[i];
    g.y = fe_from_canonical<B>(c);
  }
  if (!aff_on_curve(g)) printf("generator not on curve!\n");
  hipLaunchKernelGGL(k_mults, dim3(1), dim3(1), 0, 0, jac_from_affine(g), n, pts);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  jac_t res[2];
  for (int coop = 0; coop < 2; ++coop) {
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, pts, n, coop, out + coop);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, pts, n, coop, out + coop);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%s: %d dependent additions in %.1f us -> %.2f us per addition\n", coop ? "coop4  " : "ordinary", n - 1, ms * 1e3, ms * 1e3 / (n - 1));
    hipMemcpy(&res[coop], out + coop, sizeof(jac_t), hipMemcpyDeviceToHost);
  }
  aff_t x = jac_to_affine(res[0]), y = jac_to_affine(res[1]);
  printf("same point: %s\n", (fe_eq(x.x, y.x) && fe_eq(x.y, y.y)) ? "yes" : "NO");
  return 0;
}
