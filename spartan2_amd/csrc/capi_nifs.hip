// C ABI of the NeutronNova NIFS data path (include/spartan_hip.h, "NeutronNova NIFS rounds"): device-resident instance layers, the per-round
// (e0, quad) sums with the fold of the previous round merged in, the O(1) `finish_round!` algebra on the host side of the library.
// Reference: src/neutronnova_zk.rs:511-1273. gfx950 only; no CPU fallback.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "core.hpp"
#include "kernels_nifs.hpp"

using sp::fail;
typedef FqP SF;

struct sp_nifs {
  sp_ctx* ctx = nullptr;
  size_t n_padded = 0, left = 0, right = 0, total = 0, ell_b = 0;
  unsigned left_log2 = 0;
  bool factored = false;
  // layer storage for A and B: buf[0] holds the n_padded instance layers and is only ever READ by the rounds (so the layers prep_prove cached,
  // cached_step_matvec src/neutronnova_zk.rs:1520-1590, serve every prove without a copy); the folds ping-pong between buf[1] (n_padded / 2
  // layers) and buf[2] (n_padded / 4)
  fe_t *A[3] = {nullptr, nullptr, nullptr}, *B[3] = {nullptr, nullptr, nullptr}, *C = nullptr;
  long long *A64 = nullptr, *B64 = nullptr, *C64 = nullptr;  // small-value mirrors (built by sp_nifs_begin(small_values = 1))
  unsigned char* d_flags = nullptr;
  unsigned* d_large = nullptr;
  unsigned nlarge = 0;
  fe_t *d_E = nullptr, *d_w = nullptr, *d_part = nullptr, *d_part2 = nullptr, *d_cvals = nullptr;
  // the pair weights of every round depend on the rhos alone: computed and uploaded once at begin (offsets w_off, counts w_pairs), a round only picks its
  // range (d_w_cur); a round whose pair count differs from what begin assumed uploads into the second half of d_w as before
  fe_t* d_w_cur = nullptr;
  std::vector<fe_t> h_w;
  std::vector<size_t> w_off, w_pairs;
  fe_t* h_pin = nullptr;  // pinned landing area of the rounds' sums (a copy into pageable memory goes through the runtime's staging buffer)
  size_t part_elems = 0;
  // round state
  int cur = 0;       // buffer holding the current layers
  size_t m = 0;      // layers in buf[cur]
  size_t rounds_done = 0;
  bool have_poly = false;
  bool small = false;
  std::vector<fe_t> rhos, r_bs, c_vals, prefix;  // c_vals: one entry per instance of the WHOLE batch (2^ell_b)
  fe_t T_cur, acc_eq, poly[4];
  // sharding: this object holds instances [first, first + n_padded) of a batch of 2^ell_b; ell_b = log2(n_padded), first = 0 when unsharded
  size_t first = 0;
  bool fold_pending = false;  // a challenge has been received whose fold has not been applied to the layers yet
  bool mirrors_ready = false; // sp_nifs_prepare_small has run on the current layers
};

namespace {

fe_t h_one() { return fe_one<SF>(); }
int next_buf(int cur) { return cur == 1 ? 2 : 1; }  // buf[0] (the instance layers) is never a destination
fe_t load_fe(const uint64_t* p) {
  fe_t r;
  memcpy(r.v, p, 32);
  return r;
}
void store_fe(uint64_t* p, const fe_t& a) { memcpy(p, a.v, 32); }

spk::NifsGeom geom(const sp_nifs* n) {
  spk::NifsGeom g;
  g.total = n->total;
  g.left_log2 = n->left_log2;
  g.e_left = n->d_E;
  g.f = n->d_E + n->left;
  return g;
}

// sum `rows` rows of `n` NACC-interleaved partials in d_part -> host (rows * NACC elements)
template <int NACC>
int sum_partials(sp_nifs* n, size_t rows, size_t cnt, fe_t* out_host) {
  sp_ctx* c = n->ctx;
  const fe_t* src = n->d_part;
  fe_t* dst = n->d_part2;
  while (true) {
    size_t chunk = 4096, slices = (cnt + chunk - 1) / chunk;
    hipLaunchKernelGGL(spk::k_nifs_sum<NACC>, dim3((unsigned)slices, (unsigned)rows), dim3(256), 0, c->stream, src, (unsigned long long)cnt,
                       (unsigned long long)chunk, dst);
    if (slices == 1) break;
    cnt = slices;
    const fe_t* t = src;
    src = dst;
    dst = const_cast<fe_t*>(t);
  }
  const size_t cnt_out = rows * NACC;
  if (cnt_out <= 64) {
    if (!n->h_pin) SP_HIP(hipHostMalloc((void**)&n->h_pin, 64 * sizeof(fe_t)));
    SP_HIP(hipMemcpyAsync(n->h_pin, dst, cnt_out * sizeof(fe_t), hipMemcpyDeviceToHost, c->stream));
    SP_HIP(sp::stream_sync_short(c->stream));
    memcpy(out_host, n->h_pin, cnt_out * sizeof(fe_t));
    return SP_OK;
  }
  SP_HIP(hipMemcpyAsync(out_host, dst, cnt_out * sizeof(fe_t), hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync_short(c->stream));
  return SP_OK;
}

fe_t suffix_weight_full(size_t t, size_t ell_b, size_t pair_idx, const std::vector<fe_t>& rhos) {  // src/neutronnova_zk.rs:77-87
  fe_t w = h_one();
  size_t k = pair_idx;
  for (size_t s = t + 1; s < ell_b; ++s) {
    w = fe_mul<SF>(w, (k & 1) ? rhos[s] : fe_sub<SF>(h_one(), rhos[s]));
    k >>= 1;
  }
  return w;
}

// global index of local pair 0 at round t: a round-t pair covers 2^(t+1) instances
size_t pair_base(const sp_nifs* n, size_t t) { return n->first >> (t + 1); }

int upload_weights(sp_nifs* n, size_t t, size_t pairs) {
  if (t < n->w_pairs.size() && n->w_pairs[t] == pairs) {  // uploaded at begin
    n->d_w_cur = n->d_w + n->w_off[t];
    return SP_OK;
  }
  std::vector<fe_t> w(pairs);
  for (size_t p = 0; p < pairs; ++p) w[p] = suffix_weight_full(t, n->ell_b, pair_base(n, t) + p, n->rhos);
  n->d_w_cur = n->d_w + n->n_padded;
  SP_HIP(hipMemcpyAsync(n->d_w_cur, w.data(), pairs * sizeof(fe_t), hipMemcpyHostToDevice, n->ctx->stream));
  SP_HIP(sp::stream_sync(n->ctx->stream));  // w is a stack-lifetime host buffer
  return SP_OK;
}
// every round's weights in one upload (the rounds of an unsharded or local run halve their pairs: n_padded / 2, / 4, ...)
int upload_all_weights(sp_nifs* n) {
  n->w_off.clear();
  n->w_pairs.clear();
  n->h_w.clear();
  for (size_t t = 0; t < n->ell_b; ++t) {
    const size_t pairs = n->n_padded >> (t + 1);
    if (pairs == 0) break;
    n->w_off.push_back(n->h_w.size());
    n->w_pairs.push_back(pairs);
    for (size_t p = 0; p < pairs; ++p) n->h_w.push_back(suffix_weight_full(t, n->ell_b, pair_base(n, t) + p, n->rhos));
  }
  if (n->h_w.empty()) return SP_OK;
  // (h_w lives in the object: no synchronise for the buffer's sake; the stream orders the copy in front of the rounds' kernels)
  SP_HIP(hipMemcpyAsync(n->d_w, n->h_w.data(), n->h_w.size() * sizeof(fe_t), hipMemcpyHostToDevice, n->ctx->stream));
  return SP_OK;
}

}  // namespace

extern "C" {

int sp_nifs_create(sp_ctx* c, size_t n_padded, size_t left, size_t right, sp_nifs** out) {
  if (n_padded < 2 || (n_padded & (n_padded - 1))) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_create: n_padded must be a power of two >= 2");
  if (left == 0 || right == 0 || (left & (left - 1)) || (right & (right - 1))) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_create: left/right must be powers of two");
  sp_nifs* n = new sp_nifs();
  n->ctx = c;
  n->n_padded = n_padded;
  n->left = left;
  n->right = right;
  n->total = left * right;
  while ((size_t(1) << n->left_log2) < left) ++n->left_log2;
  n->factored = (left % 256 == 0);
  const size_t layer = n->total * sizeof(fe_t);
  const size_t blocks = (n->total + 255) / 256;
  n->part_elems = 2 * blocks * n_padded;
  hipError_t e = hipSuccess;
  auto al = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16);
  };
  al((void**)&n->A[0], n_padded * layer);
  al((void**)&n->B[0], n_padded * layer);
  al((void**)&n->A[1], n_padded / 2 * layer);
  al((void**)&n->B[1], n_padded / 2 * layer);
  al((void**)&n->A[2], (n_padded / 4 ? n_padded / 4 : 1) * layer);
  al((void**)&n->B[2], (n_padded / 4 ? n_padded / 4 : 1) * layer);
  al((void**)&n->C, n_padded * layer);
  al((void**)&n->d_E, (left + right) * sizeof(fe_t));
  al((void**)&n->d_w, 2 * n_padded * sizeof(fe_t));
  al((void**)&n->d_part, n->part_elems * sizeof(fe_t));
  al((void**)&n->d_part2, n->part_elems * sizeof(fe_t));
  al((void**)&n->d_cvals, n_padded * sizeof(fe_t));
  if (e != hipSuccess) {
    sp_nifs_free(n);
    return fail(SP_ERR_NO_DEVICE, std::string("sp_nifs_create: hipMalloc: ") + hipGetErrorString(e));
  }
  *out = n;
  return SP_OK;
}

void sp_nifs_free(sp_nifs* n) {
  if (!n) return;
  void* ptrs[] = {n->A[0], n->A[1], n->A[2], n->B[0], n->B[1], n->B[2], n->C, n->A64, n->B64, n->C64, n->d_flags, n->d_large, n->d_E, n->d_w, n->d_part, n->d_part2, n->d_cvals};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (n->h_pin) hipHostFree(n->h_pin);
  delete n;
}

int sp_nifs_layer(sp_nifs* n, int which, size_t idx, sp_table** view) {
  if (which < 0 || which > 2 || idx >= n->n_padded) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_layer: bad matrix or layer index");
  sp_table* t = new sp_table();
  t->ctx = n->ctx;
  t->d = (which == 0 ? n->A[0] : which == 1 ? n->B[0] : n->C) + idx * n->total;
  t->cap = t->len = n->total;
  t->view = true;
  n->mirrors_ready = false;  // the caller is about to (re)write this layer
  *view = t;
  return SP_OK;
}

// The i64 mirrors and the global large-position list of the current layers (prep_prove's cached_step_i64, src/neutronnova_zk.rs:1548-1586):
// transcript-independent, so a caller that follows the reference builds them at prep time and passes small_values = 2 to sp_nifs_begin.
int sp_nifs_prepare_small(sp_nifs* n) {
  sp_ctx* c = n->ctx;
  const size_t np = n->n_padded;
  const unsigned blocks = (unsigned)((n->total + 255) / 256);
  if (!n->A64) {
    hipError_t e = hipMalloc((void**)&n->A64, np * n->total * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&n->B64, np * n->total * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&n->C64, np * n->total * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&n->d_flags, n->total);
    if (e == hipSuccess) e = hipMalloc((void**)&n->d_large, n->total * 4);
    if (e != hipSuccess) return fail(SP_ERR_NO_DEVICE, std::string("sp_nifs_prepare_small: hipMalloc: ") + hipGetErrorString(e));
  }
  SP_HIP(hipMemsetAsync(n->d_flags, 0, n->total, c->stream));
  // every layer's flags are OR-ed into one array: the global large_positions of prep_prove (:1548-1572)
  const fe_t* src[3] = {n->A[0], n->B[0], n->C};
  long long* dst[3] = {n->A64, n->B64, n->C64};
  for (int q = 0; q < 3; ++q)
    c->timed("nifs_to_small", 40ull * n->total * np, [&] {
      hipLaunchKernelGGL(spk::k_to_small, dim3(blocks, (unsigned)np), dim3(256), 0, c->stream, src[q], (unsigned long long)n->total, dst[q], n->d_flags);
    });
  std::vector<unsigned char> flags(n->total);
  SP_HIP(hipMemcpyAsync(flags.data(), n->d_flags, n->total, hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  std::vector<unsigned> large;
  for (size_t k = 0; k < n->total; ++k)
    if (flags[k]) large.push_back((unsigned)k);
  n->nlarge = (unsigned)large.size();
  if (n->nlarge) {
    SP_HIP(hipMemcpyAsync(n->d_large, large.data(), large.size() * 4, hipMemcpyHostToDevice, c->stream));
    for (int q = 0; q < 3; ++q)
      hipLaunchKernelGGL(spk::k_small_mask, dim3(blocks, (unsigned)np), dim3(256), 0, c->stream, dst[q], (unsigned long long)n->total, n->d_flags);
    SP_HIP(sp::stream_sync(c->stream));
  }
  n->mirrors_ready = true;
  return SP_OK;
}

int sp_nifs_begin(sp_nifs* n, const uint64_t* E_eq, const uint64_t* rhos, size_t ell_b, int small_values) {
  if ((size_t(1) << ell_b) != n->n_padded) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_begin: expected log2(n_padded) rhos");
  return sp_nifs_begin_shard(n, E_eq, rhos, ell_b, 0, small_values);
}

int sp_nifs_begin_shard(sp_nifs* n, const uint64_t* E_eq, const uint64_t* rhos, size_t ell_b, size_t first_instance, int small_values) {
  sp_ctx* c = n->ctx;
  if (ell_b > 40 || (size_t(1) << ell_b) < n->n_padded || first_instance % n->n_padded || first_instance + n->n_padded > (size_t(1) << ell_b))
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_begin_shard: the shard must be an aligned block of the 2^ell_b instances");
  n->ell_b = ell_b;
  n->first = first_instance;
  n->fold_pending = false;
  n->rhos.resize(ell_b);
  for (size_t i = 0; i < ell_b; ++i) n->rhos[i] = load_fe(rhos + 4 * i);
  n->r_bs.clear();
  n->prefix.clear();
  n->cur = 0;
  n->m = n->n_padded;
  n->rounds_done = 0;
  n->have_poly = false;
  n->T_cur = fe_zero();
  n->acc_eq = h_one();
  n->small = small_values != 0;
  SP_HIP(hipMemcpyAsync(n->d_E, E_eq, (n->left + n->right) * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  {
    int wrc = upload_all_weights(n);
    if (wrc) return wrc;
  }
  const spk::NifsGeom g = geom(n);
  const unsigned blocks = (unsigned)((n->total + 255) / 256);
  const size_t np = n->n_padded;
  if (small_values == 1 || (small_values == 2 && !n->mirrors_ready)) {
    int prc = sp_nifs_prepare_small(n);
    if (prc) return prc;
  }
  if (!n->small) n->nlarge = 0;
  // c_vals (:652-703): the local entries of the batch-wide vector (a sharded caller all-gathers the rest, sp_nifs_set_cvals)
  n->c_vals.assign(size_t(1) << ell_b, fe_zero());
  fe_t* cv_local = n->c_vals.data() + n->first;
  // small kernels: a block owns 256 * ppt consecutive k (within one x_out when factored)
  const int ppt = n->factored ? (int)std::min<size_t>(8, n->left / 256) : 1;
  const unsigned sblocks = (unsigned)((n->total + 256 * ppt - 1) / (256 * ppt));
  if (n->small) {
    c->timed("nifs_cvals", 8ull * np * n->total, [&] {
      if (n->factored)
        hipLaunchKernelGGL(spk::k_nifs_cvals_small<true>, dim3(sblocks, (unsigned)np), dim3(256), 0, c->stream, n->C64, g, n->d_part, ppt);
      else
        hipLaunchKernelGGL(spk::k_nifs_cvals_small<false>, dim3(sblocks, (unsigned)np), dim3(256), 0, c->stream, n->C64, g, n->d_part, ppt);
    });
  } else {
    c->timed("nifs_cvals", 32ull * np * n->total, [&] {
      if (n->factored)
        hipLaunchKernelGGL(spk::k_nifs_cvals<true>, dim3(blocks, (unsigned)np), dim3(256), 0, c->stream, n->C, g, n->d_part);
      else
        hipLaunchKernelGGL(spk::k_nifs_cvals<false>, dim3(blocks, (unsigned)np), dim3(256), 0, c->stream, n->C, g, n->d_part);
    });
  }
  int rc = sum_partials<1>(n, np, n->small ? sblocks : blocks, cv_local);
  if (rc) return rc;
  if (n->small && n->nlarge) {
    const unsigned lb = (n->nlarge + 255) / 256;
    hipLaunchKernelGGL(spk::k_nifs_cvals_large, dim3(lb, (unsigned)np), dim3(256), 0, c->stream, n->C, g, n->d_large, n->nlarge, n->d_part);
    std::vector<fe_t> corr(np);
    if ((rc = sum_partials<1>(n, np, lb, corr.data()))) return rc;
    for (size_t b = 0; b < np; ++b) cv_local[b] = fe_add<SF>(cv_local[b], corr[b]);
  }
  return SP_OK;
}

int sp_nifs_round(sp_nifs* n, size_t t, uint64_t out_coeffs[16]) {
  uint64_t sums[8];
  int rc = sp_nifs_round_sums(n, t, sums);
  if (rc) return rc;
  return sp_nifs_round_finish(n, t, sums, out_coeffs);
}

int sp_nifs_round_sums(sp_nifs* n, size_t t, uint64_t out_sums[8]) {
  sp_ctx* c = n->ctx;
  if (t != n->rounds_done || t >= n->ell_b || n->have_poly) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_round: rounds must be driven in order, each followed by sp_nifs_challenge");
  const spk::NifsGeom g = geom(n);
  const unsigned blocks = (unsigned)((n->total + 255) / 256);
  fe_t e0 = fe_zero(), quad = fe_zero();
  int rc;
  if (t == 0) {  // :779-851
    const size_t pairs = n->m / 2;
    if ((rc = upload_weights(n, 0, pairs))) return rc;
    fe_t q[1];
    if (n->small) {
      const int ppt = n->factored ? (int)std::min<size_t>(8, n->left / 256) : 1;
      const unsigned sblocks = (unsigned)((n->total + 256 * ppt - 1) / (256 * ppt));
      c->timed("nifs_round0_small", 32ull * pairs * n->total, [&] {
        if (n->factored)
          hipLaunchKernelGGL(spk::k_nifs_round0_small<true>, dim3(sblocks, (unsigned)pairs), dim3(256), 0, c->stream, n->A64, n->B64, g, n->d_w_cur, n->d_part, ppt);
        else
          hipLaunchKernelGGL(spk::k_nifs_round0_small<false>, dim3(sblocks, (unsigned)pairs), dim3(256), 0, c->stream, n->A64, n->B64, g, n->d_w_cur, n->d_part, ppt);
      });
      if ((rc = sum_partials<1>(n, 1, (size_t)sblocks * pairs, q))) return rc;
      quad = q[0];
      if (n->nlarge) {
        const unsigned lb = (n->nlarge + 255) / 256;
        hipLaunchKernelGGL(spk::k_nifs_round0_large, dim3(lb, (unsigned)pairs), dim3(256), 0, c->stream, n->A[0], n->B[0], g, n->d_large, n->nlarge, n->d_w_cur,
                           n->d_part);
        if ((rc = sum_partials<1>(n, 1, (size_t)lb * pairs, q))) return rc;
        quad = fe_add<SF>(quad, q[0]);
      }
    } else {
      c->timed("nifs_round0", 128ull * pairs * n->total, [&] {
        if (n->factored)
          hipLaunchKernelGGL(spk::k_nifs_round0<true>, dim3(blocks, (unsigned)pairs), dim3(256), 0, c->stream, n->A[0], n->B[0], g, n->d_w_cur, n->d_part);
        else
          hipLaunchKernelGGL(spk::k_nifs_round0<false>, dim3(blocks, (unsigned)pairs), dim3(256), 0, c->stream, n->A[0], n->B[0], g, n->d_w_cur, n->d_part);
      });
      if ((rc = sum_partials<1>(n, 1, (size_t)blocks * pairs, q))) return rc;
      quad = q[0];
    }
  } else {  // merged fold (previous challenge) + prove (:855-1097)
    const size_t fold_pairs = n->fold_pending ? n->m / 2 : n->m, prove_pairs = fold_pairs / 2;
    if (prove_pairs == 0) return fail(SP_ERR_INTERNAL, "sp_nifs_round: no local pair left to prove (a sharded batch continues on the gathered layers)");
    if ((rc = upload_weights(n, t, prove_pairs))) return rc;
    fe_t s[2];
    if (n->fold_pending) {
      const fe_t r = n->r_bs[t - 1];
      const int src = n->cur, dst = next_buf(n->cur);
      c->timed("nifs_fold_prove", 384ull * prove_pairs * n->total, [&] {
        if (n->factored)
          hipLaunchKernelGGL(spk::k_nifs_fold_prove<true>, dim3(blocks, (unsigned)prove_pairs), dim3(256), 0, c->stream, n->A[src], n->B[src], n->A[dst], n->B[dst],
                             g, r, n->d_w_cur, n->d_part);
        else
          hipLaunchKernelGGL(spk::k_nifs_fold_prove<false>, dim3(blocks, (unsigned)prove_pairs), dim3(256), 0, c->stream, n->A[src], n->B[src], n->A[dst], n->B[dst],
                             g, r, n->d_w_cur, n->d_part);
      });
      n->cur = dst;
      n->m = fold_pairs;
      n->fold_pending = false;
    } else {  // the layers are already folded (sp_nifs_fold_pending / sp_nifs_resume): evaluate only
      c->timed("nifs_prove_pairs", 128ull * prove_pairs * n->total, [&] {
        if (n->factored)
          hipLaunchKernelGGL(spk::k_nifs_prove_pairs<true>, dim3(blocks, (unsigned)prove_pairs), dim3(256), 0, c->stream, n->A[n->cur], n->B[n->cur], g, n->d_w_cur, n->d_part);
        else
          hipLaunchKernelGGL(spk::k_nifs_prove_pairs<false>, dim3(blocks, (unsigned)prove_pairs), dim3(256), 0, c->stream, n->A[n->cur], n->B[n->cur], g, n->d_w_cur, n->d_part);
      });
    }
    if ((rc = sum_partials<2>(n, 1, (size_t)blocks * prove_pairs, s))) return rc;
    // e0 = sum_j w_j (e0_ab_j - sum_v prefix[v] c_vals[2 j n_prefix + v])  (:919-925, :1016-1022); prefix = eq table of the challenges so far,
    // bit t in the upper half (:1103-1119)
    const size_t n_prefix = n->prefix.size();
    fe_t csum = fe_zero();
    for (size_t jl = 0; jl < prove_pairs; ++jl) {
      const size_t j = pair_base(n, t) + jl;  // global pair index
      fe_t cv = fe_zero();
      for (size_t v = 0; v < n_prefix; ++v) cv = fe_add<SF>(cv, fe_mul<SF>(n->prefix[v], n->c_vals[(2 * j) * n_prefix + v]));
      csum = fe_add<SF>(csum, fe_mul<SF>(cv, suffix_weight_full(t, n->ell_b, j, n->rhos)));
    }
    e0 = fe_sub<SF>(s[0], csum);
    quad = s[1];
  }
  store_fe(out_sums, e0);
  store_fe(out_sums + 4, quad);
  return SP_OK;
}

int sp_nifs_round_finish(sp_nifs* n, size_t t, const uint64_t sums[8], uint64_t out_coeffs[16]) {
  if (t != n->rounds_done || t >= n->ell_b || n->have_poly) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_round_finish: out of order");
  const fe_t e0 = load_fe(sums), quad = load_fe(sums + 4);
  // finish_round! (:703-721)
  const fe_t rho_t = n->rhos[t], one = h_one();
  const fe_t one_minus_rho = fe_sub<SF>(one, rho_t), two_rho_minus_one = fe_sub<SF>(rho_t, one_minus_rho);
  const fe_t cc = fe_mul<SF>(e0, n->acc_eq), a = fe_mul<SF>(quad, n->acc_eq);
  if (fe_is_zero(rho_t)) return fail(SP_ERR_DIVISION_BY_ZERO, "sp_nifs_round: rho_t is not invertible");
  const fe_t a_b_c = fe_mul<SF>(fe_sub<SF>(n->T_cur, fe_mul<SF>(cc, one_minus_rho)), fe_inv_vartime<SF>(rho_t));
  const fe_t b = fe_sub<SF>(fe_sub<SF>(a_b_c, a), cc);
  n->poly[0] = fe_mul<SF>(cc, one_minus_rho);
  n->poly[1] = fe_add<SF>(fe_mul<SF>(cc, two_rho_minus_one), fe_mul<SF>(b, one_minus_rho));
  n->poly[2] = fe_add<SF>(fe_mul<SF>(b, two_rho_minus_one), fe_mul<SF>(a, one_minus_rho));
  n->poly[3] = fe_mul<SF>(a, two_rho_minus_one);
  for (int i = 0; i < 4; ++i) store_fe(out_coeffs + 4 * i, n->poly[i]);
  n->have_poly = true;
  return SP_OK;
}

int sp_nifs_challenge(sp_nifs* n, const uint64_t r_b_in[4]) {
  if (!n->have_poly) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_challenge: no round polynomial pending");
  const fe_t r_b = load_fe(r_b_in), one = h_one();
  const size_t t = n->rounds_done;
  const fe_t rho_t = n->rhos[t];
  n->r_bs.push_back(r_b);
  // acc_eq *= (1 - r_b)(1 - rho_t) + r_b rho_t ; T_cur = poly_t(r_b)  (:729-731; UniPoly::evaluate, src/polys/univariate.rs:136-144)
  n->acc_eq = fe_mul<SF>(n->acc_eq, fe_add<SF>(fe_mul<SF>(fe_sub<SF>(one, r_b), fe_sub<SF>(one, rho_t)), fe_mul<SF>(r_b, rho_t)));
  fe_t eval = n->poly[0], power = r_b;
  for (int i = 1; i < 4; ++i) {
    eval = fe_add<SF>(eval, fe_mul<SF>(power, n->poly[i]));
    power = fe_mul<SF>(power, r_b);
  }
  n->T_cur = eval;
  // prefix coefficients for the c_vals term (:863-868, :1103-1119)
  if (n->prefix.empty()) {
    n->prefix = {fe_sub<SF>(one, r_b), r_b};
  } else {
    std::vector<fe_t> old = n->prefix;
    n->prefix.clear();
    for (const fe_t& x : old) n->prefix.push_back(fe_mul<SF>(x, fe_sub<SF>(one, r_b)));
    for (const fe_t& x : old) n->prefix.push_back(fe_mul<SF>(x, r_b));
  }
  n->rounds_done = t + 1;
  n->have_poly = false;
  n->fold_pending = true;
  return SP_OK;
}

// ---- sharded batches: hand-off between the shard-local rounds and the rounds on the gathered layers ------------------------
int sp_nifs_cvals(const sp_nifs* n, uint64_t* out_local) {
  memcpy(out_local, n->c_vals.data() + n->first, n->n_padded * sizeof(fe_t));
  return SP_OK;
}
int sp_nifs_set_cvals(sp_nifs* n, const uint64_t* all, size_t count) {
  if (count != n->c_vals.size()) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_set_cvals: expected one value per instance of the batch");
  memcpy(n->c_vals.data(), all, count * sizeof(fe_t));
  return SP_OK;
}
int sp_nifs_fold_pending(sp_nifs* n) {
  sp_ctx* c = n->ctx;
  if (!n->fold_pending || n->m < 2) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_fold_pending: nothing to fold");
  const unsigned blocks = (unsigned)((n->total + 255) / 256);
  const size_t pairs = n->m / 2;
  const int src = n->cur, dst = next_buf(n->cur);
  const fe_t r = n->r_bs.back();
  c->timed("nifs_fold", 96ull * 2 * pairs * n->total, [&] {
    hipLaunchKernelGGL(spk::k_nifs_fold, dim3(blocks, (unsigned)pairs, 2), dim3(256), 0, c->stream, n->A[src], n->B[src], n->A[dst], n->B[dst], (unsigned long long)n->total, r);
  });
  n->cur = dst;
  n->m = pairs;
  n->fold_pending = false;
  return SP_OK;
}
// view of current layer `idx` (after sp_nifs_fold_pending) of A (which = 0) or B (1), to ship it to the rank that continues
int sp_nifs_current_layer(sp_nifs* n, int which, size_t idx, sp_table** view) {
  if (which < 0 || which > 1 || idx >= n->m) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_current_layer: bad matrix or layer index");
  sp_table* t = new sp_table();
  t->ctx = n->ctx;
  t->d = (which == 0 ? n->A[n->cur] : n->B[n->cur]) + idx * n->total;
  t->cap = t->len = n->total;
  t->view = true;
  *view = t;
  return SP_OK;
}
int sp_nifs_state(const sp_nifs* n, uint64_t out_T_cur[4], uint64_t out_acc_eq[4]) {
  store_fe(out_T_cur, n->T_cur);
  store_fe(out_acc_eq, n->acc_eq);
  return SP_OK;
}
// Continue a batch on gathered layers: `n` holds n_padded already-folded layers of A and B (written through sp_nifs_layer views), each the
// fold of 2^t_start consecutive instances; rounds t_start .. ell_b-1 follow. c_vals = the batch-wide vector, r_bs = the t_start challenges so far.
int sp_nifs_resume(sp_nifs* n, const uint64_t* E_eq, const uint64_t* rhos, size_t ell_b, size_t t_start, const uint64_t* r_bs, const uint64_t T_cur[4],
                   const uint64_t acc_eq[4], const uint64_t* c_vals_all) {
  sp_ctx* c = n->ctx;
  if (t_start == 0 || t_start >= ell_b || (n->n_padded << t_start) != (size_t(1) << ell_b))
    return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_resume: n_padded layers of 2^t_start instances must make up the 2^ell_b batch");
  n->ell_b = ell_b;
  n->first = 0;
  n->rhos.resize(ell_b);
  for (size_t i = 0; i < ell_b; ++i) n->rhos[i] = load_fe(rhos + 4 * i);
  n->w_pairs.clear();  // (a resumed object starts at round t_start with its own pair counts: its rounds upload their weights themselves)
  n->w_off.clear();
  n->r_bs.resize(t_start);
  for (size_t i = 0; i < t_start; ++i) n->r_bs[i] = load_fe(r_bs + 4 * i);
  n->c_vals.resize(size_t(1) << ell_b);
  memcpy(n->c_vals.data(), c_vals_all, n->c_vals.size() * sizeof(fe_t));
  // prefix = eq table of the challenges so far, later challenges in the upper half (:863-868, :1103-1119)
  const fe_t one = h_one();
  n->prefix = {fe_sub<SF>(one, n->r_bs[0]), n->r_bs[0]};
  for (size_t i = 1; i < t_start; ++i) {
    std::vector<fe_t> old = n->prefix;
    n->prefix.clear();
    for (const fe_t& x : old) n->prefix.push_back(fe_mul<SF>(x, fe_sub<SF>(one, n->r_bs[i])));
    for (const fe_t& x : old) n->prefix.push_back(fe_mul<SF>(x, n->r_bs[i]));
  }
  n->T_cur = load_fe(T_cur);
  n->acc_eq = load_fe(acc_eq);
  n->cur = 0;
  n->m = n->n_padded;
  n->rounds_done = t_start;
  n->have_poly = false;
  n->fold_pending = false;
  n->small = false;
  n->nlarge = 0;
  // NOTE: weights of round t use pair indices relative to layers of 2^t instances; with first = 0 and m layers of 2^t_start they coincide
  SP_HIP(hipMemcpyAsync(n->d_E, E_eq, (n->left + n->right) * sizeof(fe_t), hipMemcpyHostToDevice, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  return SP_OK;
}

int sp_nifs_finish(sp_nifs* n, sp_table* A_out, sp_table* B_out, sp_table* C_out, uint64_t out_T_out[4], uint64_t out_eq[4]) {
  sp_ctx* c = n->ctx;
  if (n->rounds_done != n->ell_b || n->have_poly) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_finish: rounds not complete");
  if (A_out->cap < n->total || B_out->cap < n->total || (C_out && C_out->cap < n->total)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_finish: output tables too short");
  if (n->m != 2 || !n->fold_pending) return fail(SP_ERR_INTERNAL, "sp_nifs_finish: expected two layers and a pending fold before the final fold");
  const unsigned blocks = (unsigned)((n->total + 255) / 256);
  const fe_t r = n->r_bs.back();
  // final fold of the last pair (:1122-1165), straight into the caller's tables
  c->timed("nifs_fold", 96ull * 2 * n->total, [&] {
    hipLaunchKernelGGL(spk::k_nifs_fold, dim3(blocks, 1, 2), dim3(256), 0, c->stream, n->A[n->cur], n->B[n->cur], A_out->d, B_out->d, (unsigned long long)n->total, r);
  });
  n->fold_pending = false;
  int rc = SP_OK;
  if (C_out) {
  // Cz = sum_b w_b Cz_b with w = weights_from_r(r_bs) (:1168-1203); the C layers were never folded (c_vals carried their contribution).
  // (A sharded batch passes C_out = NULL here and folds each shard's C layers with its slice of the weights.)
  if ((size_t(1) << n->ell_b) != n->n_padded) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_nifs_finish: the C fold needs every instance in this object");
  std::vector<uint64_t> rb(4 * n->ell_b), w(4 * n->n_padded);
  for (size_t i = 0; i < n->ell_b; ++i) store_fe(rb.data() + 4 * i, n->r_bs[i]);
  rc = sp_weights_from_r(rb.data(), n->ell_b, n->n_padded, w.data());
  if (rc) return rc;
  std::vector<sp_table> views(n->n_padded);
  std::vector<const sp_table*> ptrs(n->n_padded);
  for (size_t b = 0; b < n->n_padded; ++b) {
    views[b].ctx = c;
    views[b].d = n->C + b * n->total;
    views[b].cap = views[b].len = n->total;
    views[b].view = true;
    ptrs[b] = &views[b];
  }
  if ((rc = sp_fold_tables(c, ptrs.data(), n->n_padded, w.data(), n->total, C_out))) return rc;
  }
  SP_HIP(sp::stream_sync_short(c->stream));
  for (sp_table* t : {A_out, B_out}) {
    t->len = n->total;
    t->lo_eff = t->hi_eff = (size_t)-1;
  }
  if (fe_is_zero(n->acc_eq)) return fail(SP_ERR_DIVISION_BY_ZERO, "sp_nifs_finish: eq(r_b, rho) is not invertible");
  store_fe(out_T_out, fe_mul<SF>(n->T_cur, fe_inv_vartime<SF>(n->acc_eq)));  // :1205-1206
  store_fe(out_eq, n->acc_eq);
  n->m = 1;
  return SP_OK;
}

int sp_to_small_vec_or_zero(sp_ctx* c, const sp_table* t, size_t cnt, int64_t* out_i64, uint8_t* out_large) {
  if (cnt > t->cap) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_to_small_vec_or_zero: range exceeds the table");
  if (cnt == 0) return SP_OK;
  long long* d_out = nullptr;
  unsigned char* d_fl = nullptr;
  SP_HIP(hipMalloc((void**)&d_out, cnt * 8));
  SP_HIP(hipMalloc((void**)&d_fl, cnt));
  SP_HIP(hipMemsetAsync(d_fl, 0, cnt, c->stream));
  hipLaunchKernelGGL(spk::k_to_small, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, t->d, (unsigned long long)cnt, d_out, d_fl);
  SP_HIP(hipMemcpyAsync(out_i64, d_out, cnt * 8, hipMemcpyDeviceToHost, c->stream));
  SP_HIP(hipMemcpyAsync(out_large, d_fl, cnt, hipMemcpyDeviceToHost, c->stream));
  SP_HIP(sp::stream_sync(c->stream));
  hipFree(d_out);
  hipFree(d_fl);
  return SP_OK;
}

int sp_pow_split_evals(const uint64_t tau_in[4], size_t ell, size_t left, size_t right, uint64_t* out) {
  if (left * right != (size_t(1) << ell) || left == 0 || right == 0) return fail(SP_ERR_INVALID_INPUT_LENGTH, "split_evals: left * right must equal 2^ell");
  const fe_t t = load_fe(tau_in);
  fe_t p = h_one(), last = h_one();
  for (size_t i = 0; i < left; ++i) {
    store_fe(out + 4 * i, p);
    last = p;
    p = fe_mul<SF>(p, t);
  }
  const fe_t step = fe_mul<SF>(last, t);
  fe_t q = h_one();
  for (size_t i = 0; i < right; ++i) {
    store_fe(out + 4 * (left + i), q);
    q = fe_mul<SF>(q, step);
  }
  return SP_OK;
}

}  // extern "C"
