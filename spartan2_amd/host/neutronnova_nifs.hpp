// NeutronNovaNIFS::prove above the C ABI (neutronnova_nifs.cpp), shared with the ZK wrapper's driver (neutronnova_zk.cpp).
#pragma once
#include <functional>

#include "host_common.hpp"

namespace spartan2 {

typedef void (*nn_round_hook)(void* user, size_t t, const uint64_t* coeffs16, uint64_t* r_b);
void compute_tensor_decomp(size_t n, size_t* ell, size_t* left, size_t* right);  // src/neutronnova_zk.rs:56-67

struct NifsOutputs {
  uint64_t *polys, *r_bs, *E_eq, *tail, *folded_rW, *folded_X, *folded_comm;
  sp_table *A, *B, *C, *folded_W;
  // when set, nifs_prove does not fold the commitments itself: it leaves the fold here as a job that owns its inputs and writes folded_comm when run
  // with a context of the caller's choice — the folded commitment is not read by the transcript before the opening (src/neutronnova_zk.rs:2019-2065),
  // so the ZK driver runs it on a second context beside its sum-checks
  std::function<void(sp_ctx*)>* deferred_fold_commitments = nullptr;
};
// layers (and i64 mirrors) of the instances: the transcript-independent part the reference caches in prep_prove (:1520-1600)
sp_nifs* nifs_prepare(sp_ctx* ctx, const sp_shape* shape, const sp_dims& dims, size_t n, const fe_t* X, const sp_table* const* Ws, bool small_values);
void nifs_prove(sp_ctx* ctx, const sp_shape* shape, const sp_dims& dims, const sp_ck* ckey, size_t n, size_t rows, const aff_t* comms, const fe_t* X,
                const sp_table* const* Ws, const fe_t* r_W, bool small_values, sp_nifs* prepared, sp_transcript* tr, nn_round_hook hook, void* user, NifsOutputs& out);

}  // namespace spartan2
