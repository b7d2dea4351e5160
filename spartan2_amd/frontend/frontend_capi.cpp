// C surface of the integer R1CS generators (see r1cs_builder.hpp). Built with g++ into
// spartan2_amd/frontend/libsp_frontend.so; consumed by tests, bench.py and the host driver's callers.
#include <cstring>
#include <string>

#include "r1cs_builder.hpp"

using namespace sp_frontend;
static thread_local std::string g_err;

extern "C" {
const char* spf_last_error() { return g_err.c_str(); }

void* spf_sha256_circuit(const uint8_t* msg, size_t n) {
  try {
    return new R1CSInstanceInt(sha256_spartan_circuit(std::vector<uint8_t>(msg, msg + n)));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void* spf_synthetic_circuit(size_t n_groups, uint64_t seed, size_t num_public, unsigned shared_permille, unsigned precommitted_permille,
                            uint64_t witness_seed) {
  try {
    return new R1CSInstanceInt(synthetic_circuit(n_groups, seed, num_public, shared_permille, precommitted_permille, witness_seed));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void* spf_sha256_step_circuit(const uint8_t* block64) {
  try {
    return new R1CSInstanceInt(sha256_step_circuit(block64));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void* spf_sha256_rest_circuit(const uint8_t* msg, size_t n) {
  try {
    return new R1CSInstanceInt(sha256_rest_circuit(std::vector<uint8_t>(msg, msg + n)));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void* spf_cubic_circuit() {
  try {
    return new R1CSInstanceInt(cubic_circuit());
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void spf_free(void* p) { delete (R1CSInstanceInt*)p; }
// out: num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges, nnzA, nnzB, nnzC
void spf_dims(void* p, uint64_t out[9]) {
  auto* R = (R1CSInstanceInt*)p;
  uint64_t v[9] = {R->num_cons, R->num_shared, R->num_precommitted, R->num_rest, R->num_public, R->num_challenges,
                   R->A.data.size(), R->B.data.size(), R->C.data.size()};
  memcpy(out, v, sizeof v);
}
// see MultiEqSim (r1cs_builder.hpp): out = equality rows emitted by the SHA-256 compressions' additions, rows bellpepper's MultiEq packs them into
void spf_multieq_stats(void* p, uint64_t out[2]) {
  auto* R = (R1CSInstanceInt*)p;
  out[0] = R->stat_addmany_rows;
  out[1] = R->stat_multieq_rows;
}
void spf_csr(void* p, int which, const int64_t** data, const uint32_t** indices, const uint64_t** indptr) {
  auto* R = (R1CSInstanceInt*)p;
  CsrInt& M = which == 0 ? R->A : which == 1 ? R->B : R->C;
  *data = M.data.data();
  *indices = M.indices.data();
  *indptr = M.indptr.data();
}
const uint64_t* spf_witness(void* p) { return ((R1CSInstanceInt*)p)->witness.data(); }
const uint64_t* spf_publics(void* p) { return ((R1CSInstanceInt*)p)->publics.data(); }
// integer satisfiability self-check of (A z) o (B z) == C z
int spf_sha256_selfcheck(const uint8_t* msg, size_t n, uint64_t* num_cons, uint64_t* num_aux) {
  try {
    ConstraintSystem cs;
    std::vector<Boolean> bits;
    for (size_t k = 0; k < n; ++k)
      for (int i = 7; i >= 0; --i) bits.push_back(alloc_bit(cs, (msg[k] >> i) & 1));
    std::vector<Boolean> hash = sha256_gadget(cs, bits);
    uint8_t expect[32];
    sha256_plain(msg, n, expect);
    for (int i = 0; i < 256; ++i)
      if (hash[i].val != (bool)((expect[i / 8] >> (7 - i % 8)) & 1)) return 2;
    *num_cons = cs.num_constraints;
    *num_aux = cs.aux.size();
    return cs.is_satisfied() ? 0 : 1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
void spf_sha256_plain(const uint8_t* msg, size_t n, uint8_t* out32) { sha256_plain(msg, n, out32); }
}
